#!/usr/bin/env python3
"""Development aid: the cooperative walk (render_kernel.hip: coop_walk) against the ordinary walk -- same bytes, audited answers, and what it buys where the chain binds.

usage: tools/coop_check.py [--tris 1000000] [--codes 1,3,4,5,7] [--quick]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

THRESHOLD = {0: "default", 1: 0, 2: 4, 3: 8, 4: 16, 5: 24, 6: 32, 7: 64}


def word(code):
    return ((code & 3) << 25) | ((code >> 2) << 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--codes", type=str, default="1,3,4,5,7")
    ap.add_argument("--quick", action="store_true", help="parity and audit only")
    ap.add_argument("--tile", type=int, default=25144)
    a = ap.parse_args()
    import torch
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj)
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a.frame])
    stream = torch.cuda.current_stream().cuda_stream
    codes = [int(x) for x in a.codes.split(",")]

    # 1. parity and audit at a small size
    W, H, spp = 320, 180, 16
    ctx = d.Context(0).set_certified_tree(True)
    ctx.upload(hs.view(d.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model)))
    d.set_experiment(word(1))
    want, _, st0 = ctx.render_to_host(d.make_desc(W, H, spp, 50), want_f32=False)
    for code in codes:
        d.set_experiment(word(code))
        for mm in (0, 1):
            d.set_experiment(word(1))
            ref, _, _ = ctx.render_to_host(d.make_desc(W, H, spp, 50, math_mode=mm), want_f32=False)
            d.set_experiment(word(code))
            got, _, st = ctx.render_to_host(d.make_desc(W, H, spp, 50, math_mode=mm), want_f32=False)
            _, _, sa = ctx.render_to_host(d.make_desc(W, H, spp, 50, math_mode=mm, collect_counters=3), want_f32=False)
            _, _, sk = ctx.render_to_host(d.make_desc(W, H, spp, 50, math_mode=mm, checked=1), want_f32=False)
            print(json.dumps({"check": "small frame", "coop_threshold": THRESHOLD[code], "math_mode": mm, "bytes_equal": bool(np.array_equal(got, ref)), "differing_pixels": int((got != ref).any(axis=2).sum()),
                              "audited": sa.certificate_audited, "audit_mismatches": sa.certificate_audit_mismatches, "fallbacks": sa.certificate_fallbacks, "rays": sa.rays,
                              "coop_rays": sa.coop_rays, "coop_visits_per_ray": round(sa.coop_visits / max(1, sa.coop_rays), 2), "coop_overflows": sa.coop_overflows,
                              "checked_flags": sk.device_flags, "kernel_ms": round(st.kernel_ms, 3)}), flush=True)
    if a.quick:
        return
    # 2. where the chain binds: one tile alone, an eighth of the frame, the whole frame
    W, H, spp = 1920, 1080, 1000
    ctx.upload(hs.view(d.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model)))
    tiles = (W // 8) * (H // 8)

    def timed(desc, n):
        lay = d.shard_layout(desc)
        buf = torch.zeros(max(lay["rgb8_bytes_padded"], W * H * 3), dtype=torch.uint8, device="cuda")
        ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)
        ms = min(ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True).kernel_ms for _ in range(n))
        return ms, buf

    base = {}
    for code in [1] + [c for c in codes if c != 1]:
        d.set_experiment(word(code))
        row = {"coop_threshold": THRESHOLD[code]}
        ms, buf = timed(d.make_desc(W, H, spp, 50, shard_rank=a.tile, shard_count=tiles, tune=(0, 0, 0, 1)), 2)
        row["one_tile_ms"] = round(ms, 2)
        ms, buf = timed(d.make_desc(W, H, spp, 50, shard_rank=4, shard_count=8), 2)
        row["share_4_of_8_ms"] = round(ms, 2)
        h8 = buf.cpu().numpy().copy()
        ms, buf = timed(d.make_desc(W, H, spp, 50, shard_rank=1, shard_count=2), 2)
        row["share_1_of_2_ms"] = round(ms, 2)
        ms, buf = timed(d.make_desc(W, H, spp, 50), 3)
        row["frame_ms"] = round(ms, 2)
        hf = buf.cpu().numpy().copy()
        if code == 1:
            base = {"h8": h8, "hf": hf}
        row["share_bytes_equal"] = bool(np.array_equal(h8, base["h8"]))
        row["frame_bytes_equal"] = bool(np.array_equal(hf, base["hf"]))
        print(json.dumps(row), flush=True)
    d.set_experiment(0)


if __name__ == "__main__":
    main()
