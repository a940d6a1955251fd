#!/usr/bin/env python3
"""Development aid: the lone sample chain, measured.  One 8x8 tile of the headline frame as a launch of its own (shard `t` of `tiles`): one wave holds its 64 pixels,
nothing else runs on the device, and the kernel time is the longest pixel's serial chain (rng_mode 0: one LCG stream per pixel).  The counting build says how many wave
iterations of the node loop, the leaf pass and the advance pass that chain took, so time / iterations prices one dependent step of each kind when a wave runs ALONE on its SIMD --
the number that bounds an N-GPU share, not the throughput figure of a full device.

usage: tools/chain_probe.py [--scan-step 7] [--top 6] [--spp 1000] [--certified]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--scan-spp", type=int, default=8)
    ap.add_argument("--scan-step", type=int, default=7, help="every n-th tile is looked at in the scan")
    ap.add_argument("--top", type=int, default=6)
    ap.add_argument("--certified", action="store_true")
    ap.add_argument("--tune", type=str, default="0:0:0:0")
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
    W, H = a.width, a.height
    ctx = d.Context(0)
    if a.certified:
        ctx.set_certified_tree(True)
    ctx.upload(hs.view(d.frame_camera(fr, 40.0, W, H, a.spp, 50), tuple(fr.sun_dir_model)))
    stream = torch.cuda.current_stream().cuda_stream
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    tune = tuple(int(v) for v in a.tune.split(":"))
    tune_plain = (tune[0], tune[1], tune[2], tune[3] | 1)          # natural order: no pre-pass, no probe launch around a one-tile launch

    def one(t, spp, counters):
        desc = d.make_desc(W, H, spp, 50, shard_rank=t, shard_count=tiles, collect_counters=1 if counters else 0, tune=tune_plain)
        lay = d.shard_layout(desc)
        buf = torch.zeros(lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
        return ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)

    scan = []
    for t in range(0, tiles, a.scan_step):
        st = one(t, a.scan_spp, True)
        if st.rays:
            scan.append((st.rays, t))
    scan.sort(reverse=True)
    print(json.dumps({"scan": {"tiles_looked_at": len(range(0, tiles, a.scan_step)), "with_rays": len(scan), "spp": a.scan_spp, "top": scan[:a.top]}}), flush=True)
    rows = []
    for rays, t in scan[:a.top]:
        one(t, a.spp, False)
        ms = min(one(t, a.spp, False).kernel_ms for _ in range(2))
        sc = one(t, a.spp, True)
        it = [sc.node_slots // 64, sc.tri_slots // 128, sc.adv_slots // 64]          # (the leaf pass counts two slots per pair-record step)
        row = {"tile": t, "kernel_ms": round(ms, 2), "counting_ms": round(sc.kernel_ms, 2), "rays": sc.rays, "samples": sc.samples, "rays_per_sample": round(sc.rays / max(1, sc.samples), 2),
               "wave_iterations_node_leaf_advance": it, "nodes_entered_per_ray": round(sc.nodes_entered / max(1, sc.rays), 2),
               "lane_use_node_adv": [round(sc.internal_entered / max(1, sc.node_slots), 3), round(sc.adv_active / max(1, sc.adv_slots), 3)],
               "us_per_wave_iteration_if_all_alike": round(ms * 1e3 / max(1, sum(it)), 3), "certificate_fallbacks": sc.certificate_fallbacks}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if len(rows) >= 3:
        A = np.array([r["wave_iterations_node_leaf_advance"] for r in rows], float)
        y = np.array([r["kernel_ms"] for r in rows], float) * 1e3
        x, *_ = np.linalg.lstsq(A, y, rcond=None)
        print(json.dumps({"least_squares_us_per_iteration_node_leaf_advance": [round(float(v), 3) for v in x], "certified_tree": bool(a.certified), "tune": a.tune,
                          "shares_of_the_longest_chain": [round(float(A[0][i] * x[i] / y[0]), 3) for i in range(3)]}), flush=True)


if __name__ == "__main__":
    main()
