#!/usr/bin/env python3
"""Interleaved A/B of render settings in ONE process on ONE box (development aid): each configuration `rng:switches[:loop knobs]` (the low six bits
of `switches` are DsrtRenderDesc.tune[3], the DSRT_TUNE_* flags of include/dsrt.h; the rest goes to dsrt_dev_set_experiment, the
library's development switches, csrc/device_api.hip) is rendered
--reps times, round-robin, and the medians of the kernel times (HIP events) are printed.  Devices differ by a few per cent, so
settings are only ever compared inside one run of this tool."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", help="rng_mode:tune3[:min_walk:adv_budget:leaf_ratio] (e.g. 0:0 0:32 1:16 0:0:96:12:16)")
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--bvh", type=str, default="median")
    ap.add_argument("--lambertian", action="store_true", help="the same mesh with every material made Lambertian (Ks 0, d 1): a scene that qualifies for the LEAN kernels")
    ap.add_argument("--certified", action="store_true", help="upload with the certified second tree (dsrt_ctx_set_certified_tree); a sixth config field of 1 then renders on the "
                                                              "plain reference walk (DSRT_TUNE_REFERENCE_WALK), e.g. 0:0 0:0:0:0:0:1")
    a = ap.parse_args()
    import torch
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
    if a.lambertian:
        lean = obj[:-4] + "_lambertian.obj"
        if not os.path.exists(lean):
            import re
            mtl_src = [l.split()[1] for l in open(obj) if l.startswith("mtllib")]
            text = open(os.path.join(os.path.dirname(obj), mtl_src[0])).read() if mtl_src else ""
            text = re.sub(r"^Ks .*$", "Ks 0.0 0.0 0.0", text, flags=re.M)
            text = re.sub(r"^d .*$", "d 1.0", text, flags=re.M)
            text = re.sub(r"^Ni .*$", "Ni 1.0", text, flags=re.M)
            open(lean[:-4] + ".mtl", "w").write(text)
            with open(obj) as src, open(lean, "w") as dst:
                for l in src:
                    dst.write(f"mtllib {os.path.basename(lean)[:-4]}.mtl\n" if l.startswith("mtllib") else l)
        obj = lean
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh(a.bvh)
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a.frame])
    W, H, spp = a.width, a.height, a.spp
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx = d.Context(0)
    if a.certified:
        ctx.set_certified_tree(True)
    ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
    stream = torch.cuda.current_stream().cuda_stream
    n = a.shards
    descs = []
    for c in a.configs:
        parts = [int(x) for x in c.split(":")]
        rng, t3 = parts[0], parts[1]
        t0, t1, t2 = (parts[2:5] + [0, 0, 0])[:3]                     # optional: min_walk_iters : advance_budget : leaf_ratio4
        walk = 64 if len(parts) > 5 and parts[5] else 0               # DSRT_TUNE_REFERENCE_WALK
        descs.append((d.make_desc(W, H, spp, 50, shard_rank=0, shard_count=n if n > 1 else 0, rng_mode=rng, tune=(t0, t1, t2, (t3 & 63) | walk)), (t3 & 0xFFFFFFFF) & ~63))
    lay = d.shard_layout(descs[0][0])
    buf = torch.zeros(lay["rgb8_bytes_padded"] if n > 1 else W * H * 3, dtype=torch.uint8, device="cuda")
    import time
    times = [[] for _ in descs]
    walls = [[] for _ in descs]
    for dsc, xp in descs:
        d.set_experiment(xp)
        ctx.render(dsc, buf.data_ptr(), stream=stream, want_stats=True)           # warm-up
    for _ in range(a.reps):
        for i, (dsc, xp) in enumerate(descs):
            d.set_experiment(xp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            times[i].append(ctx.render(dsc, buf.data_ptr(), stream=stream, want_stats=True).kernel_ms)
            walls[i].append((time.perf_counter() - t0) * 1e3)                      # the whole call: pre-pass, probe, render, (resolve)
    for c, t, w in zip(a.configs, times, walls):
        print(json.dumps({"config": c, "frame": a.frame, "shards": n, "spp": spp, "median_ms": round(statistics.median(t), 2), "median_wall_ms": round(statistics.median(w), 2),
                          "all_ms": [round(x, 1) for x in t]}), flush=True)


if __name__ == "__main__":
    main()
