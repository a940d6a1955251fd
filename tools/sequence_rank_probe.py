#!/usr/bin/env python3
"""Development aid: what one rank of an N-GPU pose-sequence job would do, timed on ONE GPU: rank r's frames (dealt by estimated cost, or round-robin; nearest first) as
one batch launch (dsrt_render_batch) plus the copy of its images to pinned host memory.  The N-GPU job has no collective on the data path,
so its frame rate is frames / (slowest rank's time)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=250)
    ap.add_argument("--ranks", type=str, default="1,2,4,8")
    ap.add_argument("--rng", type=int, default=0)
    ap.add_argument("--bvh", type=str, default="median")
    ap.add_argument("--deal", choices=["cost", "round-robin"], default="cost")
    ap.add_argument("--split", choices=["frames", "tiles"], default="frames", help="frames: whole frames per rank; tiles: every rank its tiles of ALL frames (sharded batch)")
    a = ap.parse_args()
    import torch
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj)
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh(a.bvh)
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    W, H, spp = a.width, a.height, a.spp
    frames = [i for i in range(len(poses)) if not d.pose_to_frame(poses[i]).skipped]
    cams = {}
    for i in frames:
        fr = d.pose_to_frame(poses[i])
        cams[i] = (d.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model))
    from dsrt_amd import sequence
    root = hs.arrays()["nodes"][0]
    radius = 0.5 * max(float(root["bbox_max"][k]) - float(root["bbox_min"][k]) for k in range(3))
    costs = [sequence.approach_cost(d.pose_to_frame(poses[i]).sep_m, radius) for i in frames] if a.deal == "cost" else None
    ctx = d.Context(0)
    ctx.upload(hs.view(*cams[frames[0]]))
    stream = torch.cuda.current_stream().cuda_stream
    n_img = W * H * 3
    buf = torch.zeros(len(frames) * n_img, dtype=torch.uint8, device="cuda")
    host = torch.empty(len(frames) * n_img, dtype=torch.uint8).pin_memory()
    base = None
    for n in [int(x) for x in a.ranks.split(",")]:
        times = []
        for r in range(n):
            ids = sorted(sequence.frame_assignment(frames, r, n, a.split, costs), reverse=True)
            if a.split == "tiles" and n > 1:
                desc = d.make_desc(W, H, spp, 50, rng_mode=a.rng, shard_rank=r, shard_count=n)
                n_img = d.shard_layout(desc)["rgb8_bytes_padded"]
            else:
                desc = d.make_desc(W, H, spp, 50, rng_mode=a.rng)
                n_img = W * H * 3
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.render_batch(desc, [cams[i][0] for i in ids], [cams[i][1] for i in ids], buf.data_ptr(), stream=stream)
                host[:len(ids) * n_img].copy_(buf[:len(ids) * n_img], non_blocking=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            times.append(best)
        worst = max(times)
        base = worst if base is None else base
        print(json.dumps({"ranks": n, "split": a.split, "dealt_by": a.deal if a.split == "frames" else None, "rng_mode": a.rng, "bvh": a.bvh, "spp": spp, "frames": len(frames), "rank_seconds": [round(t, 4) for t in times],
                          "frames_per_s": round(len(frames) / worst, 1), "speedup_vs_1": round(base / worst, 2)}), flush=True)


if __name__ == "__main__":
    main()
