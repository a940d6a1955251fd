#!/usr/bin/env python3
"""Development aid: one rank's share of an N-GPU frame (shard 0 of N) rendered K times over as ONE sharded batch launch, timed on one GPU:
what the serial sample chains of rng_mode 0 cost a rank when K frames' tiles are one pool of work instead of K launches."""
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
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--rng", type=int, default=0)
    ap.add_argument("--frames", type=str, default="1,2,3,5,8")
    a = ap.parse_args()
    import torch
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj)
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh("median")
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a.frame])
    W, H, spp = 1920, 1080, a.spp
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    sun = tuple(fr.sun_dir_model)
    ctx = d.Context(0)
    ctx.upload(hs.view(cam, sun))
    stream = torch.cuda.current_stream().cuda_stream
    n = a.shards
    desc = d.make_desc(W, H, spp, 50, rng_mode=a.rng, shard_rank=0, shard_count=n if n > 1 else 0)
    part = d.shard_layout(desc)["rgb8_bytes_padded"] if n > 1 else W * H * 3
    single = None
    for k in [int(x) for x in a.frames.split(",")]:
        buf = torch.zeros(k * part, dtype=torch.uint8, device="cuda")
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.render_batch(desc, [cam] * k, [sun] * k, buf.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        single = best if single is None else single
        print(json.dumps({"shards": n, "rng_mode": a.rng, "spp": spp, "frames_in_the_launch": k, "launch_ms": round(best, 1), "ms_per_frame": round(best / k, 1)}), flush=True)


if __name__ == "__main__":
    main()
