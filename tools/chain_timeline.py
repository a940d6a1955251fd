#!/usr/bin/env python3
"""Development aid: when did every pixel's chain of samples start and end?  (rng_mode 0: a pixel is one serial chain, and the frame ends
with its last chain.)  Renders one frame with the counting build and bit 27 of the development switches (dsrt_dev_set_experiment), which makes the kernel write (fetch time, end time,
wave) per pixel into the float image, and prints how the frame's end is composed."""
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
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--tune3", type=int, default=0)
    ap.add_argument("--dump", type=str, default="", help="write the per-pixel (start, end, wave) arrays to this .npz")
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
    W, H, spp = a.width, a.height, a.spp
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx = d.Context(0)
    ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
    stream = torch.cuda.current_stream().cuda_stream
    n = a.shards
    d.set_experiment(1 << 27)                 # the float image receives (fetch time, end time, wave) per pixel
    desc = d.make_desc(W, H, spp, 50, shard_rank=0, shard_count=n if n > 1 else 0, collect_counters=1, tune=(0, 0, 0, a.tune3 & 63))
    lay = d.shard_layout(desc)
    npx = lay["rgb8_bytes_padded"] // 3 if n > 1 else W * H
    rgb = torch.zeros(npx * 3, dtype=torch.uint8, device="cuda")
    f32 = torch.zeros(npx * 3, dtype=torch.float32, device="cuda")
    st = ctx.render(desc, rgb.data_ptr(), f32.data_ptr(), stream=stream, want_stats=True)
    t = f32.cpu().numpy().view(np.uint32).reshape(-1, 3)
    live = t[:, 1] != 0                                           # culled / padding pixels keep their zeros
    t0 = t[live, 0].astype(np.int64)
    t1 = t[live, 1].astype(np.int64)
    wave = t[live, 2]
    base = t0.min()
    start = (t0 - base) * 1e-5                                     # ms
    end = (t1 - base) * 1e-5
    dur = end - start
    frame_end = end.max()
    out = {"frame": a.frame, "shards": n, "spp": spp, "counting_kernel_ms": round(st.kernel_ms, 1), "pixels_timed": int(live.sum()),
           "frame_end_ms": round(float(frame_end), 1),
           "heavy_queue_empty_ms": round(st.heavy_queue_empty_ms, 1), "mean_wave_residency": round(st.wave_ticks / 1e5 / max(1e-9, st.kernel_ms) / max(1, st.waves_launched), 3),
           "chain_ms_percentiles_50_90_99_999_max": [round(float(np.percentile(dur, q)), 1) for q in (50, 90, 99, 99.9, 100)],
           "end_ms_percentiles_50_90_99_999": [round(float(np.percentile(end, q)), 1) for q in (50, 90, 99, 99.9)]}
    # the chains that end in the last 5 % of the frame: when did they start, how long are they
    late = end > 0.95 * frame_end
    out["ending_in_last_5pct"] = {"pixels": int(late.sum()), "waves": int(len(np.unique(wave[late]))),
                                  "start_ms_percentiles_10_50_90": [round(float(np.percentile(start[late], q)), 1) for q in (10, 50, 90)],
                                  "chain_ms_percentiles_10_50_90": [round(float(np.percentile(dur[late], q)), 1) for q in (10, 50, 90)]}
    # the 1000 longest chains: when did they start (costliest-first would start them all at once)
    k = min(1000, len(dur))
    top = np.argsort(dur)[-k:]
    out["longest_1000_chains"] = {"chain_ms_min_max": [round(float(dur[top].min()), 1), round(float(dur[top].max()), 1)],
                                  "start_ms_percentiles_50_90_99_max": [round(float(np.percentile(start[top], q)), 1) for q in (50, 90, 99, 100)],
                                  "end_ms_percentiles_50_90_99_max": [round(float(np.percentile(end[top], q)), 1) for q in (50, 90, 99, 100)]}
    # lower bound on the frame if every chain had started at t = 0 and run at the speed it ran at
    out["longest_chain_ms"] = round(float(dur.max()), 1)
    print(json.dumps(out), flush=True)
    if a.dump:
        np.savez_compressed(a.dump, start=start.astype(np.float32), end=end.astype(np.float32), wave=wave)


if __name__ == "__main__":
    main()
