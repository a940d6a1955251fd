#!/usr/bin/env python3
"""Development aid: what one rank of an N-GPU job would do, timed on ONE GPU (shard `r` of `N`, kernel time from HIP events).
Strong scaling in rng_mode 0 is bounded by the slowest pixel's serial sample chain; this shows the curve without N GPUs."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--shards", type=str, default="1,2,4,8")
    ap.add_argument("--rng", type=int, default=0)
    ap.add_argument("--bvh", type=str, default="median")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--tune", type=str, default="0:0:0:0")
    ap.add_argument("--counters", action="store_true")
    ap.add_argument("--certified", action="store_true", help="upload with the certified second tree (dsrt_ctx_set_certified_tree)")
    ap.add_argument("--ranks", type=str, default="", help="which ranks of the share to time (default: first, middle, last)")
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
    fr = d.pose_to_frame(poses[a.frame])
    W, H, spp = a.width, a.height, a.spp
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx = d.Context(0)
    if a.certified:
        ctx.set_certified_tree(True)
    ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
    stream = torch.cuda.current_stream().cuda_stream
    base = None
    for n in [int(x) for x in a.shards.split(",")]:
        ranks = [int(x) for x in a.ranks.split(",")] if a.ranks else sorted(set([0, n // 2, n - 1]))
        times = []
        for r in ranks:
            desc = d.make_desc(W, H, spp, 50, shard_rank=r if n > 1 else 0, shard_count=n if n > 1 else 0, rng_mode=a.rng, tile_size=a.tile, tune=tuple(int(v) for v in a.tune.split(":")))
            lay = d.shard_layout(desc)
            buf = torch.zeros(lay["rgb8_bytes_padded"] if n > 1 else W * H * 3, dtype=torch.uint8, device="cuda")
            ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)
            times.append(min(ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True).kernel_ms for _ in range(2)))
        extra = {}
        if a.counters:
            desc = d.make_desc(W, H, spp, 50, shard_rank=0, shard_count=n if n > 1 else 0, rng_mode=a.rng, tile_size=a.tile, collect_counters=1,
                               tune=tuple(int(v) for v in a.tune.split(":")))
            sc = ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)
            extra = {"counting_ms": round(sc.kernel_ms, 1), "samples": sc.samples, "rays": sc.rays, "util_node": round(sc.internal_entered / max(1, sc.node_slots), 3),
                     "util_adv": round(sc.adv_active / max(1, sc.adv_slots), 3),
                     "node_idle_leaf_wait_done": [round(sc.idle_at_leaf / max(1, sc.node_slots), 3), round(sc.idle_waiting / max(1, sc.node_slots), 3), round(sc.idle_done / max(1, sc.node_slots), 3)],
                     "wave_iters_node_tri_adv": [sc.node_slots // 64, sc.tri_slots // 64, sc.adv_slots // 64],
                     "queues_empty_heavy_light_last_exit_ms": [round(sc.heavy_queue_empty_ms, 1), round(sc.light_queue_empty_ms, 1), round(sc.last_wave_exit_ms, 1)], "waves": sc.waves_launched, "mean_wave_residency": round(sc.wave_ticks / 1e5 / max(1e-9, sc.kernel_ms) / max(1, sc.waves_launched), 3)}
        worst = max(times)
        base = worst if base is None else base
        print(json.dumps({"frame": a.frame, "certified_tree": bool(a.certified), "rng_mode": a.rng, "bvh": a.bvh, "shards": n, "tune": a.tune, "ranks_timed": ranks, "kernel_ms": [round(t, 2) for t in times],
                          "speedup_vs_1": round(base / worst, 2), "efficiency": round(base / worst / n, 3), **extra}), flush=True)


if __name__ == "__main__":
    main()
