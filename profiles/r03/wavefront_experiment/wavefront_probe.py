#!/usr/bin/env python3
"""EXPERIMENT (round 3): the frame as a lock-step wavefront (DSRT_EXPERIMENT_WAVEFRONT, csrc/device_api.hip) against the persistent kernel.
  1. correctness: a small frame run to the end both ways -- the images must be equal (same arithmetic, same order per path);
  2. bulk throughput: the bench frame, the first --iters iterations only (the regime with more paths than lanes), rays traced per second, against the
     persistent kernel's rays per second over its whole launch.
One JSON line per measurement.  GPU box only."""
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
    ap.add_argument("--slots", type=str, default="1048576,2097152")
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--walk", type=str, default="0,16,32", help="min_walk_iters (x10) of the trace kernel: how long empty lanes wait for a refill")
    a = ap.parse_args()
    import numpy as np
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a.frame])
    ctx = d.Context(0)

    def env(slots=None, iters=None):
        for k, v in (("DSRT_EXPERIMENT_WAVEFRONT", slots), ("DSRT_EXPERIMENT_WAVEFRONT_ITERS", iters)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    # 1. correctness on a small frame, run to the end
    W, H, spp = 320, 180, 8
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
    env()
    want, _, st0 = ctx.render_to_host(d.make_desc(W, H, spp, 50))
    env(slots=W * H)
    got, _, st1 = ctx.render_to_host(d.make_desc(W, H, spp, 50))
    print(json.dumps({"check": f"{W}x{H}x{spp} to the end", "images_equal": bool(np.array_equal(want, got)), "differing_pixels": int((want != got).any(axis=2).sum()),
                      "persistent_ms": st0.kernel_ms, "wavefront_ms": st1.kernel_ms, "wavefront_iterations": st1.samples, "wavefront_rays": st1.rays}), flush=True)

    # 2. bulk throughput on the bench frame
    W, H, spp = 1920, 1080, 1000
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
    env()
    ctx.render_to_host(d.make_desc(W, H, spp, 50))
    _, _, st = ctx.render_to_host(d.make_desc(W, H, spp, 50))
    _, _, sc = ctx.render_to_host(d.make_desc(W, H, spp, 50, collect_counters=1))
    base = sc.rays / st.kernel_ms / 1e6
    print(json.dumps({"persistent_kernel": {"kernel_ms": st.kernel_ms, "rays": sc.rays, "Grays/s": base}}), flush=True)
    for slots in [int(v) for v in a.slots.split(",")]:
        for walk in [int(v) for v in a.walk.split(",")]:
            env(slots=slots, iters=a.iters)
            desc = d.make_desc(W, H, spp, 50, tune=(walk, 0, 0, 0))
            ctx.render_to_host(desc)
            _, _, sw = ctx.render_to_host(desc)
            print(json.dumps({"wavefront": {"slots": slots, "min_walk_iters_x10": walk or 64, "iterations": sw.samples, "rays": sw.rays, "ms": sw.kernel_ms,
                                            "Grays/s": sw.rays / sw.kernel_ms / 1e6, "vs_persistent": sw.rays / sw.kernel_ms / 1e6 / base,
                                            "ms_per_iteration": sw.kernel_ms / max(1, sw.samples), "rays_per_iteration": sw.rays / max(1, sw.samples)}}), flush=True)
    # lane slots of the trace kernel (counting build of it), one configuration
    env(slots=2097152, iters=a.iters)
    os.environ["DSRT_EXPERIMENT_WAVEFRONT_COUNT"] = "1"
    _, _, sw = ctx.render_to_host(d.make_desc(W, H, spp, 50, tune=(10, 0, 0, 0)))
    os.environ.pop("DSRT_EXPERIMENT_WAVEFRONT_COUNT", None)
    print(json.dumps({"trace_kernel_lane_slots": {"node_loop_active": sw.internal_entered / max(1, sw.node_slots), "parked_at_leaf": sw.idle_at_leaf / max(1, sw.node_slots),
                                                  "empty_waiting_for_refill": sw.idle_waiting / max(1, sw.node_slots), "out_of_rays": sw.idle_done / max(1, sw.node_slots),
                                                  "leaf_loop_active": sw.tri_tests / max(1, sw.tri_slots), "refill_passes": sw.adv_slots // 64, "node_iterations": sw.node_slots // 64,
                                                  "nodes_per_ray": sw.nodes_entered / max(1, sw.rays), "rays": sw.rays, "ms": sw.kernel_ms}}), flush=True)
    env()


if __name__ == "__main__":
    main()
