#!/usr/bin/env python3
"""Quick kernel timing sweep (development aid): frames x sizes on the procedural mesh, one line per run."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frames", type=str, default="0,60,90,98")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--stack", type=str, default="0")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--counters", action="store_true")
    ap.add_argument("--rng", type=int, default=0)
    ap.add_argument("--bvh", type=str, default="median")
    ap.add_argument("--tune", type=str, default="0:0:0", help="comma list of min_walk:adv_budget:leaf_ratio4")
    a = ap.parse_args()
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj)
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh(a.bvh)
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    ctx = d.Context(0)
    up = False
    W, H, spp = a.width, a.height, a.spp
    for fi in [int(x) for x in a.frames.split(",")]:
        fr = d.pose_to_frame(poses[fi])
        cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
        if not up:
            ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
            up = True
        else:
            ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
        for K, tune in [(int(x), tuple(int(v) for v in t.split(":"))) for x in a.stack.split(",") for t in a.tune.split(",")]:
            best = None
            for _ in range(a.reps):
                _, _, st = ctx.render_to_host(d.make_desc(W, H, spp, 50, stack_entries=K, tune=tune, rng_mode=a.rng))
                best = st.kernel_ms if best is None else min(best, st.kernel_ms)
            rec = {"frame": fi, "sep_m": round(fr.sep_m, 1), "tris": hs.view().num_triangles, "WxHxspp": f"{W}x{H}x{spp}", "K": st.lds_stack_entries, "bvh": a.bvh,
                   "rng_mode": a.rng, "tune": tune, "kernel_ms": round(best, 3), "Msamples_s": round(W * H * spp / best / 1e3, 1)}
            if a.counters:
                _, _, sc = ctx.render_to_host(d.make_desc(W, H, spp, 50, stack_entries=K, collect_counters=1, tune=tune, rng_mode=a.rng))
                rec.update({"coverage": round(sc.primary_hits / sc.samples, 4), "rays_per_sample": round(sc.rays / sc.samples, 3),
                            "Mrays_s": round(sc.rays / best / 1e3, 1), "nodes_per_ray": round(sc.nodes_entered / max(1, sc.rays), 2),
                            "tris_per_ray": round(sc.tri_tests / max(1, sc.rays), 2), "max_stack": sc.max_stack, "spills": sc.stack_spills,
                            "util_node": round(sc.internal_entered / max(1, sc.node_slots), 3), "util_tri": round(sc.tri_tests / max(1, sc.tri_slots), 3),
                            "util_adv": round(sc.adv_active / max(1, sc.adv_slots), 3),
                            "visits_frac_depth_lt_6_9_12": [round(sc.visits_depth_lt6 / max(1, sc.internal_entered), 3), round(sc.visits_depth_lt9 / max(1, sc.internal_entered), 3), round(sc.visits_depth_lt12 / max(1, sc.internal_entered), 3)],
                            "node_idle_leaf_wait_done": [round(sc.idle_at_leaf / max(1, sc.node_slots), 3), round(sc.idle_waiting / max(1, sc.node_slots), 3), round(sc.idle_done / max(1, sc.node_slots), 3)],
                            "wave_iters_node_tri_adv": [sc.node_slots // 64, sc.tri_slots // 64, sc.adv_slots // 64]})
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
