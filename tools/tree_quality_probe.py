#!/usr/bin/env python3
"""CPU experiment: how many nodes and triangle tests per ray does a tree cost?  Renders rows of the headline frame with the oracle on the tree the host builder makes
(the oracle walks whatever tree the scene carries) and prints nodes entered and triangle tests per BVH query.  Builder knobs come from the environment
(DSRT_SAH_BINS, DSRT_SAH_SWEEP: host/bvh_sah.cpp).  usage: tools/tree_quality_probe.py [--bvh sah|median] [--rows 32] [--spp 4]"""
import argparse, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    a, y = job
    import dsrt_amd as d
    from dsrt_amd import meshgen
    from conftest import Oracle
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a['tris']}.obj"
    hs = d.HostScene().add_obj(obj); hs.build_bvh(a["bvh"])
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt")); fr = d.pose_to_frame(poses[a["frame"]])
    W, H, spp = 1920, 1080, a["spp"]
    scene = hs.view(d.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model))
    _, _, cnt = Oracle().render(scene, W, H, y0=y, y1=y + 1, want_f32=False)
    return cnt, len(hs.arrays()["nodes"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bvh", default="sah"); ap.add_argument("--rows", type=int, default=32); ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--tris", type=int, default=1000000); ap.add_argument("--frame", type=int, default=98)
    a = vars(ap.parse_args())
    import numpy as np
    ys = sorted(set(int(round(v)) for v in np.linspace(0, 1079, a["rows"])))
    t0 = time.time()
    with ProcessPoolExecutor(8) as ex:
        parts = list(ex.map(work, [(a, y) for y in ys]))
    tot = {k: sum(p[0][k] for p in parts) for k in parts[0][0]}
    print(json.dumps({"bvh": a["bvh"], "env": {k: v for k, v in os.environ.items() if k.startswith("DSRT_SAH")}, "tree_nodes": parts[0][1], "rays": tot["rays"],
                      "nodes_entered_per_ray": tot["nodes_entered"] / tot["rays"], "internal_per_ray": tot["internal_entered"] / tot["rays"], "tri_tests_per_ray": tot["tri_tests"] / tot["rays"],
                      "box_tests_per_ray": tot["box_tests"] / tot["rays"], "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
