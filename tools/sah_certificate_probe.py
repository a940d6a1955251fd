#!/usr/bin/env python3
"""CPU experiment (no GPU): is the reference's answer on its own median tree recoverable, with a certificate, from a walk of the SAH tree?

For rows of a pose frame of the procedural station, every BVH query of the oracle's reference walk is also answered on the SAH tree by
oracle/sah_certificate_probe.c (the accepted triangle of smallest t, with three exact checks: not unreachable on the reference tree, no exact tie, its reference-leaf
box passed with t_entry < t), and the two answers are compared bit for bit.  Prints the rate of rays the certificate cannot cover (they would be re-traced on the
reference tree) and the number of uncovered differences -- which must be zero for the idea to be worth a kernel.

usage: tools/sah_certificate_probe.py [--tris 100000] [--frame 98] [--width 640 --height 360] [--spp 4] [--rows 24] [--procs 8]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reference_tree_facts(nodes, idx, n_tris):
    """Per triangle: its leaf box on the reference tree, and whether a zero-thickness box lies on its root-to-leaf path (bbox_hit then never passes: t_max <= t_min)."""
    never = np.zeros(n_tris, np.uint8)
    box = np.zeros((n_tris, 6), np.float32)
    flat = np.zeros(len(nodes), np.uint8)
    thin = ((nodes["bbox_max"] - nodes["bbox_min"]) == 0).any(axis=1)
    stack = [(0, False)]
    while stack:
        i, dead = stack.pop()
        dead = dead or bool(thin[i])
        n = nodes[i]
        if n["tri_count"] > 0:
            tri = idx[n["tri_offset"]:n["tri_offset"] + n["tri_count"]]
            box[tri, :3] = n["bbox_min"]
            box[tri, 3:] = n["bbox_max"]
            if dead:
                never[tri] = 1
        else:
            stack.append((int(n["left"]), dead))
            stack.append((int(n["right"]), dead))
    del flat
    return never, box


def work(job):
    import dsrt_amd as d
    a, y0, y1 = job
    obj = f"/tmp/dsrt_bench_station_v{d.meshgen.VERSION}_{a['tris']}.obj" if a["tris"] == 1000000 else f"/tmp/dsrt_sahprobe_station_{a['tris']}.obj"
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh()
    hs2 = d.HostScene().add_obj(obj)
    hs2.build_bvh("sah")
    arr, arr2 = hs.arrays(), hs2.arrays()
    never, box = reference_tree_facts(arr["nodes"], arr["idx"], len(arr["tris"]))
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a["frame"]])
    W, H, spp = a["width"], a["height"], a["spp"]
    scene = hs.view(d.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model))
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libdsrt_sahprobe.so"))
    lib.dsrt_sahprobe_render_rows.restype = C.c_int
    lib.dsrt_sahprobe_render_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    nodes2, idx2 = np.ascontiguousarray(arr2["nodes"]), np.ascontiguousarray(arr2["idx"])
    out = (C.c_uint64 * 10)()
    cnt = (C.c_uint64 * 14)()
    rgb = np.zeros((H, W, 3), np.uint8)
    rc = lib.dsrt_sahprobe_render_rows(C.byref(scene), nodes2.ctypes.data, len(nodes2), idx2.ctypes.data, never.ctypes.data, box.ctypes.data, a["mu"], W, H, y0, y1, rgb.ctypes.data, out, cnt)
    assert rc == 0
    return {"rows": [y0, y1], "out": [int(v) for v in out], "never_hit_triangles": int(never.sum()), "triangles": int(len(never)),
            "reference_tri_tests": int(cnt[7]), "reference_nodes_entered": int(cnt[5])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=100000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--rows", type=int, default=24, help="rows of the image to render, spread evenly")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--mu", type=float, default=1e-4)
    a = vars(ap.parse_args())
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a['tris']}.obj" if a["tris"] == 1000000 else f"/tmp/dsrt_sahprobe_station_{a['tris']}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a["tris"]), obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
    H = a["height"]
    ys = sorted(set(int(round(v)) for v in np.linspace(0, H - 1, a["rows"])))
    t0 = time.time()
    with ProcessPoolExecutor(a["procs"]) as ex:
        parts = list(ex.map(work, [(a, y, y + 1) for y in ys]))
    tot = np.sum([p["out"][:8] for p in parts], axis=0)
    bad = [p for p in parts if p["out"][5]]
    rays = int(tot[0])
    rep = {"what": "reference walk on the median tree vs certified closest candidate on the SAH tree (oracle/sah_certificate_probe.c)", "mesh_triangles": parts[0]["triangles"],
           "triangles_unreachable_on_the_reference_tree": parts[0]["never_hit_triangles"], "frame": a["frame"], "size": [a["width"], a["height"]], "spp": a["spp"], "rows": len(ys),
           "bvh_queries": rays, "reference_hits": int(tot[1]), "flagged_exact_tie": int(tot[2]), "flagged_leaf_box": int(tot[3]), "flagged_zero_direction": int(tot[4]),
           "flagged_fraction_of_queries": (int(tot[2]) + int(tot[3]) + int(tot[4])) / max(1, rays), "NOT_FLAGGED_AND_DIFFERENT": int(tot[5]),
           "first_differences_(reference_tri,_certified_tri)": [[np.int64(p["out"][8]).item() if p["out"][8] < 2**63 else p["out"][8] - 2**64, p["out"][9] if p["out"][9] < 2**63 else p["out"][9] - 2**64] for p in bad[:5]],
           "tri_tests_per_query_reference_vs_sah": [sum(p["reference_tri_tests"] for p in parts) / max(1, rays), int(tot[6]) / max(1, rays)],
           "nodes_per_query_reference_entered_vs_sah_visited": [sum(p["reference_nodes_entered"] for p in parts) / max(1, rays), int(tot[7]) / max(1, rays)],
           "seconds": round(time.time() - t0, 1)}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
