#!/usr/bin/env python3
"""GPU: the certificate of the second tree audited at scale (DsrtRenderDesc.collect_counters = 3): for a list of pose frames of the bench mesh, every answer of the second
tree -- hit or miss -- is also walked on the reference tree by the counting build and compared (triangle, bit patterns of t, u, v; blocked-or-not for any-hit shadow rays).
Prints one JSON line per (frame, math_mode) and a total: audited answers, differing answers (must be 0), certificate fallbacks.
usage: tools/certificate_audit.py [--frames 0,30,60,70,80,85,90,95,98] [--spp 128] [--tris 1000000] [--math-modes 0,1]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="0,30,60,70,80,85,90,95,98"); ap.add_argument("--spp", type=int, default=128); ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080); ap.add_argument("--math-modes", default="0,1")
    a = ap.parse_args()
    import torch
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
    hs = d.HostScene().add_obj(obj); hs.build_bvh()
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    W, H, spp = a.width, a.height, a.spp
    ctx = d.Context(0).set_certified_tree(True)
    buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda"); ref = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    total = {"audited": 0, "differing": 0, "fallbacks": 0, "rays": 0, "images_equal_to_the_plain_walk": True}
    first = True
    for f in [int(x) for x in a.frames.split(",")]:
        fr = d.pose_to_frame(poses[f])
        cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
        if first:
            ctx.upload(hs.view(cam, tuple(fr.sun_dir_model))); first = False
        else:
            ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
        for mm in [int(x) for x in a.math_modes.split(",")]:
            st = ctx.render(d.make_desc(W, H, spp, 50, collect_counters=3, math_mode=mm), buf.data_ptr(), stream=s, want_stats=True)
            ctx.render(d.make_desc(W, H, spp, 50, math_mode=mm, tune=(0, 0, 0, 64)), ref.data_ptr(), stream=s, want_stats=True)
            same = bool(torch.equal(buf, ref))
            rec = {"frame": f, "sep_m": round(fr.sep_m, 1), "math_mode": mm, "spp": spp, "certified_tree_used": st.certified_tree_used, "rays": st.rays, "audited": st.certificate_audited,
                   "differing": st.certificate_audit_mismatches, "fallbacks": st.certificate_fallbacks, "image_equals_the_plain_reference_walk": same}
            print(json.dumps(rec), flush=True)
            total["audited"] += st.certificate_audited; total["differing"] += st.certificate_audit_mismatches; total["fallbacks"] += st.certificate_fallbacks; total["rays"] += st.rays
            total["images_equal_to_the_plain_walk"] = total["images_equal_to_the_plain_walk"] and same
    print(json.dumps({"total": total, "mesh_triangles": a.tris, "size": [W, H], "spp": spp}), flush=True)


if __name__ == "__main__":
    main()
