#!/usr/bin/env python3
"""How fast is the REFERENCE'S OWN kernel on this GPU?  oracle/_ref/ref_gpu (the reference's loader, builder and render kernel, translated CUDA -> HIP by
hipify-perl and compiled with hipcc: oracle/Makefile) on the bench's mesh and pose, at a reduced sample count, timed on its second gpu_render_scene call;
and this library on exactly the same frame.  One JSON line.  GPU box only.  What the north star calls "a hipify of src/gpu_render.cu", measured."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--compare", action="store_true", help="also render the frame in math_mode 1 (this kernel with the device math library's sinf / cosf / powf) and count differing pixels")
    a = ap.parse_args()
    import dsrt_amd as d
    from dsrt_amd import meshgen
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_gpu")
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/ref_gpu is not built")
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{a.tris}.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(a.tris), obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
    fr = d.pose_to_frame(poses[a.frame])
    W, H, spp = a.width, a.height, a.spp
    tmp = tempfile.mkdtemp(prefix="dsrt_refprobe_", dir="/tmp")
    world = os.path.join(tmp, "bench.world")
    open(world, "w").write(f"obj {obj}\n")
    cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
    r = subprocess.run([exe, world, str(W), str(H), str(spp), "50", *cam_from, "0", "0", "0", "40", *sun, os.path.join(tmp, "ref.ppm"), "1"], cwd=tmp,
                       capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        sys.exit(r.stdout[-1000:] + r.stderr[-1000:])
    ref = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    hs = d.HostScene().add_obj(obj)
    hs.build_bvh()
    cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
    ctx = d.Context(0)
    ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
    desc = d.make_desc(W, H, spp, 50)
    ctx.render_to_host(desc)
    _, _, st = ctx.render_to_host(desc)
    ms_ref = ref["gpu_render_scene_second_call_ms"]
    cmp_rec = None
    if a.compare:
        import numpy as np
        ours, _, st1 = ctx.render_to_host(d.make_desc(W, H, spp, 50, math_mode=1))
        data = open(os.path.join(tmp, "ref.ppm"), "rb").read()
        header = f"P6\n{W} {H}\n255\n".encode()
        assert data.startswith(header)
        ref_img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
        cmp_rec = {"pixels": W * H, "lit_pixels": int((ref_img.max(axis=2) > 0).sum()), "differing_pixels": int((ref_img != ours).any(axis=2).sum()),
                   "this_library_math_mode_1_kernel_ms": st1.kernel_ms,
                   "compared": "reference kernel's image_gpu.ppm against this library in math_mode 1 (device math library), rgb8 bytes"}
    print(json.dumps({"workload": f"{ref['triangles']} triangles, pose frame {a.frame}, {W}x{H} @ {spp} spp, max_depth 50", "reference_kernel_hipified": {
        "gpu_render_scene_ms_second_call_incl_copy_back_and_ppm": ms_ref, "Msamples/s": W * H * spp / ms_ref / 1e3, "build_gpu_scene_ms": ref["build_gpu_scene_ms"],
        "bvh_nodes": ref["bvh_nodes"]}, "this_library": {"kernel_ms": st.kernel_ms, "Msamples/s": W * H * spp / st.kernel_ms / 1e3},
        "ratio": ms_ref / st.kernel_ms, "image_comparison": cmp_rec}), flush=True)


if __name__ == "__main__":
    main()
