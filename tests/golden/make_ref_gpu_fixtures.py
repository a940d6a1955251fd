#!/usr/bin/env python3
"""Makes tests/golden/ref_gpu_images.json: images rendered by THE REFERENCE'S OWN KERNEL, kept as data.

Run once on the GPU box, from the repository root, through gpurun (the product library is not involved at all):

    gpurun -- python3 tests/golden/make_ref_gpu_fixtures.py --out gpurun_out/ref_gpu_fixtures

and then copy gpurun_out/ref_gpu_fixtures/{ref_gpu_images.json, ref_gpu_*.ppm} into tests/golden/ and commit them.

What runs: oracle/_ref/ref_gpu -- the reference's complete renderer (its loader classes, build_gpu_scene, gpu_render_scene -> render_kernel -> ray_color ->
scene_hit -> bvh_hit_closest, src/gpu_render.cu:387-1108), built by oracle/Makefile in the build container from the sources where they lie (hipify-perl over the
three files that name the CUDA runtime, hipcc for gfx950, -ffp-contract=off; two definitions injected on the command line in place of inc/cuda_compat.h's own
nvcc branch, and the reference's class `texture` renamed by oracle/ref_gpu_prelude.h -- a translator build, no arithmetic or control flow touched).  That binary
travels to the GPU box with the gpurun snapshot; the reference's sources do not.

What is kept: for every job of tests/ref_gpu_jobs.py (the six parity scenes, 30 randomised views over every world file, the procedural station at 100 k and at the
bench's 1 M triangles on pose frames of the reference's pose file -- including the whole headline frame, 1920 x 1080 x 1000 samples) the image's dimensions, sha256,
lit-pixel count and one CRC32 per row; two small images in full (PPM).  A fixture is data: inputs (the job) and the reference's output (the hashes).

tests/test_gpu_reference_fixtures.py then compares the product library in math_mode 1 with these records on any GPU box, with or without oracle/_ref.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_gpu_jobs as J  # noqa: E402

# variant -> (binary under oracle/_ref/, fixture file, what it is)
VARIANTS = {
    "ocml": ("ref_gpu", "ref_gpu_images.json", "oracle/_ref/ref_gpu (the reference's kernel; hipify-perl + hipcc, -ffp-contract=off; cosf / sinf / powf from the device math library)"),
    "detmath": ("ref_gpu_detmath", "ref_gpu_detmath_images.json",
                "oracle/_ref/ref_gpu_detmath (the reference's kernel as in ref_gpu, its three libm calls mapped onto include/dsrt_detmath.h by oracle/ref_gpu_detmath_prelude.h)"),
}
KEEP_AS_PPM = ("case/mixed", "case/station_near")


def run(cmd, cwd):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=1100)
    if r.returncode != 0:
        sys.exit(f"ref_gpu failed ({r.returncode}): {' '.join(cmd)}\n{r.stdout[-1500:]}{r.stderr[-1500:]}")
    try:
        rep = json.loads(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        rep = {}
    rep["wall_s"] = round(time.time() - t0, 3)
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_gpu_fixtures"))
    ap.add_argument("--variants", default="ocml,detmath")
    ap.add_argument("--skip-headline", action="store_true", help="leave out the 1000-sample headline frame (about ten seconds of reference kernel)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for variant in args.variants.split(","):
        make(variant, args)


def make(variant, args):
    exe_name, fixture_name, what = VARIANTS[variant]
    REF_GPU = os.path.join(ROOT, "oracle", "_ref", exe_name)
    if not os.path.exists(REF_GPU):
        sys.exit(f"oracle/_ref/{exe_name} is not there: build it in the container that has /root/reference (make -C oracle), it travels with the gpurun snapshot")
    scratch = tempfile.mkdtemp(prefix="dsrt_ref_gpu_fixtures.")        # meshes and full-size images: not under gpurun_out/ (it is copied back)
    import dsrt_amd as d           # host-side only here: the pose file -> camera arithmetic (pinned bit for bit by tests/test_host_golden.py) and the mesh generator

    entries = {}
    for job in J.case_jobs() + J.fuzz_jobs():
        out = os.path.join(scratch, job["key"].replace("/", "_") + ".ppm")
        rep = run(J.ref_gpu_command(REF_GPU, job, out), J.ASSETS)
        img = J.read_ppm(out)
        entries[job["key"]] = {"job": job, "image": J.image_record(img), "reference_report": rep}
        if job["key"] in KEEP_AS_PPM:
            os.replace(out, os.path.join(args.out, exe_name + "_" + job["key"].split("/")[1] + ".ppm"))
        print(job["key"], entries[job["key"]]["image"]["sha256"][:16], entries[job["key"]]["image"]["lit"], "lit", flush=True)

    poses = d.read_pose_file(J.POSES)
    for tris, W, H, spp, frames in J.STATION_JOBS:
        if spp >= 1000 and args.skip_headline:
            continue
        obj = J.station_obj(tris, scratch)
        world = os.path.join(scratch, f"station_{tris}.world")
        open(world, "w").write(f"obj {obj}\n")
        for frame in frames:
            fr = d.pose_to_frame(poses[frame])
            cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
            out = os.path.join(scratch, f"station_{tris}_{frame}_{spp}.ppm")
            rep = run([REF_GPU, world, str(W), str(H), str(spp), "50", *cam_from, "0", "0", "0", "40", *sun, out], scratch)
            key = J.station_key(tris, W, H, spp, frame)
            entries[key] = {"job": {"key": key, "triangles": tris, "obj_sha256": J.file_sha256(obj), "pose_frame": frame, "W": W, "H": H, "spp": spp, "depth": 50, "vfov": 40.0,
                                    "from": [float(v) for v in fr.cam_in_model], "sun": [float(v) for v in fr.sun_dir_model]},
                            "image": J.image_record(J.read_ppm(out)), "reference_report": rep}
            print(key, entries[key]["image"]["sha256"][:16], entries[key]["image"]["lit"], "lit", rep.get("gpu_render_scene_ms"), "ms", flush=True)

    from dsrt_amd import meshgen
    head = {"made_by": "tests/golden/make_ref_gpu_fixtures.py", "renderer": what, "ref_gpu_sha256": J.file_sha256(REF_GPU), "mesh_generator_version": meshgen.VERSION}
    with open(os.path.join(args.out, fixture_name), "w") as f:          # one entry per line: the file diffs entry by entry
        f.write("{" + json.dumps(head)[1:-1] + ',\n"entries": {\n')
        f.write(",\n".join(json.dumps(k) + ": " + json.dumps(v, separators=(",", ":")) for k, v in entries.items()))
        f.write("\n}}\n")
    print(f"{len(entries)} images -> {os.path.join(args.out, fixture_name)}", flush=True)


if __name__ == "__main__":
    main()
