#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE'S OWN host code (oracle/_ref/ref_host).

Run in the build container only (needs /root/reference to have built oracle/_ref):
    make -C oracle ref && python3 tests/golden/make_golden.py

What is a reference-made vector here (ref_*.json / ref_*.npz): ABI offsets; pose -> cam_in_model / sun_dir_model /
GPUCamera for all 99 poses; flattened triangles / materials / spheres / texture pool and the median-split BVH (nodes +
tri_indices) for the asset scenes; ray-level answers of the reference's CPU classes.  The inputs (assets/) are ours:
small OBJ/MTL/PPM/PNG files and "world description" files written by this script.

What is NOT made here: the render loop's reference-made images.  Those need a GPU: tests/golden/make_ref_gpu_fixtures.py runs the reference's
own kernel (oracle/_ref/ref_gpu, ref_gpu_detmath) on an MI355X and writes ref_gpu_images.json / ref_gpu_detmath_images.json -- see oracle/dsrt_oracle.h.
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ASSETS = os.path.join(HERE, "assets")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_host")
POSE_FILE = "/root/reference/orbit_sim/rendezvous_1s_dt0_01s.txt"
sys.path.insert(0, ROOT)


def write(path, text, newline="\n"):
    with open(path, "w", newline=newline) as f:
        f.write(text)


def png_bytes(rgb):
    """Minimal 8-bit RGB PNG (filter 0 rows, zlib)."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")


def make_assets():
    os.makedirs(ASSETS, exist_ok=True)
    import dsrt_amd  # noqa: F401  (package import needs the built library; only meshgen is used here)
    from dsrt_amd import meshgen
    meshgen.generate(os.path.join(ASSETS, "station_3k.obj"), 3000)

    # C1: the RTIOW three-sphere scene on its ground sphere (BASELINE.json configs[0]); placement is ours.
    write(os.path.join(ASSETS, "c1_spheres.world"), """# ground + lambertian + dielectric + metal
mat ground lambertian 0.8 0.8 0.0
mat center lambertian 0.1 0.2 0.5
mat left dielectric 1.5
mat right metal 0.8 0.6 0.2 0.3
sphere 0 -100.5 -1 100 ground
sphere 0 0 -1 0.5 center
sphere -1 0 -1 0.5 left
sphere 1 0 -1 0.5 right
""")
    # emissive sphere + triangles: exercises the 50/50 light/BRDF mixture (src/gpu_render.cu:871-932)
    write(os.path.join(ASSETS, "lights.world"), """mat floor lambertian 0.7 0.7 0.7
mat red lambertian 0.65 0.05 0.05
mat lamp light 15 15 12
mat lamp2 light 4 6 9
mat mirror metal 0.9 0.9 0.9 0.05
mat glass dielectric 1.5
tri -5 0 -5  5 0 -5  5 0 5 floor
tri -5 0 -5  5 0 5  -5 0 5 floor
tri -5 0 -5  -5 6 -5  5 6 -5 red
tri -5 0 -5  5 6 -5  5 0 -5 red
sphere 0 5 0 0.8 lamp
sphere -2.5 3.5 1.5 0.4 lamp2
sphere 1.5 1 0 1.0 mirror
sphere -1.5 0.8 1 0.8 glass
""")
    # textured quad: v/vt faces, one PPM texture and one PNG texture, one material with map_Kd AND Ks
    tex = np.zeros((8, 8, 3), np.uint8)
    for y in range(8):
        for x in range(8):
            tex[y, x] = (32 * x + 7, 255 - 30 * y, 255 if (x + y) % 2 else 20)
    with open(os.path.join(ASSETS, "checker.ppm"), "wb") as f:
        f.write(b"P6\n8 8\n255\n" + tex.tobytes())
    tex2 = np.zeros((5, 7, 3), np.uint8)
    for y in range(5):
        for x in range(7):
            tex2[y, x] = (40 * y + 10, 35 * x + 5, (x * y * 11) % 256)
    with open(os.path.join(ASSETS, "stripes.png"), "wb") as f:
        f.write(png_bytes(tex2))
    write(os.path.join(ASSETS, "textured.mtl"), """newmtl checker
Kd 0.5 0.5 0.5
Ks 0.9 0.9 0.9
map_Kd checker.ppm
newmtl stripes
Kd 0.2 0.3 0.4
map_Kd stripes.png
newmtl missing_tex
Kd 0.3 0.3 0.3
map_Kd does_not_exist.png
newmtl plain
Kd 0.6 0.4 0.2
""")
    write(os.path.join(ASSETS, "textured.obj"), """mtllib textured.mtl
v -2 0 -1
v 2 0 -1
v 2 3 -1
v -2 3 -1
v -2 0 -3
v 2 0 -3
v 2 3 -3.5
v -2 3 -3.5
v -4 0 2
v 4 0 2
v 4 0 -6
v -4 0 -6
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vt 2.5 -0.75
usemtl checker
f 1/1 2/2 3/3 4/4
usemtl stripes
f 5/1 6/2 7/5 8/4
usemtl missing_tex
f 9/1 10/2 11/3
usemtl plain
f 9 11 12
""")
    # loader quirks: CRLF, comments, polygon fan, v//vn and v/vt/vn tokens, unknown and absent usemtl, Ke, d, short face,
    # nine coincident triangles (forces a leaf bigger than 4: zero centroid extent), a material defined twice.
    # (No out-of-range face index here: the reference reads past its vertex array for those -- undefined behaviour, not a
    # vector.  Our loader skips such faces; tests/test_host_golden.py covers that on its own.)
    lines = ["# quirks", "mtllib quirks.mtl", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 1.5 0.25", "v 2 0 1", "v 2 1 1", "v 3 0.5 2",
             "vn 0 0 1", "vt 0.25 0.5",
             "f 1 2 3 4 5", "usemtl glow", "f 1//1 2//1 3//1", "usemtl nosuch", "f 2/1/1 6/1/1 7/1/1", "usemtl glassy", "f 6 8 7",
             "usemtl", "f 1 3 4", "f 1 2", "usemtl shiny"]
    lines += ["f 6 7 8"] * 9
    lines += ["usemtl dull", "f 4 3 5"]
    write(os.path.join(ASSETS, "quirks.obj"), "\r\n".join(lines) + "\r\n", newline="")
    write(os.path.join(ASSETS, "quirks.mtl"), """newmtl glow
Kd 0.1 0.1 0.1
Ke 3 2 1
newmtl glassy
Kd 0.9 0.9 0.9
d 0.5
Ni 1.33
newmtl shiny
Ks 0.04 0.03 0.02
Ns 25
newmtl dull
Ks 0.02 0.02 0.02
Kd 0.25 0.5 0.75
newmtl shiny
Ks 0.4 0.3 0.2
Ns 25
""")
    for name, body in (("station_3k", "obj station_3k.obj\n"), ("textured", "obj textured.obj\n"), ("quirks", "obj quirks.obj 2.5\n"),
                       ("mixed", "mat m lambertian 0.5 0.5 0.5\nsphere 0 8 0 2 m\nobj quirks.obj\nobj textured.obj 0.5\n")):
        write(os.path.join(ASSETS, name + ".world"), body)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_ref(*args, cwd=None):
    out = subprocess.run([REF, *map(str, args)], check=True, capture_output=True, cwd=cwd)
    return json.loads(out.stdout.decode())


def make_matkat():
    """The reference's material / frame helpers, executed (inc/vec3.h:136-147 reflect / refract, inc/material.h:28-32 reflectance, :123-137
    metal::scatter at fuzz 0, :153-180 dielectric::scatter on the total-internal-reflection branch, inc/onb.h:47-56 build_from_w)."""
    json.dump(run_ref("matkat", 64), open(os.path.join(HERE, "ref_matkat.json"), "w"))


def main():
    if "--matkat-only" in sys.argv:
        make_matkat()
        return
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_host is missing: run `make -C oracle ref` in the build container")
    make_assets()
    from dsrt_amd import capi

    json.dump(run_ref("abi"), open(os.path.join(HERE, "ref_abi.json"), "w"), indent=1)
    json.dump(run_ref("poses", POSE_FILE, 640, 360, 64, 50, 40), open(os.path.join(HERE, "ref_poses_640x360.json"), "w"))
    json.dump(run_ref("hitkat", 240), open(os.path.join(HERE, "ref_hitkat.json"), "w"))
    # the reference's host-compilable DEVICE helpers (inc/rtweekend.h:126-202, inc/camera.h:35-61): LCG, rejection loop, cosine direction, camera ray
    json.dump(run_ref("devkat", 96), open(os.path.join(HERE, "ref_devkat.json"), "w"))
    make_matkat()
    cams = []
    for spec in ((-2, 2, 1, 0, 0, -1, 20, 200, 112, 16, 50), (0, 0, 60, 0, 0, 0, 40, 640, 360, 64, 50), (13, 2, 3, 0, 0, -1, 20, 1920, 1080, 1000, 50),
                 (0, 3, 9, 0, 2, 0, 45, 200, 112, 16, 12), (0, 40, 0.001, 0, 0, 0, 35, 320, 240, 4, 5)):
        cams.append({"args": list(spec), **run_ref("camera", *spec)})
    json.dump(cams, open(os.path.join(HERE, "ref_cameras.json"), "w"), indent=1)

    # the pose file itself is a data fixture of the reference (input, not code)
    with open(POSE_FILE) as f, open(os.path.join(HERE, "rendezvous_1s_dt0_01s.txt"), "w") as g:
        g.write(f.read())

    tmp = os.path.join(HERE, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    scenes = {}
    for name in ("c1_spheres", "lights", "station_3k", "textured", "quirks", "mixed"):
        prefix = os.path.join(tmp, name)
        counts = run_ref("scene", name + ".world", prefix, cwd=ASSETS)
        arrs = {
            "tris": np.fromfile(prefix + ".tris.bin", capi.TRI_DTYPE), "spheres": np.fromfile(prefix + ".spheres.bin", capi.SPHERE_DTYPE),
            "mats": np.fromfile(prefix + ".mats.bin", capi.MAT_DTYPE), "idx": np.fromfile(prefix + ".idx.bin", "<i4"),
            "nodes": np.fromfile(prefix + ".nodes.bin", capi.NODE_DTYPE), "texhdr": np.fromfile(prefix + ".texhdr.bin", capi.TEXHDR_DTYPE),
            "texpool": np.fromfile(prefix + ".texpool.bin", "<f4"),
        }
        scenes[name] = {"counts": counts, "sha256": {k: sha(v) for k, v in arrs.items()}}
        np.savez_compressed(os.path.join(HERE, f"ref_scene_{name}.npz"), **arrs)
    json.dump(scenes, open(os.path.join(HERE, "ref_scenes.json"), "w"), indent=1)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
