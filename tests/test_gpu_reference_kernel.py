"""The reference's OWN render kernel, executed -- the pin nothing else in this pipeline could give.

oracle/_ref/ref_gpu is the reference's complete renderer (its OBJ / world loader classes, build_gpu_scene, gpu_render_scene -> render_kernel ->
ray_color -> scene_hit -> bvh_hit_closest, src/gpu_render.cu:387-1108), built by oracle/Makefile from the sources where they lie: the three files
that name the CUDA runtime are translated to HIP by the image's own hipify-perl in a scratch directory and compiled with hipcc for gfx950 (no line
of arithmetic or control flow is touched; oracle/ref_gpu_driver.cpp has the details).  What such a build cannot share with the product is the
three library functions the product replaces by include/dsrt_detmath.h by default (sinf, cosf, powf: DESIGN.md section 2); DsrtRenderDesc.math_mode 1
selects the same kernels compiled with exactly those three taken from the device math library -- and then the two programs compute, operation for
operation, the same thing, and their images must be equal BYTE FOR BYTE: every scene of the parity suite, ties, traversal order, Russian roulette,
the mixture branch, dielectrics, textures and all.  math_mode 0 equals the CPU oracle bit for bit (tests/test_gpu_parity.py); the two modes are one
source compiled twice and differ in those three functions only.  Everything here goes through the product library's C ABI.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ASSETS, load_world, require_ref_binary
from test_oracle import CASES, SUN

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "ref_gpu")


def _read_ppm(path):
    data = open(path, "rb").read()
    assert data[:3] == b"P6\n"
    parts, pos = [], 3
    while len(parts) < 3:                                   # width height maxval, whitespace separated
        end = pos
        while data[end:end + 1] not in (b" ", b"\n"):
            end += 1
        parts.append(int(data[pos:end]))
        pos = end + 1
    w, h, _ = parts
    return np.frombuffer(data[pos:pos + w * h * 3], np.uint8).reshape(h, w, 3)


REF_GPU_FMA = os.path.join(ROOT, "oracle", "_ref", "ref_gpu_fma")


def _reference_image(name, tmp_path, exe=REF_GPU):
    world, cam_args, spp = CASES[name]
    (fx, fy, fz), (ax, ay, az), vfov, W, H, depth = cam_args
    out = tmp_path / f"ref_{name}_{os.path.basename(exe)}.ppm"
    cmd = [exe, world + ".world", W, H, spp, depth, fx, fy, fz, ax, ay, az, vfov, *SUN, out]
    r = subprocess.run([str(c) for c in cmd], cwd=ASSETS, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return _read_ppm(out)


def _our_image(dsrt, gpu_ctx, name, math_mode):
    world, cam_args, spp = CASES[name]
    hs = load_world(dsrt, world)
    W, H, depth = cam_args[3], cam_args[4], cam_args[5]
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth)
    gpu_ctx.upload(hs.view(cam, SUN))
    rgb, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, math_mode=math_mode))
    return rgb


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_kernel_and_this_kernel_give_the_same_bytes(dsrt, gpu_ctx, name, tmp_path):
    require_ref_binary(REF_GPU)
    ref = _reference_image(name, tmp_path)
    ours = _our_image(dsrt, gpu_ctx, name, 1)
    assert ref.shape == ours.shape
    lit = int((ref.max(axis=2) > 0).sum())
    assert lit > 150, "the reference's image should not be empty"
    differing = int((ref != ours).any(axis=2).sum())
    assert differing == 0, f"{name}: {differing} of {ref.shape[0] * ref.shape[1]} pixels differ from the reference kernel's image ({lit} lit)"


@pytest.mark.parametrize("tris,W,H,spp,frames", [(100000, 640, 360, 32, (98, 60)), (1000000, 1920, 1080, 12, (98,))])
def test_reference_kernel_on_the_bench_mesh_pose_frames(dsrt, gpu_ctx, tmp_path, tris, W, H, spp, frames):
    """The bench's own kind of workload through both programs: the procedural station at 100,000 triangles, pose frames 98 (camera 36 m from the station, which
    fills the view) and 60 (715 m) of the reference's pose file, 640 x 360 at 32 samples; and THE BENCH'S MESH -- 1,000,308 triangles, a tree that needs 18 stack
    entries -- at the bench's size, frame 98, 12 samples; max_depth 50 -- every byte.  (The whole headline frame at 1000 samples: tools/reference_kernel_probe.py
    --spp 1000 --compare, profiles/r03/reference_kernel_hipified_headline_frame.json: 0 of 2,073,600 pixels differ.)"""
    require_ref_binary(REF_GPU)
    from dsrt_amd import meshgen
    from conftest import GOLDEN
    if tris == 1000000:                                       # the file bench.py and test_headline_mesh_rows_match_the_oracle use
        obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
        if not os.path.exists(obj):
            tmp_obj = obj + f".{os.getpid()}.tmp"
            meshgen.write_obj(meshgen.build_station(1000000), tmp_obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
            os.replace(tmp_obj, obj)
    else:
        obj = tmp_path / f"station_{tris}.obj"
        meshgen.generate(obj, tris)
    (tmp_path / "station.world").write_text(f"obj {obj}\n")
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    assert hs.stack_need > 8
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    depth = 50
    for frame in frames:
        fr = dsrt.pose_to_frame(poses[frame])
        cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
        ref_out = tmp_path / f"ref_{frame}.ppm"
        r = subprocess.run([REF_GPU, str(tmp_path / "station.world"), str(W), str(H), str(spp), str(depth), *cam_from, "0", "0", "0", "40", *sun, str(ref_out)],
                           cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        cam = dsrt.frame_camera(fr, 40.0, W, H, spp, depth)
        gpu_ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
        ours, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, math_mode=1))
        ref = _read_ppm(ref_out)
        lit = int((ref.max(axis=2) > 0).sum())
        assert lit > (0.08 * W * H if frame == 98 else 100), (frame, lit)
        differing = int((ref != ours).any(axis=2).sum())
        assert differing == 0, f"frame {frame}: {differing} of {W * H} pixels differ from the reference kernel's image ({lit} lit)"


def test_math_mode_0_differs_from_the_reference_kernel_only_statistically(dsrt, gpu_ctx, tmp_path):
    """The default mode (deterministic sin / cos / pow shared with the CPU oracle) against the reference kernel with the device math library: one differing ulp
    in cosf de-synchronises the rest of a pixel's random stream, so some pixels differ -- but the images are the same picture."""
    require_ref_binary(REF_GPU)
    name = "station_near"
    ref = _reference_image(name, tmp_path).astype(np.int32)
    ours = _our_image(dsrt, gpu_ctx, name, 0).astype(np.int32)
    assert abs(float(ref.mean()) - float(ours.mean())) < 0.6                  # mean level within 0.6 / 255
    assert ((ref > 0).any(axis=2) == (ours > 0).any(axis=2)).mean() > 0.995  # the same pixels are lit


def test_a_contracted_build_of_the_reference_is_the_same_picture_not_the_same_bytes(tmp_path):
    """What parity with a real CUDA run can look like.  nvcc contracts a * b + c into one fused operation by default; oracle/_ref/ref_gpu_fma is the reference's
    kernel compiled that way (-ffp-contract=fast, everything else as ref_gpu).  Against the strict build of the SAME source its image differs in some pixels -- one
    rounding moves a sample across a branch and the rest of that pixel's random stream follows -- while mean level and coverage agree: the reason the product states
    a bit-exact contract (the reference's operations, uncontracted) and calls the comparison with any contracted or other-libm build statistical."""
    require_ref_binary(REF_GPU)
    require_ref_binary(REF_GPU_FMA)
    name = "station_near"
    strict = _reference_image(name, tmp_path).astype(np.int32)
    fused = _reference_image(name, tmp_path, REF_GPU_FMA).astype(np.int32)
    differing = float((strict != fused).any(axis=2).mean())
    assert 0.0 < differing < 0.5, differing                                   # not the same bytes ...
    assert abs(float(strict.mean()) - float(fused.mean())) < 0.6             # ... the same picture
    assert ((strict > 0).any(axis=2) == (fused > 0).any(axis=2)).mean() > 0.995


def test_randomised_views_of_every_world_match_the_reference_kernel(dsrt, gpu_ctx, tmp_path):
    """A bounded fuzz of the executed pin: 30 random views over all six world files (spheres, lights with mixture sampling, metal, dielectric, textures, the
    3k-triangle station, the `quirks` mesh with its degenerate and duplicated faces), ragged image sizes, 1 to 24 samples, depths 1 to 50, cameras from inside
    the geometry to far outside, random un-normalised sun directions -- each rendered by the reference's own kernel (one ref_gpu process per view) and by this
    library in math_mode 1, byte for byte."""
    require_ref_binary(REF_GPU)
    import ref_gpu_jobs as J
    jobs = [dict(j, trial=int(j["key"].split("/")[1])) for j in J.fuzz_jobs()]       # the same 30 views the committed fixtures hold (tests/ref_gpu_jobs.py)
    cache = {}
    failures, lit_total = [], 0
    for j in jobs:
        ref_out = tmp_path / f"ref_{j['trial']}.ppm"
        cmd = [REF_GPU, j["world"] + ".world", j["W"], j["H"], j["spp"], j["depth"], *[repr(v) for v in j["from"]], *[repr(v) for v in j["at"]], repr(j["vfov"]),
               *[repr(v) for v in j["sun"]], ref_out]
        rr = subprocess.run([str(c) for c in cmd], cwd=ASSETS, capture_output=True, text=True, timeout=600)
        assert rr.returncode == 0, rr.stdout[-1500:] + rr.stderr[-1500:]
        ref = _read_ppm(ref_out)
        if j["world"] not in cache:
            cache[j["world"]] = load_world(dsrt, j["world"])
        cam = dsrt.camera_look_at(tuple(j["from"]), tuple(j["at"]), j["vfov"], j["W"], j["H"], j["spp"], j["depth"])
        gpu_ctx.upload(cache[j["world"]].view(cam, tuple(j["sun"])))
        ours, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(j["W"], j["H"], j["spp"], j["depth"], math_mode=1))
        lit_total += int((ref.max(axis=2) > 0).sum())
        bad = int((ref != ours).any(axis=2).sum())
        if bad:
            failures.append((j["trial"], j["world"], j["W"], j["H"], j["spp"], j["depth"], j["from"], bad))
    assert not failures, failures
    assert lit_total > 20000                                   # the views do see things


@pytest.mark.parametrize("name", ["station_near", "mixed", "lights"])
def test_reference_kernel_with_the_shared_math_header_equals_the_default_mode(dsrt, gpu_ctx, name, tmp_path):
    """oracle/_ref/ref_gpu_detmath -- the reference's kernel with its three libm calls mapped onto include/dsrt_detmath.h -- live, against this library's DEFAULT mode
    (math_mode 0, the benched one).  The committed form of this comparison, all 40 images: tests/test_gpu_reference_fixtures.py; the CPU oracle against the same
    images: tests/test_oracle_reference_fixtures.py."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_gpu_detmath")
    require_ref_binary(exe)
    ref = _reference_image(name, tmp_path, exe)
    ours = _our_image(dsrt, gpu_ctx, name, 0)
    differing = int((ref != ours).any(axis=2).sum())
    assert differing == 0 and int((ref.max(axis=2) > 0).sum()) > 150, (name, differing)


def test_the_drop_in_gpu_render_scene_writes_the_reference_s_file(dsrt, tmp_path):
    """The reference's three calls (src/main.cpp:405-428) through THIS library -- dsrt_build_gpu_scene, gpu_render_scene, dsrt_free_gpu_scene -- with DSRT_MATH_MODE=1,
    against the same three calls of the reference's own code (ref_gpu): image_gpu.ppm, the whole file, header included."""
    require_ref_binary(REF_GPU)
    import ctypes as C
    name = "mixed"
    world, cam_args, spp = CASES[name]
    (fx, fy, fz), (ax, ay, az), vfov, W, H, depth = cam_args
    ref_out = tmp_path / "reference_image_gpu.ppm"
    r = subprocess.run([str(c) for c in [REF_GPU, world + ".world", W, H, spp, depth, fx, fy, fz, ax, ay, az, vfov, *SUN, ref_out]], cwd=ASSETS, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    hs = load_world(dsrt, world)
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth)
    dev = dsrt.GPUScene()
    assert dsrt.lib.dsrt_build_gpu_scene(hs._h, C.byref(cam), (C.c_float * 3)(*SUN), C.byref(dev)) == 0, dsrt.lib.dsrt_last_error()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    os.environ["DSRT_MATH_MODE"] = "1"
    try:
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)
        ours = open("image_gpu.ppm", "rb").read()
    finally:
        os.environ.pop("DSRT_MATH_MODE", None)
        os.chdir(cwd)
        dsrt.lib.dsrt_free_gpu_scene(C.byref(dev))
    assert ours == open(ref_out, "rb").read()
