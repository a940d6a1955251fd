"""The reference's OWN render kernel, executed -- the pin nothing else in this pipeline could give.

oracle/_ref/ref_gpu is the reference's complete renderer (its OBJ / world loader classes, build_gpu_scene, gpu_render_scene -> render_kernel ->
ray_color -> scene_hit -> bvh_hit_closest, src/gpu_render.cu:387-1108), built by oracle/Makefile from the sources where they lie: the three files
that name the CUDA runtime are translated to HIP by the image's own hipify-perl in a scratch directory and compiled with hipcc for gfx950 (no line
of arithmetic or control flow is touched; oracle/ref_gpu_driver.cpp has the details).  What such a build cannot share with the product is the
three library functions the product replaces by include/dsrt_detmath.h (sinf, cosf, powf: DESIGN.md section 2); so the product's kernel is built
once more with exactly those three taken from the device math library (oracle/_ref/libdsrt_hip_devlibm.so, -DDSRT_DEVICE_LIBM) -- and then the two
programs compute, operation for operation, the same thing, and their images must be equal BYTE FOR BYTE: every scene of the parity suite, ties,
traversal order, Russian roulette, the mixture branch, dielectrics, textures and all.  The product build itself (detmath) equals the CPU oracle
bit for bit (tests/test_gpu_parity.py); the two builds differ in those three functions only.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ASSETS
from test_oracle import CASES, SUN

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "ref_gpu")
DEVLIBM = os.path.join(ROOT, "oracle", "_ref", "libdsrt_hip_devlibm.so")


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


def _our_image(name, tmp_path, lib=None):
    world, cam_args, spp = CASES[name]
    W, H = cam_args[3], cam_args[4]
    out = tmp_path / f"ours_{name}_{'devlibm' if lib else 'product'}.rgb"
    env = dict(os.environ)
    if lib:
        env["DSRT_LIB"] = lib
    else:
        env.pop("DSRT_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_render_case_worker.py"), name, str(out)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return np.frombuffer(open(out, "rb").read(), np.uint8).reshape(H, W, 3)


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_kernel_and_this_kernel_give_the_same_bytes(name, tmp_path):
    if not (os.path.exists(REF_GPU) and os.path.exists(DEVLIBM)):
        pytest.skip("oracle/_ref/ref_gpu or libdsrt_hip_devlibm.so not built (oracle/Makefile builds them where /root/reference and hipify-perl exist)")
    ref = _reference_image(name, tmp_path)
    ours = _our_image(name, tmp_path, DEVLIBM)
    assert ref.shape == ours.shape
    lit = int((ref.max(axis=2) > 0).sum())
    assert lit > 150, "the reference's image should not be empty"
    differing = int((ref != ours).any(axis=2).sum())
    assert differing == 0, f"{name}: {differing} of {ref.shape[0] * ref.shape[1]} pixels differ from the reference kernel's image ({lit} lit)"


@pytest.mark.parametrize("tris,W,H,spp,frames", [(100000, 640, 360, 32, (98, 60)), (1000000, 1920, 1080, 12, (98,))])
def test_reference_kernel_on_the_bench_mesh_pose_frames(tmp_path, tris, W, H, spp, frames):
    """The bench's own kind of workload through both programs: the procedural station at 100,000 triangles, pose frames 98 (camera 36 m from the station, which
    fills the view) and 60 (715 m) of the reference's pose file, 640 x 360 at 32 samples; and THE BENCH'S MESH -- 1,000,308 triangles, a tree that needs 18 stack
    entries -- at the bench's size, frame 98, 12 samples; max_depth 50 -- every byte.  (The whole headline frame at 1000 samples: tools/reference_kernel_probe.py
    --spp 1000 --compare, profiles/r03/reference_kernel_hipified_headline_frame.json: 0 of 2,073,600 pixels differ.)"""
    if not (os.path.exists(REF_GPU) and os.path.exists(DEVLIBM)):
        pytest.skip("oracle/_ref/ref_gpu or libdsrt_hip_devlibm.so not built")
    sys.path.insert(0, ROOT)
    import dsrt_amd as d                                      # (host-side helpers only: mesh writer, pose arithmetic; no GPU call in this process)
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
    poses = d.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    depth = 50
    worker = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "import dsrt_amd as d\n"
        "world, out = sys.argv[2], sys.argv[3]\n"
        "W, H, spp, depth = (int(v) for v in sys.argv[4:8])\n"
        "cam_from = tuple(float(v) for v in sys.argv[8:11]); sun = tuple(float(v) for v in sys.argv[11:14])\n"
        "hs = d.HostScene().add_world_file(world); hs.build_bvh()\n"
        "assert hs.stack_need > 8\n"
        "cam = d.camera_look_at(cam_from, (0.0, 0.0, 0.0), 40.0, W, H, spp, depth)\n"
        "ctx = d.Context(0); ctx.upload(hs.view(cam, sun))\n"
        "rgb, _, _ = ctx.render_to_host(d.make_desc(W, H, spp, depth))\n"
        "open(out, 'wb').write(rgb.tobytes())\n")
    for frame in frames:
        fr = d.pose_to_frame(poses[frame])
        cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
        ref_out, our_out = tmp_path / f"ref_{frame}.ppm", tmp_path / f"ours_{frame}.rgb"
        r = subprocess.run([REF_GPU, str(tmp_path / "station.world"), str(W), str(H), str(spp), str(depth), *cam_from, "0", "0", "0", "40", *sun, str(ref_out)],
                           cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        r = subprocess.run([sys.executable, "-c", worker, ROOT, str(tmp_path / "station.world"), str(our_out), str(W), str(H), str(spp), str(depth), *cam_from, *sun],
                           cwd=tmp_path, capture_output=True, text=True, timeout=900, env=dict(os.environ, DSRT_LIB=DEVLIBM))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        ref = _read_ppm(ref_out)
        ours = np.frombuffer(open(our_out, "rb").read(), np.uint8).reshape(H, W, 3)
        lit = int((ref.max(axis=2) > 0).sum())
        assert lit > (0.08 * W * H if frame == 98 else 100), (frame, lit)
        differing = int((ref != ours).any(axis=2).sum())
        assert differing == 0, f"frame {frame}: {differing} of {W * H} pixels differ from the reference kernel's image ({lit} lit)"


def test_the_product_build_differs_from_the_reference_kernel_only_statistically(tmp_path):
    """The product (deterministic sin / cos / pow shared with the CPU oracle) against the reference kernel with the device math library: one differing ulp
    in cosf de-synchronises the rest of a pixel's random stream, so some pixels differ -- but the images are the same picture."""
    if not os.path.exists(REF_GPU):
        pytest.skip("oracle/_ref/ref_gpu not built")
    name = "station_near"
    ref = _reference_image(name, tmp_path).astype(np.int32)
    ours = _our_image(name, tmp_path).astype(np.int32)
    assert abs(float(ref.mean()) - float(ours.mean())) < 0.6                  # mean level within 0.6 / 255
    assert ((ref > 0).any(axis=2) == (ours > 0).any(axis=2)).mean() > 0.995  # the same pixels are lit


def test_a_contracted_build_of_the_reference_is_the_same_picture_not_the_same_bytes(tmp_path):
    """What parity with a real CUDA run can look like.  nvcc contracts a * b + c into one fused operation by default; oracle/_ref/ref_gpu_fma is the reference's
    kernel compiled that way (-ffp-contract=fast, everything else as ref_gpu).  Against the strict build of the SAME source its image differs in some pixels -- one
    rounding moves a sample across a branch and the rest of that pixel's random stream follows -- while mean level and coverage agree: the reason the product states
    a bit-exact contract (the reference's operations, uncontracted) and calls the comparison with any contracted or other-libm build statistical."""
    if not (os.path.exists(REF_GPU) and os.path.exists(REF_GPU_FMA)):
        pytest.skip("oracle/_ref/ref_gpu or ref_gpu_fma not built")
    name = "station_near"
    strict = _reference_image(name, tmp_path).astype(np.int32)
    fused = _reference_image(name, tmp_path, REF_GPU_FMA).astype(np.int32)
    differing = float((strict != fused).any(axis=2).mean())
    assert 0.0 < differing < 0.5, differing                                   # not the same bytes ...
    assert abs(float(strict.mean()) - float(fused.mean())) < 0.6             # ... the same picture
    assert ((strict > 0).any(axis=2) == (fused > 0).any(axis=2)).mean() > 0.995


def test_randomised_views_of_every_world_match_the_reference_kernel(tmp_path):
    """A bounded fuzz of the executed pin: 30 random views over all six world files (spheres, lights with mixture sampling, metal, dielectric, textures, the
    3k-triangle station, the `quirks` mesh with its degenerate and duplicated faces), ragged image sizes, 1 to 24 samples, depths 1 to 50, cameras from inside
    the geometry to far outside, random un-normalised sun directions -- each rendered by the reference's own kernel (one ref_gpu process per view) and by this
    kernel's device-libm build (one worker process for all), byte for byte."""
    if not (os.path.exists(REF_GPU) and os.path.exists(DEVLIBM)):
        pytest.skip("oracle/_ref/ref_gpu or libdsrt_hip_devlibm.so not built")
    import json
    rng = np.random.default_rng(20251005)
    worlds = ("c1_spheres", "lights", "station_3k", "textured", "mixed", "quirks")
    f32 = lambda v: float(np.float32(v))                      # noqa: E731 -- every number crosses both command lines as an exactly representable float
    jobs = []
    for trial in range(30):
        world = worlds[trial % len(worlds)]
        W, H = int(rng.integers(8, 150)), int(rng.integers(6, 100))
        spp = int(rng.choice([1, 2, 5, 9, 24]))
        depth = int(rng.choice([1, 2, 5, 12, 50]))
        dist = float(rng.choice([0.5, 3.0, 9.0, 30.0, 120.0, 600.0])) * (0.15 if world in ("c1_spheres", "lights", "textured", "mixed") else 1.0)
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        jobs.append({"trial": trial, "world": world, "W": W, "H": H, "spp": spp, "depth": depth, "from": [f32(v) for v in direction * dist + (0.0, 1.0, 0.0)],
                     "at": [f32(v) for v in rng.normal(size=3) * (0.0 if trial % 3 else 0.5)], "vfov": f32(rng.choice([20.0, 40.0, 75.0])),
                     "sun": [f32(v) for v in rng.normal(size=3)], "out": str(tmp_path / f"ours_{trial}.rgb")})
    (tmp_path / "jobs.json").write_text(json.dumps(jobs))
    worker = (
        "import sys, os, json\n"
        "root, assets, jobs = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))\n"
        "sys.path.insert(0, root)\n"
        "import dsrt_amd as d\n"
        "os.chdir(assets)\n"
        "ctx, cache = d.Context(0), {}\n"
        "for j in jobs:\n"
        "    if j['world'] not in cache:\n"
        "        hs = d.HostScene().add_world_file(j['world'] + '.world'); hs.build_bvh(); cache[j['world']] = hs\n"
        "    hs = cache[j['world']]\n"
        "    cam = d.camera_look_at(tuple(j['from']), tuple(j['at']), j['vfov'], j['W'], j['H'], j['spp'], j['depth'])\n"
        "    ctx.upload(hs.view(cam, tuple(j['sun'])))\n"
        "    rgb, _, _ = ctx.render_to_host(d.make_desc(j['W'], j['H'], j['spp'], j['depth']))\n"
        "    open(j['out'], 'wb').write(rgb.tobytes())\n")
    r = subprocess.run([sys.executable, "-c", worker, ROOT, ASSETS, str(tmp_path / "jobs.json")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DSRT_LIB=DEVLIBM))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    failures, lit_total = [], 0
    for j in jobs:
        ref_out = tmp_path / f"ref_{j['trial']}.ppm"
        cmd = [REF_GPU, j["world"] + ".world", j["W"], j["H"], j["spp"], j["depth"], *[repr(v) for v in j["from"]], *[repr(v) for v in j["at"]], repr(j["vfov"]),
               *[repr(v) for v in j["sun"]], ref_out]
        rr = subprocess.run([str(c) for c in cmd], cwd=ASSETS, capture_output=True, text=True, timeout=600)
        assert rr.returncode == 0, rr.stdout[-1500:] + rr.stderr[-1500:]
        ref = _read_ppm(ref_out)
        ours = np.frombuffer(open(j["out"], "rb").read(), np.uint8).reshape(j["H"], j["W"], 3)
        lit_total += int((ref.max(axis=2) > 0).sum())
        bad = int((ref != ours).any(axis=2).sum())
        if bad:
            failures.append((j["trial"], j["world"], j["W"], j["H"], j["spp"], j["depth"], j["from"], bad))
    assert not failures, failures
    assert lit_total > 20000                                   # the views do see things
