"""The reference's own main(), unmodified, linked against this library: the drop-in at LINK level.

oracle/_ref/ref_main_on_dsrt is the reference's src/main.cpp and src/stb_image_impl.cpp compiled from where they lie (g++, the reference's own headers)
together with ONE file of this repository, integration/reference_entry_points.cpp, which provides the two C++ entry points main() calls --
build_gpu_scene(const hittable_list&, const camera&, const vec3&) and free_gpu_scene(GPUScene&), inc/gpu_scene_builder.h:72-73 -- on top of the library's C
ABI; gpu_render_scene is the library's own export.  Neither src/gpu_scene_builder.cpp nor src/gpu_render.cu is in the program.  So what runs here is the
reference's argument parsing, pose reader, double-precision world -> model transform, OBJ / MTL loader classes and camera (src/main.cpp:139-431) feeding this
library's flattening, BVH build, upload and render kernel, and writing frame_%04zu.ppm exactly where the reference writes them.

main() hard-codes its mesh path (../../iss_model/ISS_stationary.obj relative to the working directory, src/main.cpp:238) and its frame size (800 x 450 at
1000 samples, max_depth 50, src/main.cpp:254-258): the test builds that directory layout in a scratch directory around a procedural station.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, require_ref_binary
from test_gpu_reference_kernel import REF_GPU, _read_ppm

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "ref_main_on_dsrt")
W, H, SPP, DEPTH = 800, 450, 1000, 50                       # src/main.cpp:254-258
FRAMES = (0, 60, 98)                                        # 1787 m, 715 m and 36 m from the station


def _station(obj):
    from dsrt_amd import meshgen
    meshgen.generate(obj, 60000)


def _run_main(tmp_path, pose_lines, math_mode, write_mesh=_station):
    top = tmp_path / "run"
    (top / "iss_model").mkdir(parents=True)
    cwd = top / "build" / "bin"
    cwd.mkdir(parents=True)
    obj = top / "iss_model" / "ISS_stationary.obj"
    write_mesh(obj)
    (top / "poses.txt").write_text("# three poses of the reference's rendezvous file\n" + "".join(pose_lines))
    env = dict(os.environ)
    env.pop("DSRT_MATH_MODE", None)
    if math_mode:
        env["DSRT_MATH_MODE"] = "1"
    r = subprocess.run([REF_MAIN, "--input_txt", str(top / "poses.txt"), "--output_dir", str(top / "out")], cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"Loaded {len(pose_lines)} poses." in r.stdout, r.stdout[-2000:]
    return obj, top / "out", r.stdout


def _pose_lines():
    rows = [l for l in open(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt")) if l.strip() and not l.lstrip().startswith("#")]
    return [rows[k] for k in FRAMES]


def test_the_reference_main_renders_its_frames_through_this_library(dsrt, gpu_ctx, tmp_path):
    require_ref_binary(REF_MAIN)
    obj, out, log = _run_main(tmp_path, _pose_lines(), math_mode=0)
    hs = dsrt.HostScene().add_obj(obj)                       # this library's own loader, flattener and builder on the same file
    hs.build_bvh()
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    lit_total = 0
    for i, k in enumerate(FRAMES):
        theirs = _read_ppm(out / f"frame_{i:04d}.ppm")
        assert theirs.shape == (H, W, 3)
        fr = dsrt.pose_to_frame(poses[k])
        gpu_ctx.upload(hs.view(dsrt.frame_camera(fr, 40.0, W, H, SPP, DEPTH), tuple(fr.sun_dir_model)))
        ours, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, SPP, DEPTH))
        differing = int((theirs != ours).any(axis=2).sum())
        assert differing == 0, f"pose {k}: {differing} of {W * H} pixels differ between the reference main()'s frame and this library's own pipeline"
        lit_total += int((theirs.max(axis=2) > 0).sum())
    assert lit_total > 0.05 * W * H, lit_total              # the near frame fills a good part of the view
    assert "GPU scene build time" in log and "Saved" in log  # main()'s own progress lines: it ran its whole loop


def test_the_reference_main_on_this_library_writes_what_the_reference_program_writes(tmp_path):
    """With the device math library selected (DSRT_MATH_MODE=1) the frame main() writes through this library is, byte for byte, the frame the reference's whole
    program (its own builder and its own kernel, oracle/_ref/ref_gpu) writes for the same mesh, camera and sun."""
    require_ref_binary(REF_MAIN)
    require_ref_binary(REF_GPU)
    import dsrt_amd as d
    lines = _pose_lines()[2:]                                # the near pose
    obj, out, _ = _run_main(tmp_path, lines, math_mode=1)
    theirs = _read_ppm(out / "frame_0000.ppm")
    fr = d.pose_to_frame(d.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))[FRAMES[2]])
    (tmp_path / "station.world").write_text(f"obj {obj}\n")
    ref_out = tmp_path / "ref_gpu.ppm"
    cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
    r = subprocess.run([REF_GPU, str(tmp_path / "station.world"), str(W), str(H), str(SPP), str(DEPTH), *cam_from, "0", "0", "0", "40", *sun, str(ref_out)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    ref = _read_ppm(ref_out)
    assert int((ref.max(axis=2) > 0).sum()) > 0.05 * W * H
    differing = int((ref != theirs).any(axis=2).sum())
    assert differing == 0, f"{differing} of {W * H} pixels differ"


def _textured_panels(obj):
    """Six 9 m panels tilted towards the light and the camera, one per kind of MTL entry the reference's loader tells apart (inc/triangle_mesh.h:75-112): a PPM map,
    a PNG map, a map that does not exist (the builder's white texel), a plain diffuse, a metal (Ks, Ns -> fuzz) and a dielectric (d, Ni); every panel a fan of two
    triangles with texture coordinates."""
    import shutil
    from conftest import ASSETS
    for name in ("checker.ppm", "stripes.png"):
        shutil.copy(os.path.join(ASSETS, name), obj.parent / name)
    (obj.parent / "panels.mtl").write_text(
        "newmtl checker\nKd 0.5 0.5 0.5\nKs 0.9 0.9 0.9\nmap_Kd checker.ppm\n"
        "newmtl stripes\nKd 0.2 0.3 0.4\nmap_Kd stripes.png\n"
        "newmtl missing_tex\nKd 0.3 0.3 0.3\nmap_Kd does_not_exist.png\n"
        "newmtl plain\nKd 0.6 0.4 0.2\n"
        "newmtl steel\nKd 0.1 0.1 0.1\nKs 0.8 0.8 0.7\nNs 300\n"
        "newmtl glass\nKd 0.9 0.9 0.9\nd 0.5\nNi 1.45\n")
    ex, ey = np.array([1.0, 0.0, 0.0]), np.array([0.0, -0.2873, 0.9578])          # the panels' plane: normal (0, 0.958, 0.287)
    lines = ["mtllib panels.mtl", "vt 0 0", "vt 3 0", "vt 3 2", "vt 0 2"]
    for k, mat in enumerate(("checker", "stripes", "missing_tex", "plain", "steel", "glass")):
        c = ex * (10.0 * (k % 3) - 10.0) + ey * (10.0 * (k // 3) - 5.0)
        for a, b in ((-4.5, -4.5), (4.5, -4.5), (4.5, 4.5), (-4.5, 4.5)):
            p = c + ex * a + ey * b
            lines.append(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}")
        lines.append(f"usemtl {mat}")
        lines.append("f " + " ".join(f"{4 * k + i}/{i}" for i in (1, 2, 3, 4)))
    obj.write_text("\n".join(lines) + "\n")


def test_textured_and_metal_materials_through_the_reference_loader_and_main(dsrt, gpu_ctx, tmp_path):
    """The reference's OBJ / MTL loader (Ks -> metal, map_Kd -> per-triangle texture path, uv stored as (u, 1 - v)) and its stb flip flag on one side, this library's
    loader on the other, same files: the frames must be the same bytes.  Exercises dsrt_host_scene_add_texture_file behind build_gpu_scene."""
    require_ref_binary(REF_MAIN)
    obj, out, log = _run_main(tmp_path, _pose_lines()[2:], math_mode=0, write_mesh=_textured_panels)
    theirs = _read_ppm(out / "frame_0000.ppm")
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    a = hs.arrays()
    assert len(a["tris"]) == 12 and len(a["texhdr"]) == 3 and sorted(set(int(t) for t in a["mats"]["type"])) == [0, 1, 2]      # lambertian, metal, dielectric
    fr = dsrt.pose_to_frame(dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))[FRAMES[2]])
    gpu_ctx.upload(hs.view(dsrt.frame_camera(fr, 40.0, W, H, SPP, DEPTH), tuple(fr.sun_dir_model)))
    ours, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, SPP, DEPTH))
    lit = theirs.max(axis=2) > 0
    assert int(lit.sum()) > 0.05 * W * H, int(lit.sum())
    assert len(np.unique(theirs[lit].reshape(-1, 3), axis=0)) > 8          # texels, not one flat colour
    differing = int((theirs != ours).any(axis=2).sum())
    assert differing == 0, f"{differing} of {W * H} pixels differ"
