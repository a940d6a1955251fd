"""The certified second tree (include/dsrt.h: dsrt_ctx_set_certified_tree): rays walk a binned-SAH tree, the kernel certifies every answer against the reference's
median-split tree -- not unreachable there, no exact tie, inside its reference-leaf box's slab interval -- and re-walks the reference tree when it cannot.  The claim is
that the image is THE REFERENCE'S, byte for byte; so every comparison here is with the reference kernel's own images (tests/golden/ref_gpu*_images.json, both math
modes), with the CPU oracle, and with this library's plain reference walk -- bytes, float bit patterns, never a tolerance.  Covered on purpose: the `mixed` scene,
which has a panel the reference's tree can never reach (zero-thickness leaf); the `quirks` mesh with duplicated faces (exact ties -> the fallback path); spheres
beside a mesh; a scene without any mesh; ragged sizes; tile shards; batch launches; the 1 M-triangle bench mesh; the whole headline frame.
"""
import json
import os

import numpy as np
import pytest

import ref_gpu_jobs as J
from conftest import load_world
from test_oracle import CASES, SUN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cert_ctx(dsrt):
    if dsrt.lib.dsrt_device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    ctx = dsrt.Context(0).set_certified_tree(True)
    yield ctx
    ctx.close()


def _fixtures(math_mode):
    doc = json.load(open(os.path.join(J.GOLDEN, "ref_gpu_images.json" if math_mode == 1 else "ref_gpu_detmath_images.json")))
    return doc["entries"]


def _job_image(dsrt, ctx, cache, job, **kw):
    if job["world"] not in cache:
        cache[job["world"]] = load_world(dsrt, job["world"])
    cam = dsrt.camera_look_at(tuple(job["from"]), tuple(job["at"]), job["vfov"], job["W"], job["H"], job["spp"], job["depth"])
    ctx.upload(cache[job["world"]].view(cam, tuple(job["sun"])))
    return ctx.render_to_host(dsrt.make_desc(job["W"], job["H"], job["spp"], job["depth"], **kw), want_f32=True)


@pytest.mark.parametrize("name", sorted(CASES))
def test_parity_scenes_on_the_certified_tree_equal_the_oracle_and_the_plain_walk(dsrt, cert_ctx, oracle, name):
    world, cam_args, spp = CASES[name]
    hs = load_world(dsrt, world)
    W, H, depth = cam_args[3], cam_args[4], cam_args[5]
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth)
    scene = hs.view(cam, SUN)
    want_rgb, want_f32, _ = oracle.render(scene, W, H)
    cert_ctx.upload(scene)
    has_mesh = scene.num_triangles > 0 and scene.num_bvh_nodes > 0
    assert cert_ctx.has_certified_tree == has_mesh
    for kw in ({}, {"checked": 1}, {"collect_counters": 1}, {"rng_mode": 0, "tune": (0, 0, 0, 4)}):
        rgb, f32, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, **kw), want_f32=True)
        assert st.certified_tree_used == (1 if has_mesh else 0), kw
        assert np.array_equal(rgb, want_rgb), f"{name} {kw}: {(rgb != want_rgb).any(axis=2).sum()} pixels differ from the oracle"
        assert np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32)), (name, kw)
    plain, plain32, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, tune=(0, 0, 0, 64)), want_f32=True)        # DSRT_TUNE_REFERENCE_WALK
    assert st.certified_tree_used == 0
    assert np.array_equal(plain, want_rgb) and np.array_equal(plain32.view(np.uint32), want_f32.view(np.uint32))


@pytest.mark.parametrize("math_mode", [0, 1])
def test_randomised_views_on_the_certified_tree_equal_the_reference_kernels_images(dsrt, cert_ctx, math_mode):
    """The 30 randomised views and the six parity scenes against the images the reference's own kernel rendered -- committed data (tests/golden/ref_gpu*_images.json)."""
    entries, cache, failures, fallbacks = _fixtures(math_mode), {}, [], 0
    for job in J.case_jobs() + J.fuzz_jobs():
        rgb, _, st = _job_image(dsrt, cert_ctx, cache, job, math_mode=math_mode)
        if J.image_record(rgb)["sha256"] != entries[job["key"]]["image"]["sha256"]:
            failures.append((job["key"], len(J.differing_rows(rgb, entries[job["key"]]["image"]))))
        if job["world"] == "quirks":                         # duplicated faces: exact ties, so the certificate must refuse some answers -- and the image still be right
            _, _, sc = _job_image(dsrt, cert_ctx, cache, job, math_mode=math_mode, collect_counters=1)
            fallbacks += sc.certificate_fallbacks
    assert not failures, failures
    assert fallbacks > 0, "the quirks mesh has duplicated faces: some ray must have failed its certificate and been walked again on the reference tree"


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("tris,W,H,spp,frames", J.STATION_JOBS)
def test_station_pose_frames_on_the_certified_tree_equal_the_reference_kernels_images(dsrt, cert_ctx, tmp_path, math_mode, tris, W, H, spp, frames):
    """The bench's workload: the station at 100 k and 1 M triangles on pose frames, the last job being THE HEADLINE FRAME (1920 x 1080 x 1000 x depth 50): every byte
    against the reference kernel's image of the same frame, in both math modes."""
    entries = _fixtures(math_mode)
    obj = J.station_obj(tris, tmp_path)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(J.POSES)
    for frame in frames:
        entry = entries[J.station_key(tris, W, H, spp, frame)]
        assert entry["job"]["obj_sha256"] == J.file_sha256(obj)
        fr = dsrt.pose_to_frame(poses[frame])
        cert_ctx.upload(hs.view(dsrt.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model)))
        assert cert_ctx.has_certified_tree
        rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, math_mode=math_mode))
        assert st.certified_tree_used == 1
        got = J.image_record(rgb)
        assert got["sha256"] == entry["image"]["sha256"], \
            f"station {tris} frame {frame} {W}x{H}x{spp} math_mode {math_mode}: {len(J.differing_rows(rgb, entry['image']))} rows differ from the reference kernel's image"


def test_certified_tree_in_shards_batches_far_cameras_and_rng_mode_1(dsrt, cert_ctx, oracle, tmp_path):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    obj = J.station_obj(100000, tmp_path)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(J.POSES)
    W, H, spp, depth = 150, 85, 16, 50
    frames = [0, 60, 90, 98]

    def cam_sun(i):
        fr = dsrt.pose_to_frame(poses[i])
        return dsrt.frame_camera(fr, 40.0, W, H, spp, depth), tuple(fr.sun_dir_model)
    cams, suns = zip(*[cam_sun(i) for i in frames])
    cert_ctx.upload(hs.view(cams[0], suns[0]))
    want = [oracle.render(hs.view(c, s), W, H)[0] for c, s in zip(cams, suns)]
    used = []
    for k in range(len(frames)):
        cert_ctx.set_camera_sun(cams[k], suns[k])
        rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth))
        used.append(st.certified_tree_used)
        assert np.array_equal(rgb, want[k]), frames[k]
        # three tile shards + de-interleave
        world = 3
        lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, shard_count=world))
        gathered = torch.zeros(world * lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
        for rank in range(world):
            part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
            cert_ctx.render(dsrt.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=world), part.data_ptr(), stream=stream)
        image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
        cert_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), want[k]), (frames[k], "shards")
    # pose 0 is 1787 m from a 109 m station: within 30 extents, so the second tree is used on every frame of this list; a camera 100 extents away is not
    assert used == [1, 1, 1, 1], used
    far = dsrt.camera_look_at((0.0, 0.0, 20000.0), (0.0, 0.0, 0.0), 1.0, W, H, spp, depth)
    cert_ctx.set_camera_sun(far, suns[0])
    rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth))
    assert st.certified_tree_used == 0
    assert np.array_equal(rgb, oracle.render(hs.view(far, suns[0]), W, H)[0])
    # all four poses as ONE batch launch
    n = len(frames)
    out = torch.zeros(n * H * W * 3, dtype=torch.uint8, device="cuda")
    st = cert_ctx.render_batch(dsrt.make_desc(W, H, spp, depth), list(cams), list(suns), out.data_ptr(), stream=stream, want_stats=True)
    assert st.certified_tree_used == 1
    got = out.cpu().numpy().reshape(n, H, W, 3)
    for k in range(n):
        assert np.array_equal(got[k], want[k]), (frames[k], "batch")
    # rng_mode 1 has no reference counterpart: on the certified tree it must be what the plain walk gives in the same mode (every BVH answer is the same)
    cert_ctx.set_camera_sun(cams[3], suns[3])
    a, _, _ = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
    b, _, _ = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1, tune=(0, 0, 0, 64)))
    assert np.array_equal(a, b) and a.max() > 0


def test_certificate_audit_every_answer_of_the_second_tree_against_the_reference_walk(dsrt, cert_ctx, oracle, tmp_path):
    """collect_counters = 3: the counting build walks EVERY answer of the second tree -- certified hits and misses -- on the reference tree as well and compares them ray by
    ray (triangle, bit patterns of t, u, v; blocked-or-not for any-hit shadow rays).  The certificate's claim, checked directly: audited > 0, mismatches == 0, on the parity
    scenes, the quirks mesh, the 100 k station and the bench's 1 M-triangle mesh (1080p x 8 samples: 30 M rays).  Then the negative control: with one box of the second tree
    made empty (the test hook overwrites a word of its root record) the second tree loses geometry -- the audit must SEE that (mismatches > 0), and because the audit uses
    the reference walk's answers the image is still the oracle's."""
    import struct
    total_audited = 0
    for name in ("station_near", "mixed", "textured"):
        world, cam_args, spp = CASES[name]
        hs = load_world(dsrt, world)
        W, H, depth = cam_args[3], cam_args[4], cam_args[5]
        scene = hs.view(dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth), SUN)
        want, _, _ = oracle.render(scene, W, H)
        cert_ctx.upload(scene)
        rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=3))
        assert st.certified_tree_used == 1 and st.certificate_audited > 1000 and st.certificate_audit_mismatches == 0, (name, st.certificate_audited, st.certificate_audit_mismatches)
        assert np.array_equal(rgb, want), name
        total_audited += st.certificate_audited
    quirks = next(j for j in J.fuzz_jobs() if j["world"] == "quirks" and j["W"] * j["H"] > 2000)
    _, _, st = _job_image(dsrt, cert_ctx, {}, quirks, collect_counters=3)
    assert st.certificate_audit_mismatches == 0 and st.certificate_audited > 0
    poses = dsrt.read_pose_file(J.POSES)
    fr = dsrt.pose_to_frame(poses[98])
    for tris, W, H, spp in ((100000, 640, 360, 8), (1000000, 1920, 1080, 8)):
        hs = dsrt.HostScene().add_obj(J.station_obj(tris, tmp_path))
        hs.build_bvh()
        scene = hs.view(dsrt.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model))
        cert_ctx.upload(scene)
        rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, collect_counters=3))
        plain, _, _ = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, tune=(0, 0, 0, 64)))
        assert st.certificate_audit_mismatches == 0, (tris, st.certificate_audited, st.certificate_audit_mismatches)
        assert st.certificate_audited >= st.rays - st.certificate_fallbacks - 8 and st.certificate_audited > 0.9 * W * H * spp
        assert np.array_equal(rgb, plain)
        total_audited += st.certificate_audited
    assert total_audited > 25_000_000
    # negative control on the 1 M mesh (still resident): the left child box of the second tree's root made empty -> geometry lost on the second tree -> the audit reports it
    old = cert_ctx.poke_node_word(0, struct.unpack("<I", struct.pack("<f", 1.0e30))[0])            # record 0 = the second tree's root; word 0 = L.lo.x
    try:
        rgb2, _, st2 = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, collect_counters=3))
    finally:
        cert_ctx.poke_node_word(0, old)
    assert st2.certificate_audit_mismatches > 1000, st2.certificate_audit_mismatches
    assert np.array_equal(rgb2, plain), "the audit renders with the reference walk's answers: a broken second tree must not change the image"


def test_certified_tree_through_the_environment_reaches_the_drop_in_and_the_multi_gpu_host(dsrt, oracle, tmp_path, monkeypatch):
    """DSRT_CERTIFIED_TREE=1: contexts the library creates ITSELF -- the drop-in gpu_render_scene's and dsrt_multi_*'s -- build and use the second tree.  The drop-in's
    file must be the plain run's file, byte for byte (header included); the multi-GPU host's frame (two ranks on the one GPU there is) the oracle's image."""
    import ctypes as C
    name = "station_near"
    world, cam_args, spp = CASES[name]
    hs = load_world(dsrt, world)
    W, H, depth = cam_args[3], cam_args[4], cam_args[5]
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth)
    want, _, _ = oracle.render(hs.view(cam, SUN), W, H)
    monkeypatch.setenv("DSRT_CERTIFIED_TREE", "1")
    ctx = dsrt.Context(0)                                   # (created under the variable: the option is on without the call)
    ctx.upload(hs.view(cam, SUN))
    assert ctx.has_certified_tree
    rgb, _, st = ctx.render_to_host(dsrt.make_desc(W, H, spp, depth))
    assert st.certified_tree_used == 1 and np.array_equal(rgb, want)
    ctx.close()
    m = dsrt.Multi([0, 0])
    m.upload(hs.view(cam, SUN))
    img, _, _ = m.render_frame(dsrt.make_desc(W, H, spp, depth), cam, SUN)
    assert np.array_equal(np.asarray(img).reshape(H, W, 3), want)
    m.close()
    # the drop-in entry point: its own context is created on first use
    dev = dsrt.GPUScene()
    assert dsrt.lib.dsrt_build_gpu_scene(hs._h, C.byref(cam), (C.c_float * 3)(*SUN), C.byref(dev)) == 0, dsrt.lib.dsrt_last_error()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)
        data = open("image_gpu.ppm", "rb").read()
        assert dsrt.lib.dsrt_dropin_has_certified_tree() == 1
        monkeypatch.delenv("DSRT_CERTIFIED_TREE")
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)           # the variable gone: the same scene is converted again, without the second tree
        assert dsrt.lib.dsrt_dropin_has_certified_tree() == 0 and open("image_gpu.ppm", "rb").read() == data
    finally:
        os.chdir(cwd)
        dsrt.lib.dsrt_free_gpu_scene(C.byref(dev))
    assert data == b"P6\n%d %d\n255\n" % (W, H) + want.tobytes()


@pytest.mark.parametrize("scale", [1.0e-15, 1.0 / 64.0, 4096.0, 1.0e9])
def test_scaled_scenes_on_the_certified_tree_match_the_oracle_and_pass_the_audit(dsrt, cert_ctx, oracle, scale):
    """The widening of the second tree's boxes and the relaxation of its distance culling are RELATIVE (2^-16 of the scene's extent, 2^-10 of t): the station from 1e-15 to
    1e9 times its size, camera moved with it -- image bits equal the oracle's, and the audit finds no answer of the second tree that differs from the reference walk's."""
    from conftest import ASSETS
    hs = dsrt.HostScene().add_obj(os.path.join(ASSETS, "station_3k.obj"), scale=scale)
    hs.build_bvh()
    W, H, spp, depth = 160, 90, 8, 50
    cam = dsrt.camera_look_at(tuple(c * scale for c in (12.0, 9.0, 38.0)), (0.0, 0.0, 0.0), 40.0, W, H, spp, depth)
    scene = hs.view(cam, SUN)
    want_rgb, want_f32, _ = oracle.render(scene, W, H)
    cert_ctx.upload(scene)
    assert cert_ctx.has_certified_tree
    rgb, f32, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth), want_f32=True)
    assert st.certified_tree_used == 1
    assert np.array_equal(rgb, want_rgb) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32))
    rgb, _, st = cert_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=3))
    assert np.array_equal(rgb, want_rgb) and st.certificate_audit_mismatches == 0 and st.certificate_audited > 0
