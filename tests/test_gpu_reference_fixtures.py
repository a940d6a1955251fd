"""The product library against images THE REFERENCE'S OWN KERNEL rendered -- from committed data, no reference binary needed.  Both math modes:

  math_mode 1  ==  tests/golden/ref_gpu_images.json          written by oracle/_ref/ref_gpu: the reference's complete renderer (src/gpu_render.cu:387-1108 through
                                                             hipify-perl + hipcc, -ffp-contract=off; oracle/Makefile), cosf / sinf / powf from the device math library;
  math_mode 0  ==  tests/golden/ref_gpu_detmath_images.json  written by oracle/_ref/ref_gpu_detmath: the same build with those three names mapped onto
  (the default,                                              include/dsrt_detmath.h (oracle/ref_gpu_detmath_prelude.h) -- the images the CPU oracle reproduces too
   the benched one)                                          (tests/test_oracle_reference_fixtures.py, CPU suite).

Both files were written on an MI355X by tests/golden/make_ref_gpu_fixtures.py and hold, for every job of tests/ref_gpu_jobs.py, the image's sha256, lit-pixel
count and one CRC32 per row.  A clean checkout + build() on a box that has never seen /root/reference therefore still checks ray_color /
scene_hit / bvh_hit_closest (:387-936) against reference-made data: every scene of the parity suite, 30 randomised views, the station on pose frames, and the whole
headline frame.  Also here: the math_mode 1 compilation's OTHER launch paths (batch launch, tile shards + de-interleave, rng_mode 1), which the fixture images pin
through the same records.  Everything goes through the C ABI.
"""
import json
import os

import numpy as np
import pytest

import ref_gpu_jobs as J
from conftest import load_world

pytestmark = pytest.mark.gpu


VARIANTS = {1: ("ref_gpu_images.json", "ref_gpu"), 0: ("ref_gpu_detmath_images.json", "ref_gpu_detmath")}


@pytest.fixture(scope="module", params=[1, 0], ids=["math_mode_1_vs_ref_gpu", "math_mode_0_vs_ref_gpu_detmath"])
def fixtures(request):
    name, exe = VARIANTS[request.param]
    path = os.path.join(J.GOLDEN, name)
    assert os.path.exists(path), f"tests/golden/{name} is missing: it is committed data (tests/golden/make_ref_gpu_fixtures.py makes it on a GPU box)"
    doc = json.load(open(path))
    assert doc["made_by"] == "tests/golden/make_ref_gpu_fixtures.py" and len(doc["entries"]) >= 36 and exe in doc["renderer"]
    doc["math_mode"], doc["exe"] = request.param, exe
    return doc


def _render_job(dsrt, gpu_ctx, cache, job, math_mode, **desc_kw):
    if job["world"] not in cache:
        cache[job["world"]] = load_world(dsrt, job["world"])
    cam = dsrt.camera_look_at(tuple(job["from"]), tuple(job["at"]), job["vfov"], job["W"], job["H"], job["spp"], job["depth"])
    gpu_ctx.upload(cache[job["world"]].view(cam, tuple(job["sun"])))
    rgb, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(job["W"], job["H"], job["spp"], job["depth"], math_mode=math_mode, **desc_kw))
    return rgb


def _assert_is(rgb, record, what):
    got = J.image_record(rgb)
    if got["sha256"] != record["sha256"]:
        rows = J.differing_rows(rgb, record)
        pytest.fail(f"{what}: differs from the reference kernel's image in {len(rows)} of {record['H']} rows (first: {rows[:8]}); lit {got['lit']} vs {record['lit']}")
    assert got["lit"] == record["lit"]


def test_the_fixture_file_is_what_the_job_list_describes(fixtures):
    """The inputs are part of the fixture: every job the list generates today is in the file with exactly these parameters (a changed generator or seed would silently
    compare different renders otherwise)."""
    for job in J.case_jobs() + J.fuzz_jobs():
        assert fixtures["entries"][job["key"]]["job"] == job, job["key"]
    lit = sum(e["image"]["lit"] for e in fixtures["entries"].values())
    assert lit > 1000000                                              # the images do see things (the 1080p frames alone have ~0.9 M lit pixels each)


def test_parity_scenes_and_randomised_views_equal_the_reference_kernels_images(dsrt, gpu_ctx, fixtures):
    cache, failures, lit = {}, [], 0
    for job in J.case_jobs() + J.fuzz_jobs():
        rec = fixtures["entries"][job["key"]]["image"]
        rgb = _render_job(dsrt, gpu_ctx, cache, job, fixtures["math_mode"])
        if J.image_record(rgb)["sha256"] != rec["sha256"]:
            failures.append((job["key"], len(J.differing_rows(rgb, rec))))
        lit += rec["lit"]
    assert not failures, failures
    assert lit > 30000


def test_committed_reference_images_pixel_by_pixel(dsrt, gpu_ctx, fixtures):
    """Two of the reference kernel's images are committed whole (tests/golden/ref_gpu_*.ppm): a failure here names pixels, not hashes."""
    cache = {}
    for job in J.case_jobs():
        name = job["key"].split("/")[1]
        ppm = os.path.join(J.GOLDEN, f"{fixtures['exe']}_{name}.ppm")
        if not os.path.exists(ppm):
            continue
        ref = J.read_ppm(ppm)
        assert J.image_record(ref)["sha256"] == fixtures["entries"][job["key"]]["image"]["sha256"]
        ours = _render_job(dsrt, gpu_ctx, cache, job, fixtures["math_mode"])
        bad = np.argwhere((ours != ref).any(axis=2))
        assert len(bad) == 0, (name, len(bad), bad[:5].tolist())
        cache["seen"] = cache.get("seen", 0) + 1
    assert cache.get("seen", 0) >= 2


@pytest.mark.parametrize("tris,W,H,spp,frames", J.STATION_JOBS)
def test_station_pose_frames_equal_the_reference_kernels_images(dsrt, gpu_ctx, fixtures, tmp_path, tris, W, H, spp, frames):
    """The bench's kind of workload: the procedural station on frames of the reference's pose file, at 100 k triangles and at the bench's 1 M (a tree that needs 18 stack
    entries), the last one being THE HEADLINE FRAME, 1920 x 1080 x 1000 samples x depth 50 -- every byte of it against the reference kernel's."""
    obj = J.station_obj(tris, tmp_path)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(J.POSES)
    for frame in frames:
        entry = fixtures["entries"][J.station_key(tris, W, H, spp, frame)]
        assert entry["job"]["obj_sha256"] == J.file_sha256(obj), "the mesh generator's output changed: regenerate the fixtures (tests/golden/make_ref_gpu_fixtures.py)"
        fr = dsrt.pose_to_frame(poses[frame])
        assert [float(v) for v in fr.cam_in_model] == entry["job"]["from"] and [float(v) for v in fr.sun_dir_model] == entry["job"]["sun"]
        cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
        gpu_ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
        rgb, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, math_mode=fixtures["math_mode"]))
        _assert_is(rgb, entry["image"], f"station {tris}, frame {frame}, {W}x{H}x{spp}")


def test_both_math_modes_through_shards_batch_and_rng_mode_1(dsrt, gpu_ctx, fixtures):
    """Each compilation of the kernels has launch paths of its own (device_api.hip: devlibm::launch_render_batch, launch_resolve, sharded launches).  In rng_mode 0 each must give the
    REFERENCE KERNEL's image (the fixture); in rng_mode 1 -- which has no reference counterpart -- each must give what a single whole-frame launch gives in the same modes."""
    import torch
    cache = {}
    mm = fixtures["math_mode"]
    stream = torch.cuda.current_stream().cuda_stream
    for key in ("case/station_near", "case/mixed", "case/lights"):
        job = next(j for j in J.case_jobs() if j["key"] == key)
        rec = fixtures["entries"][key]["image"]
        W, H, spp, depth = job["W"], job["H"], job["spp"], job["depth"]
        whole = _render_job(dsrt, gpu_ctx, cache, job, mm)                   # (also uploads the scene and sets the camera)
        _assert_is(whole, rec, key)
        cam = dsrt.camera_look_at(tuple(job["from"]), tuple(job["at"]), job["vfov"], W, H, spp, depth)
        sun = tuple(job["sun"])
        for rng_mode in (0, 1):
            if rng_mode == 1:
                whole, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, math_mode=mm, rng_mode=1))
                again, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, math_mode=mm, rng_mode=1))
                assert np.array_equal(whole, again) and whole.max() > 0
            # three tile shards + de-interleave
            world = 3
            lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, shard_count=world))
            gathered = torch.zeros(world * lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
            for rank in range(world):
                part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
                gpu_ctx.render(dsrt.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=world, math_mode=mm, rng_mode=rng_mode), part.data_ptr(), stream=stream)
            image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
            gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            sharded = image.cpu().numpy().reshape(H, W, 3)
            assert np.array_equal(sharded, whole), (key, rng_mode, "3 shards")
            # the same view twice and a second view as ONE batch launch
            cam2 = dsrt.camera_look_at(tuple(np.float32(job["from"]) * np.float32(1.25)), tuple(job["at"]), job["vfov"], W, H, spp, depth)
            rgb = torch.zeros(3 * H * W * 3, dtype=torch.uint8, device="cuda")
            gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth, math_mode=mm, rng_mode=rng_mode), [cam, cam2, cam], [sun] * 3, rgb.data_ptr(), stream=stream, want_stats=True)
            got = rgb.cpu().numpy().reshape(3, H, W, 3)
            assert np.array_equal(got[0], whole) and np.array_equal(got[2], whole), (key, rng_mode, "batch")
            gpu_ctx.set_camera_sun(cam2, sun)
            alone2, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, math_mode=mm, rng_mode=rng_mode))
            assert np.array_equal(got[1], alone2), (key, rng_mode, "batch, second view")
            gpu_ctx.set_camera_sun(cam, sun)
            if rng_mode == 0:
                _assert_is(sharded, rec, key + " (3 shards)")
                _assert_is(got[0], rec, key + " (batch launch)")
