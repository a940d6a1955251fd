"""The CPU oracle against images THE REFERENCE'S OWN KERNEL rendered -- directly, in the CPU suite, from committed data.

tests/golden/ref_gpu_detmath_images.json was written on an MI355X by tests/golden/make_ref_gpu_fixtures.py from oracle/_ref/ref_gpu_detmath: the reference's
complete renderer (src/gpu_render.cu:387-1108 through hipify-perl + hipcc, -ffp-contract=off) with exactly one difference from a plain build -- the three libm
names its kernel calls (cosf, sinf :104-106, 157-158; powf :211, 1019-1021) resolve to include/dsrt_detmath.h, the deterministic versions this oracle uses
(oracle/ref_gpu_detmath_prelude.h: no libm exists on both a CPU and that GPU, and one differing ulp de-synchronises a pixel's random stream).  Every other
operation, the control flow of ray_color, the traversal order, ties and all are the reference's, executed.  So oracle/dsrt_oracle.c must reproduce these images
BYTE FOR BYTE -- one hop from the reference's loop to the oracle, no GPU and no product library in between.
"""
import json
import os

import numpy as np
import pytest

import ref_gpu_jobs as J
from conftest import load_world

FIXTURE = os.path.join(J.GOLDEN, "ref_gpu_detmath_images.json")


@pytest.fixture(scope="module")
def fixtures():
    assert os.path.exists(FIXTURE), "tests/golden/ref_gpu_detmath_images.json is missing: it is committed data (tests/golden/make_ref_gpu_fixtures.py makes it on a GPU box)"
    doc = json.load(open(FIXTURE))
    assert doc["made_by"] == "tests/golden/make_ref_gpu_fixtures.py" and "ref_gpu_detmath" in doc["renderer"]
    return doc


def _oracle_image(dsrt, oracle, cache, job):
    if job["world"] not in cache:
        cache[job["world"]] = load_world(dsrt, job["world"])
    cam = dsrt.camera_look_at(tuple(job["from"]), tuple(job["at"]), job["vfov"], job["W"], job["H"], job["spp"], job["depth"])
    scene = cache[job["world"]].view(cam, tuple(job["sun"]))
    scene.params.samples_per_pixel = job["spp"]
    rgb, _, _ = oracle.render(scene, job["W"], job["H"], want_f32=False)
    return rgb


def test_oracle_reproduces_the_reference_kernels_images(dsrt, oracle, fixtures):
    """All six parity scenes (spheres, lights + mixture sampling, metal, dielectric, textures, the station far and near) and 30 randomised views over every world file."""
    cache, failures, lit = {}, [], 0
    for job in J.case_jobs() + J.fuzz_jobs():
        entry = fixtures["entries"][job["key"]]
        assert entry["job"] == job, job["key"]
        rgb = _oracle_image(dsrt, oracle, cache, job)
        if J.image_record(rgb)["sha256"] != entry["image"]["sha256"]:
            failures.append((job["key"], len(J.differing_rows(rgb, entry["image"]))))
        lit += entry["image"]["lit"]
    assert not failures, failures
    assert lit > 30000


def test_oracle_against_the_committed_reference_images_pixel_by_pixel(dsrt, oracle, fixtures):
    cache, seen = {}, 0
    for job in J.case_jobs():
        name = job["key"].split("/")[1]
        ppm = os.path.join(J.GOLDEN, f"ref_gpu_detmath_{name}.ppm")
        if not os.path.exists(ppm):
            continue
        ref = J.read_ppm(ppm)
        assert J.image_record(ref)["sha256"] == fixtures["entries"][job["key"]]["image"]["sha256"]
        bad = np.argwhere((_oracle_image(dsrt, oracle, cache, job) != ref).any(axis=2))
        assert len(bad) == 0, (name, len(bad), bad[:5].tolist())
        seen += 1
    assert seen >= 2


def test_oracle_rows_of_the_station_pose_frame_equal_the_reference_kernels(dsrt, oracle, fixtures, tmp_path):
    """The 100,000-triangle station on pose frame 98 of the reference's pose file, 640 x 360 x 32 samples, depth 50: 24 rows spread over the image (the oracle needs a
    second per row) against the reference kernel's per-row CRCs."""
    tris, W, H, spp, _ = J.STATION_JOBS[0]
    obj = J.station_obj(tris, tmp_path)
    entry = fixtures["entries"][J.station_key(tris, W, H, spp, 98)]
    assert entry["job"]["obj_sha256"] == J.file_sha256(obj), "the mesh generator's output changed: regenerate the fixtures (tests/golden/make_ref_gpu_fixtures.py)"
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    fr = dsrt.pose_to_frame(dsrt.read_pose_file(J.POSES)[98])
    scene = hs.view(dsrt.frame_camera(fr, 40.0, W, H, spp, 50), tuple(fr.sun_dir_model))
    import zlib
    lit = 0
    for y in range(7, H, 15):                         # kernel row y is image row H - 1 - y
        rgb, _, _ = oracle.render(scene, W, H, y0=y, y1=y + 1, want_f32=False)
        row = H - 1 - y
        assert zlib.crc32(rgb[row].tobytes()) == entry["image"]["row_crc32"][row], f"image row {row} differs from the reference kernel's"
        lit += int((rgb[row].max(axis=1) > 0).sum())
    assert lit > 500
