"""The renders the reference's own kernel is pinned on -- ONE list, used by three parties:

  * tests/golden/make_ref_gpu_fixtures.py   runs oracle/_ref/ref_gpu (the reference's renderer, oracle/Makefile) on every job, on the GPU box, and
                                            writes tests/golden/ref_gpu_images.json (sha256 + per-row CRC32 + lit-pixel count of each image);
  * tests/test_gpu_reference_fixtures.py    renders every job with the PRODUCT library in math_mode 1 and compares with that file -- no reference binary needed;
  * tests/test_gpu_reference_kernel.py      renders the same jobs with the live binary where it exists.

A job is plain data: which world, which camera, which size.  Nothing here touches /root/reference.
"""
import hashlib
import os
import zlib

import numpy as np

from test_oracle import CASES, SUN

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ASSETS = os.path.join(GOLDEN, "assets")
POSES = os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt")
FIXTURE = os.path.join(GOLDEN, "ref_gpu_images.json")

FUZZ_WORLDS = ("c1_spheres", "lights", "station_3k", "textured", "mixed", "quirks")


def _f32(v):
    return float(np.float32(v))           # every number crosses both command lines as an exactly representable float


def case_jobs():
    """The six scenes of the parity suite (tests/test_oracle.py CASES)."""
    jobs = []
    for name in sorted(CASES):
        world, (lookfrom, lookat, vfov, W, H, depth), spp = CASES[name]
        jobs.append({"key": "case/" + name, "world": world, "W": W, "H": H, "spp": spp, "depth": depth, "from": [_f32(v) for v in lookfrom],
                     "at": [_f32(v) for v in lookat], "vfov": _f32(vfov), "sun": [_f32(v) for v in SUN]})
    return jobs


def fuzz_jobs():
    """30 random views over all six world files: ragged sizes, 1-24 samples, depths 1-50, cameras from inside the geometry to far outside, random
    un-normalised sun directions.  (The generator and its seed are part of the fixture: changing either invalidates ref_gpu_images.json.)"""
    rng = np.random.default_rng(20251005)
    jobs = []
    for trial in range(30):
        world = FUZZ_WORLDS[trial % len(FUZZ_WORLDS)]
        W, H = int(rng.integers(8, 150)), int(rng.integers(6, 100))
        spp = int(rng.choice([1, 2, 5, 9, 24]))
        depth = int(rng.choice([1, 2, 5, 12, 50]))
        dist = float(rng.choice([0.5, 3.0, 9.0, 30.0, 120.0, 600.0])) * (0.15 if world in ("c1_spheres", "lights", "textured", "mixed") else 1.0)
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        jobs.append({"key": f"fuzz/{trial:02d}", "world": world, "W": W, "H": H, "spp": spp, "depth": depth,
                     "from": [_f32(v) for v in direction * dist + (0.0, 1.0, 0.0)], "at": [_f32(v) for v in rng.normal(size=3) * (0.0 if trial % 3 else 0.5)],
                     "vfov": _f32(rng.choice([20.0, 40.0, 75.0])), "sun": [_f32(v) for v in rng.normal(size=3)]})
    return jobs


# The procedural station on frames of the reference's pose file: (triangles, W, H, spp, pose frames).  The last two are THE BENCH'S mesh at the bench's size:
# 12 samples (seconds for the reference kernel) and the whole headline frame, 1000 samples (about ten seconds for the reference kernel).
STATION_JOBS = ((100000, 640, 360, 32, (98, 60)), (1000000, 1920, 1080, 12, (98,)), (1000000, 1920, 1080, 1000, (98,)))


def station_key(tris, W, H, spp, frame):
    return f"station/{tris}/frame{frame:02d}/{W}x{H}x{spp}"


def station_obj(tris, scratch):
    """Path of the OBJ for a station job (written if absent).  The 1 M mesh is the file bench.py and test_headline_mesh_rows_match_the_oracle use."""
    from dsrt_amd import meshgen
    if tris == 1000000:
        obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
        if not os.path.exists(obj):
            tmp_obj = obj + f".{os.getpid()}.tmp"
            meshgen.write_obj(meshgen.build_station(1000000), tmp_obj, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
            os.replace(tmp_obj, obj)
        return obj
    obj = os.path.join(str(scratch), f"station_{tris}.obj")
    if not os.path.exists(obj):
        meshgen.generate(obj, tris)
    return obj


def file_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    return h.hexdigest()


def image_record(rgb):
    """What the fixture keeps of an H x W x 3 uint8 image: enough to prove equality (sha256) and to say WHERE a mismatch is (one CRC32 per row)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    return {"H": int(rgb.shape[0]), "W": int(rgb.shape[1]), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest(), "lit": int((rgb.max(axis=2) > 0).sum()),
            "row_crc32": [zlib.crc32(rgb[r].tobytes()) for r in range(rgb.shape[0])]}


def differing_rows(rgb, record):
    ours = image_record(rgb)
    if (ours["H"], ours["W"]) != (record["H"], record["W"]):
        return list(range(ours["H"]))
    return [r for r, (a, b) in enumerate(zip(ours["row_crc32"], record["row_crc32"])) if a != b]


def read_ppm(path):
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


def ref_gpu_command(exe, job, out):
    """Command line of oracle/ref_gpu_driver.cpp for a world-file job (run with cwd = ASSETS)."""
    return [str(c) for c in [exe, job["world"] + ".world", job["W"], job["H"], job["spp"], job["depth"], *[repr(v) for v in job["from"]], *[repr(v) for v in job["at"]],
                             repr(job["vfov"]), *[repr(v) for v in job["sun"]], out]]
