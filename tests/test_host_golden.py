"""Host side vs the REFERENCE'S OWN host code (vectors made by tests/golden/make_golden.py from oracle/_ref).

Everything here is bit-exact: ABI offsets, pose -> frame, camera basis, OBJ/MTL flattening, texture pool, BVH.
"""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN, load_world


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_abi_matches_reference_layout(dsrt):
    ref = json.load(open(os.path.join(GOLDEN, "ref_abi.json")))
    capi = dsrt.capi
    for key, want in ref.items():
        if key == "end":
            continue
        if key.startswith("sizeof_"):
            assert C.sizeof(getattr(capi, key[len("sizeof_"):])) == want, key
        else:
            struct_name, field = key.split(".")
            assert getattr(getattr(capi, struct_name), field).offset == want, key


def test_library_exports_every_declared_symbol(dsrt):
    for name in dsrt.capi.EXPORTS:
        assert hasattr(dsrt.lib, name), name
    # and the header declares nothing we forgot to list
    header = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "dsrt.h")).read()
    import re
    declared = set(re.findall(r"\b(dsrt_[a-z0-9_]+|gpu_render_scene)\s*\(", header))
    assert declared == set(dsrt.capi.EXPORTS)
    assert dsrt.lib.dsrt_abi_version() == dsrt.capi.header_abi_version() >= 4


def test_pose_file_and_world_to_model_transform(dsrt):
    ref = json.load(open(os.path.join(GOLDEN, "ref_poses_640x360.json")))
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    assert len(poses) == len(ref) == 99          # 1 header + 99 pose lines
    for i, (p, want) in enumerate(zip(poses, ref)):
        f = dsrt.pose_to_frame(p)
        assert [_bits(v) for v in f.cam_in_model] == want["cam_in_model"], i
        assert [_bits(v) for v in f.sun_dir_model] == want["sun_dir_model"], i
        assert f.sep_m == want["sep_m"] and bool(f.skipped) == want["skipped"], i
        cam = dsrt.frame_camera(f, 40.0, 640, 360, 64, 50)
        assert bytes(cam).hex() == want["gpu_camera"], i
    # the two known answers SURVEY.md section 8(d) quotes
    f0, f98 = dsrt.pose_to_frame(poses[0]), dsrt.pose_to_frame(poses[98])
    assert np.allclose(list(f0.cam_in_model), [-0.72611237, 0.0, 1786.9741], rtol=0, atol=1e-4)
    assert np.allclose(list(f0.sun_dir_model), [0.32780463, -0.7564221, 0.5660121], rtol=0, atol=1e-7)
    assert np.allclose(list(f98.cam_in_model), [-0.000289917, 0.0, 35.739487], rtol=0, atol=1e-5)


def test_pose_reader_skips_malformed_lines(dsrt, tmp_path):
    p = tmp_path / "poses.txt"
    p.write_text("# header\n\n1 2 3 4 5 6 7 8 9\nnot a pose\n1 2 3 4 5 6 7 8\n10 20 30 40 50 60 70 80 90 extra\n")
    poses = dsrt.read_pose_file(p)
    assert len(poses) == 2 and list(poses[1].cam_pos_world) == [10.0, 20.0, 30.0]
    empty = tmp_path / "none.txt"
    empty.write_text("# nothing\n")
    with pytest.raises(dsrt.DsrtError):
        dsrt.read_pose_file(empty)
    with pytest.raises(dsrt.DsrtError):
        dsrt.read_pose_file(tmp_path / "missing.txt")


def test_camera_basis(dsrt):
    for case in json.load(open(os.path.join(GOLDEN, "ref_cameras.json"))):
        a = case["args"]
        cam = dsrt.camera_look_at(a[0:3], a[3:6], a[6], a[7], a[8], a[9], a[10])
        assert bytes(cam).hex() == case["gpu_camera"], a


@pytest.mark.parametrize("name", ["c1_spheres", "lights", "station_3k", "textured", "quirks", "mixed"])
def test_flatten_and_bvh_match_reference_builder(dsrt, name):
    ref = np.load(os.path.join(GOLDEN, f"ref_scene_{name}.npz"))
    counts = json.load(open(os.path.join(GOLDEN, "ref_scenes.json")))[name]["counts"]
    got = load_world(dsrt, name).arrays()
    assert len(got["tris"]) == counts["num_triangles"] and len(got["nodes"]) == counts["num_bvh_nodes"]
    assert len(got["mats"]) == counts["num_materials"] and len(got["spheres"]) == counts["num_spheres"]
    for key in ("tris", "spheres", "mats", "idx", "nodes", "texhdr", "texpool"):
        assert got[key].tobytes() == ref[key].tobytes(), f"{name}: {key} differs from the reference builder"


def test_a_scene_handed_over_as_arrays_with_texture_files_is_the_scene_the_loader_builds(dsrt):
    """What integration/reference_entry_points.cpp does behind the reference's build_gpu_scene: the host owns the flattening (there, the reference's own
    classes) and hands over reference-layout arrays plus texture FILES (dsrt_host_scene_add_texture_file); tree, texture table and pool must come out as
    they do when this library's loader reads the same OBJ -- which test_flatten_and_bvh_match_reference_builder ties to the reference's builder."""
    whole = dsrt.HostScene().add_obj(os.path.join(ASSETS, "textured.obj"))
    want = whole.arrays()
    maps = ["checker.ppm", "stripes.png", "does_not_exist.png"]                          # in the order the triangles first name them
    again = dsrt.HostScene()
    slots = [again.add_texture_file(os.path.join(ASSETS, m)) for m in maps]
    assert slots == [0, 1, 2] and again.add_texture_file(os.path.join(ASSETS, "stripes.png")) == 1      # one slot per distinct path
    assert sorted(set(int(t) for t in want["tris"]["albedo_tex"])) == [-1, 0, 1, 2]
    again.add_arrays(tris=want["tris"], mats=want["mats"])
    got = again.arrays()
    for key in ("tris", "mats", "idx", "nodes", "texhdr", "texpool"):
        assert got[key].tobytes() == want[key].tobytes(), key
    assert [os.path.basename(p) for p in again.texture_failures] == ["does_not_exist.png"]
    upright = dsrt.HostScene()
    upright.add_texture_file(os.path.join(ASSETS, "checker.ppm"), flip_vertically=False)
    w, h = int(want["texhdr"][0]["width"]), int(want["texhdr"][0]["height"])
    a = upright.arrays()["texpool"].reshape(h, w, 3)
    assert np.array_equal(a[::-1], want["texpool"][: w * h * 3].reshape(h, w, 3)) and not np.array_equal(a, a[::-1])
    with pytest.raises(Exception):
        dsrt.HostScene().add_texture_file("")


def test_big_leaf_and_stack_need(dsrt):
    hs = load_world(dsrt, "quirks")
    nodes = hs.arrays()["nodes"]
    assert nodes["tri_count"].max() == 9          # nine coincident triangles cannot be split: one leaf of 9
    assert hs.stack_need >= 1
    hs3k = load_world(dsrt, "station_3k")
    assert 8 <= hs3k.stack_need <= 16


def test_loader_skips_out_of_range_indices(dsrt, tmp_path):
    # the reference indexes past its vertex array here (undefined behaviour); we skip the offending triangle/face
    obj = tmp_path / "bad.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nf 1 2 9\nf 7 1 2\nf -1 1 2\nf 1 2 3 9\n")
    hs = dsrt.HostScene().add_obj(obj)
    assert len(hs.arrays()["tris"]) == 2
    with pytest.raises(dsrt.DsrtError):
        dsrt.HostScene().add_obj(tmp_path / "missing.obj")


def test_ppm_writer(dsrt, tmp_path):
    rgb = (np.arange(4 * 3 * 3) % 251).astype(np.uint8).reshape(3, 4, 3)
    out = tmp_path / "o.ppm"
    dsrt.write_ppm(out, rgb, 4, 3)
    assert out.read_bytes() == b"P6\n4 3\n255\n" + rgb.tobytes()


def test_png_writer(dsrt, tmp_path):
    import zlib
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out = tmp_path / "o.png"
    dsrt.write_png(out, rgb, 53, 37)
    raw = out.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body)
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (53, 37, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(37, 1 + 53 * 3)
    assert not rows[:, 0].any() and np.array_equal(rows[:, 1:].reshape(37, 53, 3), rgb)
    with pytest.raises(RuntimeError):
        dsrt.write_png(tmp_path / "no_such_dir" / "o.png", rgb, 53, 37)


def test_undecodable_texture_is_reported_not_silently_white(dsrt, tmp_path):
    """The reference's stb_image reads JPEG; this library reads PNM and non-interlaced PNG.  A map it cannot decode becomes the reference's
    own failure fallback (1x1 white, src/gpu_scene_builder.cpp:216-221) AND is reported through the C ABI, so a host can refuse the scene."""
    (tmp_path / "panel.jpg").write_bytes(bytes.fromhex("ffd8ffe000104a46494600010100000100010000ffd9"))      # a JFIF shell, no image
    (tmp_path / "m.mtl").write_text("newmtl skin\nKd 0.5 0.4 0.3\nmap_Kd panel.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl skin\nf 1/1 2/2 3/3\n")
    hs = dsrt.HostScene().add_obj(tmp_path / "m.obj")
    hs.build_bvh()
    bad = hs.texture_failures
    assert len(bad) == 1 and bad[0].endswith("panel.jpg")
    a = hs.arrays()
    assert a["texhdr"].size == 1 and tuple(a["texhdr"][0]) [:2] == (1, 1) and np.array_equal(a["texpool"], np.float32([1, 1, 1]))
    assert a["tris"]["albedo_tex"][0] == 0
    ok = dsrt.HostScene().add_obj(os.path.join(ASSETS, "textured.obj"))
    assert [os.path.basename(p) for p in ok.texture_failures] == ["does_not_exist.png"]       # the asset's deliberately missing map; its two real maps decode


def _jpeg_fixtures():
    ref = json.load(open(os.path.join(GOLDEN, "ref_stb_decode.json")))
    return [(name, os.path.join(ASSETS, "jpeg", name + ".jpg"), r) for name, r in sorted(ref.items())]


def test_jpeg_decoder_gives_the_texels_of_the_reference_stb_image(dsrt):
    """The reference decodes texture maps with its vendored stb_image (src/gpu_scene_builder.cpp:215).  host/jpeg_decode.cpp is our own decoder;
    tests/golden/ref_stb_decode.json holds what the reference's own stb build (oracle/_ref/ref_host `decode`) made of 16 JPEG files written by
    libjpeg: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, optimised Huffman tables, restart markers, grayscale, 1-pixel-wide and -high images,
    quality 10 to 100.  Every texel must be the reference's, byte for byte."""
    seen = set()
    for name, path, r in _jpeg_fixtures():
        assert r["ok"] == 1
        want = np.frombuffer(bytes.fromhex(r["rgb"]), np.uint8).reshape(r["h"], r["w"], 3)
        got = dsrt.decode_image_file(path)
        assert got.shape == want.shape and np.array_equal(got, want), name
        flipped = dsrt.decode_image_file(path, flip_vertically=True)
        assert np.array_equal(flipped, want[::-1]), name
        raw = open(path, "rb").read()
        seen.add("progressive" if b"\xff\xc2" in raw else "sequential")
        seen.add("restart" if b"\xff\xdd" in raw else "no-restart")
    assert seen == {"progressive", "sequential", "restart", "no-restart"}


def test_jpeg_map_goes_through_the_builder_like_any_texture(dsrt, tmp_path):
    import shutil
    name, path, r = [f for f in _jpeg_fixtures() if f[0] == "prog_420_q85"][0]
    shutil.copy(path, tmp_path / "skin.jpg")
    (tmp_path / "m.mtl").write_text("newmtl skin\nKd 0.5 0.4 0.3\nmap_Kd skin.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl skin\nf 1/1 2/2 3/3\n")
    hs = dsrt.HostScene().add_obj(tmp_path / "m.obj")
    hs.build_bvh()
    assert hs.texture_failures == []
    a = hs.arrays()
    assert tuple(a["texhdr"][0])[:2] == (r["w"], r["h"])
    want = np.frombuffer(bytes.fromhex(r["rgb"]), np.uint8).reshape(r["h"], r["w"], 3)[::-1]      # the loader's flip latch is on after an OBJ with maps (SURVEY.md note T)
    lut = np.power(np.arange(256, dtype=np.float32) / np.float32(255.0), np.float32(2.2)).astype(np.float32)
    assert np.allclose(a["texpool"].reshape(r["h"], r["w"], 3), lut[want], rtol=0, atol=1e-7)


def test_damaged_jpegs_fail_cleanly(dsrt, tmp_path):
    """Truncated and corrupted files (headers come from untrusted packages): the decoder either refuses or returns an image of the declared
    size -- it never crashes and never sizes a buffer from a header the file cannot back."""
    rng = np.random.default_rng(7)
    tried = refused = 0
    for name, path, r in _jpeg_fixtures()[:6]:
        raw = bytearray(open(path, "rb").read())
        variants = [raw[:n] for n in (0, 1, 2, 3, 10, 50, len(raw) // 2, len(raw) - 2)]
        for _ in range(40):
            v = bytearray(raw)
            for _ in range(int(rng.integers(1, 6))):
                v[int(rng.integers(2, len(v)))] = int(rng.integers(0, 256))
            variants.append(v)
        huge = bytearray(raw)                                  # a frame header claiming 65535 x 65535 in a 1 KB file
        sof = max(huge.find(b"\xff\xc0"), huge.find(b"\xff\xc2"))
        huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
        variants.append(huge)
        for v in variants:
            p = tmp_path / "x.jpg"
            p.write_bytes(bytes(v))
            tried += 1
            try:
                img = dsrt.decode_image_file(p)
                assert img.ndim == 3 and img.shape[2] == 3 and img.size <= (1 << 28) * 3
            except dsrt.DsrtError:
                refused += 1
    assert tried > 250 and refused > 20


def _image_fixtures():
    ref = json.load(open(os.path.join(GOLDEN, "ref_stb_decode_images.json")))
    return [(name, os.path.join(ASSETS, "images", name), r) for name, r in sorted(ref.items())]


def test_bmp_tga_and_interlaced_png_give_the_texels_of_the_reference_stb_image(dsrt):
    """The other containers a map_Kd can name (the reference's stb_image sniffs the content, src/gpu_scene_builder.cpp:215): host/bmp_tga_decode.cpp
    and the Adam7 path of host/image_io.cpp against what the reference's own stb build made of 65 files written byte by byte by
    tests/golden/make_image_fixtures.py -- BMP with 12 / 40 / 108 / 124-byte headers, 1 / 4 / 8-bit palettes, 16 / 32 bits with default and BITFIELDS
    masks (3- to 8-bit fields), top-down, a gap before the pixels; TGA types 1 / 2 / 3 / 9 / 10 / 11, 8 to 32 bits, colour maps of 16 / 24 / 32
    bits, 16-bit indices, both row orders; PNG of every colour type and bit depth, interlaced and not, including images whose first Adam7
    passes are empty.  Same bytes where the reference decodes, a refusal where it refuses (run-length BMP, 10-bit fields)."""
    kinds = set()
    for name, path, r in _image_fixtures():
        kinds.add(name.split("_")[0].rstrip("0123456789") + ("+adam7" if "adam7" in name else ""))
        if not r["ok"]:
            with pytest.raises(dsrt.DsrtError):
                dsrt.decode_image_file(path)
            continue
        want = np.frombuffer(bytes.fromhex(r["rgb"]), np.uint8).reshape(r["h"], r["w"], 3)
        got = dsrt.decode_image_file(path)
        assert got.shape == want.shape and np.array_equal(got, want), name
        assert np.array_equal(dsrt.decode_image_file(path, flip_vertically=True), want[::-1]), name
    assert {"bmp", "tga", "png", "png+adam7"} <= kinds and len(_image_fixtures()) >= 60


def test_damaged_bmp_tga_png_fail_cleanly(dsrt, tmp_path):
    """As for the JPEGs: cut and corrupted files are refused or decode to an image of bounded size, whatever their headers claim."""
    rng = np.random.default_rng(11)
    tried = refused = 0
    picks = [f for f in _image_fixtures() if f[2]["ok"]][::3]
    for name, path, r in picks:
        raw = bytearray(open(path, "rb").read())
        variants = [raw[:n] for n in (0, 1, 2, 10, 17, 18, 30, len(raw) // 2, len(raw) - 1)]
        for _ in range(25):
            v = bytearray(raw)
            for _ in range(int(rng.integers(1, 5))):
                v[int(rng.integers(0, min(len(v), 64)))] = int(rng.integers(0, 256))        # headers live in the first bytes
            variants.append(v)
        for v in variants:
            p = tmp_path / ("x." + name.rsplit(".", 1)[1])
            p.write_bytes(bytes(v))
            tried += 1
            try:
                img = dsrt.decode_image_file(p)
                assert img.ndim == 3 and img.shape[2] == 3 and img.size <= (1 << 28) * 3
            except dsrt.DsrtError:
                refused += 1
    assert tried > 500 and refused > 50
