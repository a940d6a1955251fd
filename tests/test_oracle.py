"""The CPU oracle against everything that can pin it without a GPU.

Reference-made pins: the LCG values SURVEY.md section 8(a2) records; ray-level answers of the reference's CPU classes
(tests/golden/ref_hitkat.json: aabb::hit is the same float algorithm -> exact; sphere::hit / triangle::hit compute in
double from float inputs -> agreement to rounding).  The sampling loop as a whole is pinned by reference-made images in
tests/test_oracle_reference_fixtures.py (the reference's own kernel, executed; see oracle/dsrt_oracle.h); the image hashes at the
bottom of THIS file are regression fixtures produced by the oracle itself.
"""
import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, load_world

SUN = (0.32780463, -0.7564221, 0.5660121)


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def test_lcg_known_answers(oracle):
    s = C.c_uint32(1337)
    want = [(0xC114ED44, 0.0817453861), (0x8CAC17D3, 0.6722385287), (0x46453B16, None)]
    for state, value in want:
        v = oracle.lib.dsrt_oracle_rand01(C.byref(s))
        assert s.value == state
        if value is not None:
            assert abs(v - value) < 1e-9
        assert v == np.float32(state & 0xFFFFFF) / np.float32(16777216.0)


def _one_prim_scene(dsrt, tris=None, spheres=None):
    mats = np.zeros(1, dsrt.capi.MAT_DTYPE)
    mats["albedo"] = 0.5
    mats["albedo_tex"] = -1
    hs = dsrt.HostScene().add_arrays(tris=tris, spheres=spheres, mats=mats)
    hs.build_bvh()
    return hs


def test_ray_level_against_reference_cpu_classes(dsrt, oracle):
    kat = json.load(open(os.path.join(GOLDEN, "ref_hitkat.json")))
    capi = dsrt.capi
    sph = np.zeros(1, capi.SPHERE_DTYPE)
    sph["center"] = (0.25, -0.125, -1.0)
    sph["radius"] = 0.5
    tri = np.zeros(1, capi.TRI_DTYPE)
    tri["v"][0] = [(-1.0, -0.5, -2.0), (1.5, -0.25, -2.5), (0.0, 1.25, -1.5)]
    tri["n"][0] = [(0, 0, 1)] * 3
    tri["albedo_tex"] = -1
    hs_s, hs_t = _one_prim_scene(dsrt, spheres=sph), _one_prim_scene(dsrt, tris=tri)
    vs, vt = hs_s.view(), hs_t.view()
    lo, hi = (C.c_float * 3)(-0.5, -0.25, -1.75), (C.c_float * 3)(0.75, 0.5, -1.0)
    oracle.lib.dsrt_oracle_bbox_hit.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float]
    n_s = n_t = n_b = 0
    for k in kat:
        o = (C.c_float * 3)(*[f32(b) for b in k["o"]])
        d = (C.c_float * 3)(*[f32(b) for b in k["d"]])
        out, ids = (C.c_float * 9)(), (C.c_int * 4)()
        # aabb::hit is bbox_hit: exact agreement, including the 1/0 axis-parallel rays (every 7th)
        assert oracle.lib.dsrt_oracle_bbox_hit(lo, hi, o, d, 0.001, 1e9) == k["box"]
        n_b += k["box"]
        hit = oracle.lib.dsrt_oracle_scene_hit(C.byref(vs), o, d, 0.001, 1e9, out, ids)
        assert hit == k["sphere"]
        if hit:
            t_ref = struct.unpack("<d", struct.pack("<Q", k["sphere_t"]))[0]
            assert abs(out[0] - t_ref) <= 2e-5 * abs(t_ref)
            n_s += 1
        hit = oracle.lib.dsrt_oracle_scene_hit(C.byref(vt), o, d, 0.001, 1e9, out, ids)
        assert hit == k["tri"]
        if hit:
            t_ref = struct.unpack("<d", struct.pack("<Q", k["tri_t"]))[0]
            assert abs(out[0] - t_ref) <= 2e-5 * abs(t_ref)
            n_t += 1
    assert n_s > 20 and n_t > 20 and n_b > 20, (n_s, n_t, n_b)       # the vectors do exercise hits


def test_centre_ray_on_unit_diameter_sphere(dsrt, oracle):
    # the reference's hittable_list::hit gives t = 0.875965 for the centre ray of its probe scene;
    # here the analytic value for our C1 camera: ray from the origin straight at the sphere at z = -1, r = 0.5 -> t = 0.5.
    hs = load_world(dsrt, "c1_spheres")
    v = hs.view()
    o, d = (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 0, -1)
    out, ids = (C.c_float * 9)(), (C.c_int * 4)()
    assert oracle.lib.dsrt_oracle_scene_hit(C.byref(v), o, d, 0.001, 1e9, out, ids) == 1
    assert out[0] == 0.5 and ids[0] == 1 and ids[2] == -1 and ids[3] == 1
    assert (out[4], out[5], out[6]) == (0.0, 0.0, 1.0)


def _render(dsrt, oracle, world, cam_args, spp, sun=SUN, rows=None):
    hs = load_world(dsrt, world)
    W, H = cam_args[3], cam_args[4]
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, cam_args[5])
    scene = hs.view(cam, sun)
    y0, y1 = rows if rows else (0, H)
    return hs, scene, oracle.render(scene, W, H, y0, y1)


CASES = {
    # name: (world, (lookfrom, lookat, vfov, W, H, max_depth), spp)
    "c1_spheres": ("c1_spheres", ((-2.0, 2.0, 1.0), (0.0, 0.0, -1.0), 20.0, 200, 112, 50), 16),
    "lights": ("lights", ((0.0, 3.0, 9.0), (0.0, 2.0, 0.0), 45.0, 120, 80, 12), 8),
    "station_far": ("station_3k", ((-0.7, 0.0, 260.0), (0.0, 0.0, 0.0), 40.0, 200, 112, 50), 16),
    "station_near": ("station_3k", ((12.0, 9.0, 38.0), (0.0, 0.0, 0.0), 40.0, 200, 112, 50), 16),
    "textured": ("textured", ((0.5, 2.0, 6.0), (0.0, 1.2, -2.0), 45.0, 96, 64, 8), 8),
    "mixed": ("mixed", ((3.0, 6.0, 14.0), (0.0, 2.0, 0.0), 40.0, 96, 64, 10), 8),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_images_are_stable_and_nontrivial(dsrt, oracle, name):
    world, cam_args, spp = CASES[name]
    _, _, (rgb, f32img, cnt) = _render(dsrt, oracle, world, cam_args, spp)
    assert cnt["samples"] == cam_args[3] * cam_args[4] * spp
    assert cnt["primary_hits"] > 0 and rgb.max() > 0
    assert cnt["rays"] >= cnt["samples"] and cnt["max_stack"] <= 64
    # box_fetches identity of SURVEY.md section 8(d): one root box per BVH ray + two per internal node entered
    if cnt["box_fetches"]:
        assert cnt["box_fetches"] == cnt["rays"] + 2 * cnt["internal_entered"]
    fixtures = json.load(open(os.path.join(GOLDEN, "oracle_images.json")))
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == fixtures[name]["rgb8_sha256"], \
        "oracle output changed (regression fixture made by this oracle, not by the reference)"


def test_rows_are_independent(dsrt, oracle):
    world, cam_args, spp = CASES["station_near"]
    _, _, (full, _, _) = _render(dsrt, oracle, world, cam_args, spp)
    _, _, (part, _, _) = _render(dsrt, oracle, world, cam_args, spp, rows=(40, 60))
    H = cam_args[4]
    assert np.array_equal(part[H - 60:H - 40], full[H - 60:H - 40])      # kernel row y is image row H-1-y
    assert not part[:H - 60].any() and not part[H - 40:].any()


def test_libm_variant_is_statistically_the_same(dsrt, oracle, oracle_libm):
    # cosf/sinf/powf from glibc instead of dsrt_detmath.h: last-ulp differences de-synchronise a few pixels' LCG streams;
    # the images must still agree except for Monte-Carlo noise in those pixels.
    world, cam_args, spp = CASES["station_near"]
    hs, scene, (a, _, _) = _render(dsrt, oracle, world, cam_args, spp)
    b, _, _ = oracle_libm.render(scene, cam_args[3], cam_args[4])
    diff = np.abs(a.astype(int) - b.astype(int))
    assert (diff.max(axis=2) > 0).mean() < 0.05
    assert abs(a.astype(float).mean() - b.astype(float).mean()) < 0.5


def test_book_style_baseline_runs_on_c1(tmp_path):
    # oracle/_ref/book_render = the reference's own CPU classes under our book-style loop (BASELINE.json configs[0]).
    import subprocess
    from conftest import ASSETS, ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "book_render")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/book_render not built (needs /root/reference)")
    out = tmp_path / "c1.ppm"
    r = subprocess.run([exe, "c1_spheres.world", "100", "56", "4", "50", "1", "-2", "2", "1", "0", "0", "-1", "20", "0.3", "-0.8", "0.5", str(out)],
                       cwd=ASSETS, stdout=subprocess.PIPE, text=True, check=True)
    rep = json.loads(r.stdout)
    assert rep["samples"] == 100 * 56 * 4 and rep["seconds"] > 0
    data = out.read_bytes()
    assert data.startswith(b"P6\n100 56\n255\n")
    img = np.frombuffer(data[len(b"P6\n100 56\n255\n"):], np.uint8).reshape(56, 100, 3)
    assert (img.max(axis=2) > 0).mean() > 0.3          # ground + spheres are lit by the sun term
    # a row band renders only its rows
    r2 = subprocess.run([exe, "c1_spheres.world", "100", "56", "4", "50", "1", "-2", "2", "1", "0", "0", "-1", "20", "0.3", "-0.8", "0.5", "-", "10", "20"],
                        cwd=ASSETS, stdout=subprocess.PIPE, text=True, check=True)
    assert json.loads(r2.stdout)["samples"] == 100 * 10 * 4


def test_device_helper_known_answers_from_the_reference(oracle):
    """tests/golden/ref_devkat.json comes from the reference's own host-compilable device helpers, executed (oracle/ref_host_driver.cpp
    `devkat`): random_float_device (inc/rtweekend.h:126-133), random_in_unit_sphere_device (:164-171), random_cosine_direction_device
    (:190-202), generate_camera_ray_device with lens_radius 0 (inc/camera.h:35-61).  The kernel has float copies of the same formulas
    (src/gpu_render.cu:77-109, 941-968) which the oracle restates.  Exact where the arithmetic is float or exact (LCG, rejection loop,
    camera ray); to rounding where the helper works in double (cosine direction: cos/sin/sqrt in double rounded once vs the oracle's
    float sqrt and shared deterministic sin/cos).  ray_color's control flow and the traversal order are pinned by the reference kernel's images (tests/test_oracle_reference_fixtures.py)."""
    kat = json.load(open(os.path.join(GOLDEN, "ref_devkat.json")))
    L = oracle.lib
    for fn in (L.dsrt_oracle_random_in_unit_sphere, L.dsrt_oracle_random_cosine_direction):
        fn.restype = None
        fn.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.dsrt_oracle_camera_ray.restype = None
    L.dsrt_oracle_camera_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    bits = lambda arr: [struct.unpack("<I", struct.pack("<f", v))[0] for v in arr]            # noqa: E731
    # LCG: state sequence and the float, bit for bit, from four seeds (incl. the wrap-around ones)
    for row in kat["lcg"]:
        s = C.c_uint32(row["seed"])
        for state, fbits in row["draws"]:
            v = L.dsrt_oracle_rand01(C.byref(s))
            assert s.value == state and bits([v])[0] == fbits
    assert kat["lcg"][0]["draws"][0][0] == 0xC114ED44                    # the known answer SURVEY.md section 8(a2) quotes
    # rejection loop: same number of draws (final state) and the same point, bit for bit
    loops = 0
    for row in kat["unit_sphere"]:
        s, out = C.c_uint32(row["state_in"]), (C.c_float * 3)()
        L.dsrt_oracle_random_in_unit_sphere(C.byref(s), out)
        # random_vec3_device builds vec3(draw, draw, draw): the order in which a compiler evaluates the three arguments is unspecified and
        # g++ (which built oracle/_ref) goes right to left, so the helper's x is the THIRD draw.  The kernel's own copy
        # (src/gpu_render.cu:82-91, what the oracle restates) draws x, y, z in statement order.  Same three values, same accept/reject.
        assert s.value == row["state_out"] and bits(out)[::-1] == row["p"]
        t = C.c_uint32(row["state_in"])
        for _ in range(3):
            L.dsrt_oracle_rand01(C.byref(t))
        loops += t.value != row["state_out"]
    assert loops > 10                                                     # the vectors do exercise rejections (48 % of candidates are outside)
    # cosine direction: two draws, z exact (sqrt of an exact float difference), x / y to float rounding of double math
    worst = 0.0
    for row in kat["cosine_direction"]:
        s, out = C.c_uint32(row["state_in"]), (C.c_float * 3)()
        L.dsrt_oracle_random_cosine_direction(C.byref(s), out)
        assert s.value == row["state_out"]
        ref = [f32(b) for b in row["d"]]
        assert bits([out[2]])[0] == row["d"][2]
        worst = max(worst, abs(out[0] - ref[0]), abs(out[1] - ref[1]))
    assert worst <= 4e-7, worst                                           # |x|, |y| <= 1: a few float ulps
    # camera ray (frame 0 camera of the pose file, 200 x 112): direction and origin bit for bit given the helper's own jitter
    cam_rec = kat["camera_ray"][0]
    cam = (C.c_char * 104).from_buffer_copy(bytes.fromhex(cam_rec["camera"]))
    for row in cam_rec["rays"]:
        o, d = (C.c_float * 3)(), (C.c_float * 3)()
        L.dsrt_oracle_camera_ray(C.byref(cam), row["px"], row["py"], 200, 112, f32(row["jx"]), f32(row["jy"]), o, d)
        assert bits(o) == row["orig"] and bits(d) == row["dir"], row


# ---- known answers of the reference's material / frame helpers (tests/golden/ref_matkat.json) ------------------------------------------
def _f32a(rows, key):
    return np.array([[f32(b) for b in r[key]] for r in rows], np.float32)


class OracleMatkat:
    """The oracle's copies of reflect / refract / scatter_metal / scatter_dielectric / build_onb / schlick, batch interface (one row per case)."""

    def __init__(self, oracle):
        L = self.L = oracle.lib
        F3 = C.POINTER(C.c_float)
        L.dsrt_oracle_reflect.restype = None
        L.dsrt_oracle_reflect.argtypes = [F3, F3, F3]
        L.dsrt_oracle_refract.restype = None
        L.dsrt_oracle_refract.argtypes = [F3, F3, C.c_float, F3]
        L.dsrt_oracle_normalize.restype = None
        L.dsrt_oracle_normalize.argtypes = [F3, F3]
        L.dsrt_oracle_scatter_metal.restype = C.c_int
        L.dsrt_oracle_scatter_metal.argtypes = [F3, F3, C.c_float, C.POINTER(C.c_uint32), F3]
        L.dsrt_oracle_scatter_dielectric.restype = None
        L.dsrt_oracle_scatter_dielectric.argtypes = [F3, F3, C.c_int, C.c_float, C.POINTER(C.c_uint32), F3]
        L.dsrt_oracle_build_onb.restype = None
        L.dsrt_oracle_build_onb.argtypes = [F3, F3, F3, F3]
        L.dsrt_oracle_schlick.restype = C.c_float
        L.dsrt_oracle_schlick.argtypes = [C.c_float, C.c_float]

    @staticmethod
    def _v(a):
        return (C.c_float * 3)(*[float(x) for x in a])

    def reflect(self, v, n):
        out = np.zeros_like(v)
        for i in range(len(v)):
            o = (C.c_float * 3)()
            self.L.dsrt_oracle_reflect(self._v(v[i]), self._v(n[i]), o)
            out[i] = list(o)
        return out

    def refract(self, v, n, eta):
        out = np.zeros_like(v)
        for i in range(len(v)):
            o = (C.c_float * 3)()
            self.L.dsrt_oracle_refract(self._v(v[i]), self._v(n[i]), float(eta[i]), o)
            out[i] = list(o)
        return out

    def normalize(self, v):
        out = np.zeros_like(v)
        for i in range(len(v)):
            o = (C.c_float * 3)()
            self.L.dsrt_oracle_normalize(self._v(v[i]), o)
            out[i] = list(o)
        return out

    def metal(self, d, n, fuzz, state):
        out, ok, st = np.zeros_like(d), np.zeros(len(d), bool), np.zeros(len(d), np.uint32)
        for i in range(len(d)):
            o, s = (C.c_float * 3)(), C.c_uint32(int(state[i]))
            ok[i] = bool(self.L.dsrt_oracle_scatter_metal(self._v(d[i]), self._v(n[i]), float(fuzz[i]), C.byref(s), o))
            out[i], st[i] = list(o), s.value
        return out, ok, st

    def dielectric(self, d, n, front, ref_idx, state):
        out, st = np.zeros_like(d), np.zeros(len(d), np.uint32)
        for i in range(len(d)):
            o, s = (C.c_float * 3)(), C.c_uint32(int(state[i]))
            self.L.dsrt_oracle_scatter_dielectric(self._v(d[i]), self._v(n[i]), int(front[i]), float(ref_idx[i]), C.byref(s), o)
            out[i], st[i] = list(o), s.value
        return out, st

    def onb(self, n):
        u, v, w = np.zeros_like(n), np.zeros_like(n), np.zeros_like(n)
        for i in range(len(n)):
            a, b, c = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
            self.L.dsrt_oracle_build_onb(self._v(n[i]), a, b, c)
            u[i], v[i], w[i] = list(a), list(b), list(c)
        return u, v, w

    def schlick(self, cos, ratio):
        return np.array([self.L.dsrt_oracle_schlick(float(a), float(b)) for a, b in zip(cos, ratio)], np.float32)


def check_material_known_answers(impl, normalize):
    """`impl` (the oracle, or the device helpers behind dsrt_selftest_devkat) against what the reference's host code computed
    (oracle/ref_host_driver.cpp `matkat`).  Values are compared as floats (+0 == -0: the host adds `0.0f * vector` where the kernel adds
    `vector * 0.0f`); `normalize` is the oracle's f3_norm, used to tell which refract inputs survive the kernel's extra normalisation."""
    kat = json.load(open(os.path.join(GOLDEN, "ref_matkat.json")))
    # reflect (inc/vec3.h:136-139 = src/gpu_render.cu:195): float arithmetic on both sides
    rows = kat["reflect"]
    assert np.array_equal(impl.reflect(_f32a(rows, "v"), _f32a(rows, "n")), _f32a(rows, "out"))
    # refract (inc/vec3.h:141-147 = :199-206 after the kernel's f3_norm of its argument)
    rows = kat["refract"]
    uv, n, eta, want = _f32a(rows, "uv"), _f32a(rows, "n"), np.array([f32(r["eta"]) for r in rows], np.float32), _f32a(rows, "out")
    got = impl.refract(uv, n, eta)
    same_after_norm = (normalize(uv).view(np.uint32) == uv.view(np.uint32)).all(axis=1)
    assert same_after_norm.sum() >= 16, same_after_norm.sum()
    assert np.array_equal(got[same_after_norm], want[same_after_norm])
    assert np.abs(got - want).max() <= 2e-6                                  # the others (|out| up to 2.4) differ by the re-normalisation's rounding only
    # metal::scatter at fuzz 0 (inc/material.h:123-137 = scatter_metal :603-619): direction, accept test, and a whole number of rejection-loop trips
    rows = kat["metal_fuzz0"]
    state = np.array([(7919 * i + 13) & 0xFFFFFFFF for i in range(len(rows))], np.uint32)
    out, ok, st = impl.metal(_f32a(rows, "dir"), _f32a(rows, "n"), np.zeros(len(rows), np.float32), state)
    assert np.array_equal(out, _f32a(rows, "out"))
    want_ok = np.array([r["ok"] for r in rows], bool)
    assert np.array_equal(ok, want_ok) and 8 < want_ok.sum() < len(rows) - 8       # both outcomes are exercised
    for s0, s1 in zip(state, st):
        s, trips = int(s0), 0
        while s != int(s1) and trips < 64:
            for _ in range(3):
                s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            trips += 1
        assert 1 <= trips < 64
    # dielectric::scatter, total internal reflection (inc/material.h:153-180 = scatter_dielectric :621-661): reflect(unit_dir, n), and NO draw
    rows = kat["dielectric_tir"]
    state = np.array([(104729 * i + 7) & 0xFFFFFFFF for i in range(len(rows))], np.uint32)
    out, st = impl.dielectric(_f32a(rows, "dir"), _f32a(rows, "n"), np.array([r["front"] for r in rows], np.int32),
                              np.array([f32(r["ref_idx"]) for r in rows], np.float32), state)
    assert np.array_equal(out, _f32a(rows, "out"))
    assert np.array_equal(st, state)                                          # `cannot_refract || ...` short-circuits: the stream does not move
    assert {r["front"] for r in rows} == {0, 1}
    # reflectance (inc/material.h:28-32, double) = schlick :208-212 (float, shared deterministic powf): to rounding
    rows = kat["reflectance"]
    got = impl.schlick(np.array([f32(r["cos"]) for r in rows], np.float32), np.array([f32(r["ratio"]) for r in rows], np.float32))
    want = np.array([struct.unpack("<d", struct.pack("<Q", r["out_f64"]))[0] for r in rows])
    assert np.abs(got.astype(np.float64) - want).max() <= 3e-7
    # onb::build_from_w (inc/onb.h:47-56) = build_onb :112-118: same w and v; the class forms u = w x v, the kernel u = v x w = -(w x v), exactly
    rows = kat["onb"]
    u, v, w = impl.onb(_f32a(rows, "n"))
    assert np.array_equal(w, _f32a(rows, "w")) and np.array_equal(v, _f32a(rows, "v")) and np.array_equal(u, -_f32a(rows, "u"))
    assert sum(abs(f32(r["w"][0])) > 0.9 for r in rows) >= 6                  # both helper axes are exercised


def test_material_and_frame_known_answers_from_the_reference(oracle):
    """What ray_color's specular branches and the cosine sampler are made of, pinned by the reference's own host code, executed:
    what these leaves leave open -- ray_color's control flow, scene_hit's combination, the traversal order -- is pinned by the reference kernel's images (tests/test_oracle_reference_fixtures.py)."""
    impl = OracleMatkat(oracle)
    check_material_known_answers(impl, impl.normalize)
