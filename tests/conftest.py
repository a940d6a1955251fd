"""Shared fixtures.  `gpu` marks tests that need a real MI355X; everything else runs on CPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(GOLDEN, "assets")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(ROOT, "deep-space-ray-tracer_amd", "libdsrt_hip.so")
    orc = os.path.join(ROOT, "oracle", "libdsrt_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def dsrt():
    _ensure_built()
    import dsrt_amd
    return dsrt_amd


class Oracle:
    """ctypes handle on oracle/libdsrt_oracle.so -- the CHECKER.  Only tests (and smoke / the bench's cpu_baseline leg) use it."""

    COUNTER_NAMES = ["samples", "rays", "primary_hits", "box_tests", "box_fetches", "nodes_entered", "internal_entered", "tri_tests",
                     "hit_updates", "sphere_tests", "shaded_hits", "tex_fetches", "max_stack", "rng_draws"]

    def __init__(self, libm=False):
        path = os.path.join(ROOT, "oracle", "libdsrt_oracle_libm.so" if libm else "libdsrt_oracle.so")
        self.lib = C.CDLL(path)
        L = self.lib
        L.dsrt_oracle_render_rows.restype = C.c_int
        L.dsrt_oracle_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dsrt_oracle_rand01.restype = C.c_float
        L.dsrt_oracle_rand01.argtypes = [C.POINTER(C.c_uint32)]
        L.dsrt_oracle_scene_hit.restype = C.c_int
        L.dsrt_oracle_scene_hit.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        for n in ("sinf", "cosf"):
            f = getattr(L, "dsrt_oracle_" + n)
            f.restype = C.c_float
            f.argtypes = [C.c_float]
        L.dsrt_oracle_powf.restype = C.c_float
        L.dsrt_oracle_powf.argtypes = [C.c_float, C.c_float]

    def render(self, scene_host_view, W, H, y0=0, y1=None, want_f32=True):
        y1 = H if y1 is None else y1
        rgb = np.zeros((H, W, 3), np.uint8)
        f32 = np.zeros((H, W, 3), np.float32) if want_f32 else None
        cnt = (C.c_uint64 * len(self.COUNTER_NAMES))()
        rc = self.lib.dsrt_oracle_render_rows(C.byref(scene_host_view), W, H, y0, y1, rgb.ctypes.data, f32.ctypes.data if want_f32 else None, cnt)
        assert rc == 0
        return rgb, f32, dict(zip(self.COUNTER_NAMES, list(cnt)))


@pytest.fixture(scope="session")
def oracle(dsrt):
    return Oracle()


@pytest.fixture(scope="session")
def oracle_libm(dsrt):
    return Oracle(libm=True)


@pytest.fixture(scope="session")
def gpu_ctx(dsrt):
    if dsrt.lib.dsrt_device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    ctx = dsrt.Context(0)
    yield ctx
    ctx.close()


def load_world(dsrt, name):
    """HostScene for tests/golden/assets/<name>.world (paths inside are relative to the assets directory)."""
    cwd = os.getcwd()
    os.chdir(ASSETS)
    try:
        hs = dsrt.HostScene().add_world_file(name + ".world")
        hs.build_bvh()
    finally:
        os.chdir(cwd)
    return hs


def require_ref_binary(path):
    """The live reference binaries (oracle/_ref/, built where /root/reference exists and carried to the GPU box by the gpurun snapshot) are part of the parity pin:
    under -m gpu their absence is a FAILURE, not a skip -- a box that lost them would otherwise report green with the pin gone.  A clean checkout that is known
    not to have them (the committed fixtures of tests/test_gpu_reference_fixtures.py still check the loop there) says so with DSRT_ALLOW_NO_REF=1."""
    if os.path.exists(path):
        return
    if os.environ.get("DSRT_ALLOW_NO_REF") == "1":
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built and DSRT_ALLOW_NO_REF=1")
    pytest.fail(f"{os.path.relpath(path, ROOT)} is missing: build it with `make -C oracle` where /root/reference exists (it travels with the gpurun snapshot), "
                "or set DSRT_ALLOW_NO_REF=1 to run without the live reference binaries")
