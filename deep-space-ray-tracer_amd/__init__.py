"""Python plumbing over libdsrt_hip.so (tests, bench and the multi-GPU launcher use it).

The product is the shared library and its C ABI (include/dsrt.h); this package only moves pointers and sizes across
that boundary.  It is imported as `dsrt_amd` (see dsrt_amd.py at the repository root: the directory name mandated for
the package contains hyphens and cannot be written in an import statement).
"""
import ctypes as C

import numpy as np

try:
    # PyTorch-ROCm ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both see the GPU.  Importing torch
    # first makes the dynamic loader bind libdsrt_hip.so to that same runtime (same SONAME).  torch is plumbing only
    # (device buffers, streams, torch.distributed in bench.py); the library itself has no torch dependency.
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None

from . import capi
from .capi import (DsrtFrame, DsrtPose, DsrtRenderDesc, DsrtStats, GPUCamera, GPUScene)

lib = capi.load()


class DsrtError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.dsrt_last_error()
        super().__init__(f"{where} failed with {code}: {msg.decode() if msg else ''}")


def _check(rc, where):
    if rc != 0:
        raise DsrtError(rc, where)


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


class HostScene:
    """Flattened scene + BVH on the host (DsrtHostScene)."""

    def __init__(self):
        self._h = lib.dsrt_host_scene_create()
        self._built = False

    def close(self):
        if self._h and lib is not None:
            lib.dsrt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_obj(self, path, scale=1.0):
        _check(lib.dsrt_host_scene_add_obj(self._h, str(path).encode(), float(scale)), "dsrt_host_scene_add_obj")
        self._built = False
        return self

    def add_world_file(self, path):
        _check(lib.dsrt_host_scene_add_world_file(self._h, str(path).encode()), "dsrt_host_scene_add_world_file")
        self._built = False
        return self

    def add_arrays(self, tris=None, spheres=None, mats=None):
        def arr(a, dt):
            a = np.ascontiguousarray(a if a is not None else np.zeros(0, dt), dtype=dt)
            return a, (a.ctypes.data if a.size else None), int(a.size)
        t, tp, tn = arr(tris, capi.TRI_DTYPE)
        s, sp, sn = arr(spheres, capi.SPHERE_DTYPE)
        m, mp, mn = arr(mats, capi.MAT_DTYPE)
        _check(lib.dsrt_host_scene_add_arrays(self._h, tp, tn, sp, sn, mp, mn), "dsrt_host_scene_add_arrays")
        self._built = False
        return self

    def add_texture_file(self, path, flip_vertically=True):
        """Texture slot of an image file (decoded into the scene's pool as the reference's builder decodes map_Kd files); the id a triangle added with
        add_arrays carries in albedo_tex.  flip_vertically: the state of the reference's global stb flag (True once any MTL named a map)."""
        slot = lib.dsrt_host_scene_add_texture_file(self._h, str(path).encode(), 1 if flip_vertically else 0)
        if slot < 0:
            _check(slot, "dsrt_host_scene_add_texture_file")
        return slot

    def build_bvh(self, kind="median"):
        """kind "median": the reference's tree (parity); "sah": binned-SAH tree, "lbvh": linear BVH built on the GPU -- non-parity fast modes."""
        if kind == "median":
            _check(lib.dsrt_host_scene_build_bvh(self._h), "dsrt_host_scene_build_bvh")
        elif kind == "sah":
            _check(lib.dsrt_host_scene_build_bvh_sah(self._h), "dsrt_host_scene_build_bvh_sah")
        elif kind == "lbvh":                         # built on the GPU (device 0 unless set_lbvh_device says otherwise)
            ms, tot = C.c_float(), C.c_float()
            _check(lib.dsrt_host_scene_build_bvh_gpu(self._h, int(getattr(self, "lbvh_device", 0)), C.byref(ms), C.byref(tot)), "dsrt_host_scene_build_bvh_gpu")
            self.lbvh_build_ms, self.lbvh_total_ms = ms.value, tot.value
        else:
            raise ValueError("build_bvh kind must be 'median', 'sah' or 'lbvh'")
        self._built = True
        return self

    @property
    def texture_failures(self):
        """Paths of texture maps that could not be decoded (their texel is the reference's 1x1 white fallback)."""
        buf = C.create_string_buffer(1 << 16)
        n = lib.dsrt_host_scene_texture_failures(self._h, buf, len(buf))
        return [p for p in buf.value.decode(errors="replace").split("\n") if p][:n] if n else []

    @property
    def stack_need(self):
        return lib.dsrt_host_scene_bvh_stack_need(self._h)

    def view(self, camera=None, sun_dir=None):
        """GPUScene with HOST pointers (valid while this object lives and is not modified)."""
        if not self._built:
            self.build_bvh()
        s = GPUScene()
        _check(lib.dsrt_host_scene_view(self._h, C.byref(s)), "dsrt_host_scene_view")
        if camera is not None:
            lib.dsrt_scene_set_frame(C.byref(s), C.byref(camera), _f3(sun_dir if sun_dir is not None else (0.0, -1.0, 0.0)))
        return s

    def arrays(self):
        """numpy copies of the flattened arrays, in the reference layouts."""
        s = self.view()

        def grab(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).copy()
        return {
            "tris": grab(s.triangles, s.num_triangles, capi.TRI_DTYPE),
            "spheres": grab(s.spheres, s.num_spheres, capi.SPHERE_DTYPE),
            "mats": grab(s.materials, s.num_materials, capi.MAT_DTYPE),
            "idx": grab(s.tri_indices, s.num_triangles if s.tri_indices else 0, np.dtype("<i4")),
            "nodes": grab(s.bvh_nodes, s.num_bvh_nodes, capi.NODE_DTYPE),
            "texhdr": grab(s.textures, s.num_textures, capi.TEXHDR_DTYPE),
            "texpool": grab(s.texture_pool, s.texture_pool_floats, np.dtype("<f4")),
        }


def read_pose_file(path):
    n = C.c_int(0)
    rc = lib.dsrt_read_pose_file(str(path).encode(), None, 0, C.byref(n))
    _check(rc, "dsrt_read_pose_file")
    poses = (DsrtPose * n.value)()
    _check(lib.dsrt_read_pose_file(str(path).encode(), poses, n.value, C.byref(n)), "dsrt_read_pose_file")
    return list(poses)


def pose_to_frame(pose):
    f = DsrtFrame()
    _check(lib.dsrt_pose_to_frame(C.byref(pose), C.byref(f)), "dsrt_pose_to_frame")
    return f


def camera_look_at(lookfrom, lookat, vfov, width, height, spp, max_depth):
    cam = GPUCamera()
    _check(lib.dsrt_camera_look_at(C.byref(cam), _f3(lookfrom), _f3(lookat), float(vfov), int(width), int(height), int(spp), int(max_depth)),
           "dsrt_camera_look_at")
    return cam


def frame_camera(frame, vfov, width, height, spp, max_depth):
    """Camera for one pose frame: at cam_in_model, looking at the model origin (src/main.cpp:399)."""
    return camera_look_at(tuple(frame.cam_in_model), (0.0, 0.0, 0.0), vfov, width, height, spp, max_depth)


def decode_image_file(path, flip_vertically=False):
    """The builder's texture decoder on one file -> H x W x 3 uint8 (dsrt_decode_image_file)."""
    w, h = C.c_int(), C.c_int()
    _check(lib.dsrt_decode_image_file(str(path).encode(), int(bool(flip_vertically)), C.byref(w), C.byref(h), None, 0), "dsrt_decode_image_file")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    _check(lib.dsrt_decode_image_file(str(path).encode(), int(bool(flip_vertically)), C.byref(w), C.byref(h), out.ctypes.data, out.size), "dsrt_decode_image_file")
    return out


def write_ppm(path, rgb, width, height):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(lib.dsrt_write_ppm(str(path).encode(), rgb.ctypes.data, int(width), int(height)), "dsrt_write_ppm")


def write_png(path, rgb, width, height):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(lib.dsrt_write_png(str(path).encode(), rgb.ctypes.data, int(width), int(height)), "dsrt_write_png")


def make_desc(width, height, spp, max_depth=50, gamma=2.0, seed=1337, tile_size=0, shard_rank=0, shard_count=0,
              collect_counters=0, checked=0, stack_entries=0, tune=(0, 0, 0, 0), rng_mode=0, math_mode=0):
    d = DsrtRenderDesc()
    d.width, d.height, d.spp, d.max_depth, d.gamma, d.seed = int(width), int(height), int(spp), int(max_depth), float(gamma), int(seed)
    d.rng_mode = int(rng_mode)
    d.math_mode = int(math_mode)
    d.tile_size, d.shard_rank, d.shard_count = int(tile_size), int(shard_rank), int(shard_count)
    d.collect_counters, d.checked, d.stack_entries = int(collect_counters), int(checked), int(stack_entries)
    for i, v in enumerate(tuple(tune) + (0,) * (4 - len(tune))):
        d.tune[i] = int(v)
    return d


def shard_layout(desc):
    total, mine, padded, nbytes = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    _check(lib.dsrt_shard_layout(C.byref(desc), C.byref(total), C.byref(mine), C.byref(padded), C.byref(nbytes)), "dsrt_shard_layout")
    return {"tiles_total": total.value, "tiles_this_shard": mine.value, "tiles_per_shard_padded": padded.value, "rgb8_bytes_padded": nbytes.value}


def selftest_rccl_gather(device=0, nbytes=1 << 20):
    """dsrt_selftest_rccl_gather: a one-rank RCCL communicator and one checked ncclGather on `device`."""
    _check(lib.dsrt_selftest_rccl_gather(int(device), int(nbytes)), "dsrt_selftest_rccl_gather")


def set_experiment(word):
    """Development switches of the A/B tools (include/dsrt.h, dsrt_dev_set_experiment): process-wide, undefined bits refused."""
    _check(lib.dsrt_dev_set_experiment(int(word) & 0xFFFFFFFF), "dsrt_dev_set_experiment")


def microbench_copy(nbytes=2 << 30, blocks_per_cu=8, reps=8, device=0, mode=0):
    """HBM streaming calibration (include/dsrt.h): float4 grid-stride kernel; mode 0 copy (GB/s counts bytes read + written), 1 read only, 2 write only, 3 non-temporal copy."""
    ms, moved = C.c_float(), C.c_double()
    _check(lib.dsrt_microbench_copy(int(device), int(mode), int(nbytes), int(blocks_per_cu), int(reps), C.byref(ms), C.byref(moved)), "dsrt_microbench_copy")
    return {"mode": ("copy", "read only", "write only", "non-temporal copy")[mode], "bytes_per_buffer": int(nbytes), "blocks_per_cu": blocks_per_cu, "reps": reps, "ms": ms.value,
            "GBps": moved.value / ms.value / 1e6}


def microbench_gather(mode=0, dependent=False, live_lanes=64, pad_valu=0, table_bytes=19 << 20, iters=2000, device=0):
    """Gather-rate calibration kernel (include/dsrt.h): returns {"ms", "records", "Grecords_per_s"}."""
    ms, rec = C.c_float(), C.c_double()
    _check(lib.dsrt_microbench_gather(int(device), int(mode), int(bool(dependent)), int(live_lanes), int(pad_valu), int(table_bytes), int(iters),
                                      C.byref(ms), C.byref(rec)), "dsrt_microbench_gather")
    return {"mode": mode, "dependent": bool(dependent), "live_lanes": live_lanes, "pad_valu": pad_valu, "table_MB": table_bytes / 2**20, "iters": iters,
            "ms": ms.value, "records": rec.value, "Grecords_per_s": rec.value / ms.value / 1e6}


VALU_KINDS = tuple(lib.dsrt_microbench_valu_kind_name(k).decode() for k in range(lib.dsrt_microbench_valu_kinds()))


def microbench_valu(kind=0, waves_per_simd=8, iters=20000, lane_mask=(1 << 64) - 1, pattern=0, device=0, simds=1024):
    """VALU issue-cost calibration (include/dsrt.h).  `kind`: index or name (VALU_KINDS).  cycles_per_instruction_per_simd is for the stream as issued:
    with pattern 1 or 2 half of the instructions are v_add_f32."""
    if isinstance(kind, str):
        kind = VALU_KINDS.index(kind)
    ms, n, ghz = C.c_float(), C.c_double(), C.c_double()
    _check(lib.dsrt_microbench_valu(int(device), int(kind), int(pattern), int(waves_per_simd), int(iters), int(lane_mask), C.byref(ms), C.byref(n), C.byref(ghz)),
           "dsrt_microbench_valu")
    gips = n.value / ms.value / 1e6
    return {"kind": VALU_KINDS[kind], "pattern": ("x32", "alternating with v_add_f32", "pairs between pairs of v_add_f32", "x16 then v_add_f32 x16", "alternating with v_pk_mul_f32", "x16 then v_pk_mul_f32 x16")[pattern], "waves_per_simd": waves_per_simd, "iters": iters,
            "lanes": bin(lane_mask).count("1"), "ms": ms.value, "wave_instructions": n.value, "G_wave_instructions_per_s": gips, "shader_clock_GHz": ghz.value,
            "cycles_per_instruction_per_simd": simds * ghz.value / gips}


def stats_dict(st):
    return {name: getattr(st, name) for name, _ in DsrtStats._fields_}


class Context:
    """One GPU: resident scene in traversal layout + render launches (DsrtContext)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib.dsrt_ctx_create(int(device), C.byref(h)), "dsrt_ctx_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.dsrt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def clone(self):
        """A second context on the same device sharing this one's resident scene (dsrt_ctx_clone)."""
        h = C.c_void_p()
        _check(lib.dsrt_ctx_clone(self._h, C.byref(h)), "dsrt_ctx_clone")
        other = Context.__new__(Context)
        other._h, other.device = h, self.device
        return other

    def set_certified_tree(self, on=True):
        """The next upload also builds the certified second tree (include/dsrt.h): rays walk a SAH tree, the kernel certifies every answer against the reference tree."""
        _check(lib.dsrt_ctx_set_certified_tree(self._h, 1 if on else 0), "dsrt_ctx_set_certified_tree")
        return self

    @property
    def has_certified_tree(self):
        return bool(lib.dsrt_ctx_has_certified_tree(self._h))

    def upload(self, scene_host_view):
        _check(lib.dsrt_scene_upload(self._h, C.byref(scene_host_view)), "dsrt_scene_upload")

    def upload_device(self, scene_device_view):
        _check(lib.dsrt_scene_upload_device(self._h, C.byref(scene_device_view)), "dsrt_scene_upload_device")

    def set_camera_sun(self, camera, sun_dir):
        _check(lib.dsrt_scene_set_camera_sun(self._h, C.byref(camera), _f3(sun_dir)), "dsrt_scene_set_camera_sun")

    def render(self, desc, d_rgb8_ptr, d_f32_ptr=None, stream=None, want_stats=False):
        """Launch on `stream` (raw hipStream_t as int); with want_stats the call synchronises and returns DsrtStats."""
        st = DsrtStats() if want_stats else None
        rc = lib.dsrt_render(self._h, C.byref(desc), C.c_void_p(d_rgb8_ptr), C.c_void_p(d_f32_ptr) if d_f32_ptr else None,
                             C.c_void_p(stream) if stream else None, C.byref(st) if st is not None else None)
        _check(rc, "dsrt_render")
        return st

    def render_batch(self, desc, cameras, sun_dirs, d_rgb8_ptr, d_f32_ptr=None, stream=None, want_stats=False):
        """dsrt_render_batch: len(cameras) views of the resident scene as one launch; the images land one after another in d_rgb8."""
        n = len(cameras)
        cams = (GPUCamera * n)(*cameras)
        suns = (C.c_float * (3 * n))(*[float(v) for s in sun_dirs for v in s])
        st = DsrtStats() if want_stats else None
        rc = lib.dsrt_render_batch(self._h, C.byref(desc), n, cams, suns, C.c_void_p(d_rgb8_ptr), C.c_void_p(d_f32_ptr) if d_f32_ptr else None,
                                   C.c_void_p(stream) if stream else None, C.byref(st) if st is not None else None)
        _check(rc, "dsrt_render_batch")
        return st

    def deinterleave(self, desc, d_gathered_ptr, d_image_ptr, stream=None):
        _check(lib.dsrt_deinterleave_tiles(self._h, C.byref(desc), C.c_void_p(d_gathered_ptr), C.c_void_p(d_image_ptr),
                                           C.c_void_p(stream) if stream else None), "dsrt_deinterleave_tiles")

    def deinterleave_batch(self, desc, frames, d_gathered_ptr, d_images_ptr, stream=None):
        """dsrt_deinterleave_batch: the gathered compact buffers of sharded batch launches -> `frames` whole images."""
        _check(lib.dsrt_deinterleave_batch(self._h, C.byref(desc), int(frames), C.c_void_p(d_gathered_ptr), C.c_void_p(d_images_ptr),
                                           C.c_void_p(stream) if stream else None), "dsrt_deinterleave_batch")

    def render_to_host(self, desc, want_f32=False):
        n = desc.width * desc.height * 3
        rgb = np.zeros(n, np.uint8)
        f32 = np.zeros(n, np.float32) if want_f32 else None
        st = DsrtStats()
        rc = lib.dsrt_render_to_host(self._h, C.byref(desc), rgb.ctypes.data, f32.ctypes.data if want_f32 else None, C.byref(st))
        _check(rc, "dsrt_render_to_host")
        shape = (desc.height, desc.width, 3)
        return rgb.reshape(shape), (f32.reshape(shape) if want_f32 else None), st

    def poke_node_word(self, word_index, value):
        """Test hook (dsrt_selftest_poke_node_word): overwrite one 32-bit word of the resident node records; returns the previous value."""
        old = C.c_uint32()
        _check(lib.dsrt_selftest_poke_node_word(self._h, int(word_index), int(value) & 0xFFFFFFFF, C.byref(old)), "dsrt_selftest_poke_node_word")
        return old.value

    def selftest_philox(self, seed, subsequence, n):
        ours, theirs = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(lib.dsrt_selftest_philox(self._h, int(seed), int(subsequence), int(n), ours.ctypes.data, theirs.ctypes.data), "dsrt_selftest_philox")
        return ours, theirs

    def selftest_devkat(self, fn, in12):
        """dsrt_selftest_devkat: the kernel's material / frame helpers on n x 12 input words -> n x 12 output words (float32 arrays)."""
        a = np.ascontiguousarray(in12, np.float32).reshape(-1, 12)
        out = np.zeros_like(a)
        _check(lib.dsrt_selftest_devkat(self._h, int(fn), a.ctypes.data, out.ctypes.data, int(a.shape[0])), "dsrt_selftest_devkat")
        return out

    def selftest_math(self, fn, x, y=0.0):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        _check(lib.dsrt_selftest_math(self._h, int(fn), x.ctypes.data, float(y), out.ctypes.data, int(x.size)), "dsrt_selftest_math")
        return out


class Multi:
    """All GPUs of a node from one process (DsrtMulti): a frame, or every frame of a sequence, sharded by interleaved screen tiles; one RCCL gather
    per frame (render_frame) or per batch launch (render_sequence)."""

    def __init__(self, devices, frames_in_flight=1):
        devs = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = C.c_void_p()
        _check(lib.dsrt_multi_create(devs, len(devices), int(frames_in_flight), C.byref(h)), "dsrt_multi_create")
        self._h, self.n = h, len(devices)

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.dsrt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def uses_rccl(self):
        return bool(lib.dsrt_multi_uses_rccl(self._h))

    def upload(self, scene_host_view):
        _check(lib.dsrt_multi_scene_upload(self._h, C.byref(scene_host_view)), "dsrt_multi_scene_upload")

    def render_frame(self, desc, camera, sun_dir):
        """-> (rgb8 image H x W x 3, per-rank kernel ms, wall seconds)"""
        img = np.zeros((desc.height, desc.width, 3), np.uint8)
        ms = (C.c_float * self.n)()
        sec = C.c_double()
        _check(lib.dsrt_multi_render_frame(self._h, C.byref(desc), C.byref(camera), _f3(sun_dir), img.ctypes.data, ms, C.byref(sec)), "dsrt_multi_render_frame")
        return img, list(ms), sec.value

    def render_sequence(self, desc, cameras, sun_dirs, want_images=True):
        """Every rank renders its tiles of every frame as sharded batch launches (dsrt_multi_render_sequence).  -> (list of images or None, wall seconds)"""
        n = len(cameras)
        cams = (GPUCamera * n)(*cameras)
        suns = (C.c_float * (3 * n))(*[float(v) for s3 in sun_dirs for v in s3])
        imgs = [np.zeros((desc.height, desc.width, 3), np.uint8) for _ in range(n)] if want_images else None
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs]) if want_images else None
        sec = C.c_double()
        _check(lib.dsrt_multi_render_sequence(self._h, C.byref(desc), cams, suns, n, ptrs, C.byref(sec)), "dsrt_multi_render_sequence")
        return imgs, sec.value
