"""ctypes view of the C ABI in include/dsrt.h and the PODs of include/dsrt_scene_abi.h.

Nothing here computes anything: it declares the structs field for field (their offsets are asserted against the
reference's in tests/test_host_golden.py::test_abi_matches_reference_layout) and binds the entry points of libdsrt_hip.so.  The library is required:
importing this module without it raises -- there is no Python or CPU fallback for the render path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSRT_LIB", os.path.join(_HERE, "libdsrt_hip.so"))


class DsrtF3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class GPUTextureHeader(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("offset", C.c_int)]


class GPUMaterial(C.Structure):
    _fields_ = [("type", C.c_int), ("albedo_tex", C.c_int), ("_pad0", C.c_int), ("_pad1", C.c_int),
                ("albedo", DsrtF3), ("emissive", DsrtF3), ("fuzz", C.c_float), ("ref_idx", C.c_float)]


class GPUSphere(C.Structure):
    _fields_ = [("center", DsrtF3), ("radius", C.c_float), ("material_id", C.c_int), ("_pad", C.c_int)]


class GPUTriangle(C.Structure):
    _fields_ = [("v0", DsrtF3), ("v1", DsrtF3), ("v2", DsrtF3), ("n0", DsrtF3), ("n1", DsrtF3), ("n2", DsrtF3),
                ("uv0", DsrtF3), ("uv1", DsrtF3), ("uv2", DsrtF3), ("material_id", C.c_int), ("albedo_tex", C.c_int)]


class GPUBVHNode(C.Structure):
    _fields_ = [("bbox_min", DsrtF3), ("bbox_max", DsrtF3), ("left", C.c_int), ("right", C.c_int),
                ("tri_offset", C.c_int), ("tri_count", C.c_int)]


class GPURenderParams(C.Structure):
    _fields_ = [("img_width", C.c_int), ("img_height", C.c_int), ("samples_per_pixel", C.c_int), ("max_depth", C.c_int),
                ("use_bvh", C.c_int), ("rng_mode", C.c_int), ("tile_size", C.c_int), ("_pad0", C.c_int),
                ("gamma", C.c_float), ("exposure", C.c_float), ("env_rotation", C.c_float), ("_pad1", C.c_float)]


class GPUCamera(C.Structure):
    _fields_ = [("origin", DsrtF3), ("lower_left_corner", DsrtF3), ("horizontal", DsrtF3), ("vertical", DsrtF3),
                ("u", DsrtF3), ("v", DsrtF3), ("w", DsrtF3), ("lens_radius", C.c_float),
                ("image_width", C.c_int), ("image_height", C.c_int), ("samples_per_pixel", C.c_int), ("max_depth", C.c_int)]


class GPUScene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("num_spheres", C.c_int), ("_pad_sph0", C.c_int), ("_pad_sph1", C.c_int),
                ("triangles", C.c_void_p), ("tri_indices", C.c_void_p), ("num_triangles", C.c_int), ("_pad_geo", C.c_int),
                ("bvh_nodes", C.c_void_p), ("num_bvh_nodes", C.c_int),
                ("bvh_tri_indices", C.c_void_p),
                ("materials", C.c_void_p), ("num_materials", C.c_int), ("_pad_mat0", C.c_int), ("_pad_mat1", C.c_int),
                ("textures", C.c_void_p), ("num_textures", C.c_int), ("_pad_tex0", C.c_int), ("_pad_tex1", C.c_int),
                ("texture_pool", C.c_void_p), ("texture_pool_floats", C.c_int), ("_pad_pool0", C.c_int), ("_pad_pool1", C.c_int),
                ("camera", GPUCamera),
                ("sky_type", C.c_int), ("env_tex_id", C.c_int), ("_pad_sky0", C.c_int), ("_pad_sky1", C.c_int),
                ("sky_solid", DsrtF3), ("sky_top", DsrtF3), ("sky_bottom", DsrtF3),
                ("params", GPURenderParams),
                ("seed", C.c_uint64),
                ("sun_enabled", C.c_uint8), ("_pad_sun", C.c_uint8 * 3),
                ("sun_dir", DsrtF3), ("sun_radiance", DsrtF3)]


class DsrtPose(C.Structure):
    _fields_ = [("cam_pos_world", C.c_double * 3), ("model_pos_world", C.c_double * 3), ("model_euler_deg", C.c_float * 3)]


class DsrtFrame(C.Structure):
    _fields_ = [("cam_in_model", C.c_float * 3), ("sun_dir_model", C.c_float * 3), ("sep_m", C.c_double), ("skipped", C.c_int)]


class DsrtRenderDesc(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_depth", C.c_int), ("gamma", C.c_float),
                ("seed", C.c_uint64), ("rng_mode", C.c_int), ("tile_size", C.c_int), ("shard_rank", C.c_int),
                ("shard_count", C.c_int), ("collect_counters", C.c_int), ("checked", C.c_int), ("stack_entries", C.c_int),
                ("tune", C.c_int * 4), ("math_mode", C.c_int)]


class DsrtStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("waves_launched", C.c_int), ("device_flags", C.c_uint32), ("lds_stack_entries", C.c_int)] + \
               [(n, C.c_uint64) for n in ("samples", "rays", "primary_hits", "box_fetches", "nodes_entered", "internal_entered", "tri_tests",
                                          "hit_updates", "sphere_tests", "shaded_hits", "tex_fetches", "stack_spills", "max_stack",
                                          "node_slots", "tri_slots", "adv_slots", "adv_active", "idle_at_leaf", "idle_waiting", "idle_done", "visits_depth_lt6", "visits_depth_lt9", "visits_depth_lt12", "tiles_total", "tiles_culled", "wave_ticks", "certificate_fallbacks", "certificate_audited", "certificate_audit_mismatches")] + \
               [(n, C.c_float) for n in ("heavy_queue_empty_ms", "light_queue_empty_ms", "last_wave_exit_ms")] + [("certified_tree_used", C.c_int)]


# numpy record layouts of the reference arrays (for dumping / comparing with goldens)
F3 = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
TRI_DTYPE = np.dtype([("v", "<f4", (3, 3)), ("n", "<f4", (3, 3)), ("uv", "<f4", (3, 3)), ("material_id", "<i4"), ("albedo_tex", "<i4")])
NODE_DTYPE = np.dtype([("bbox_min", "<f4", 3), ("bbox_max", "<f4", 3), ("left", "<i4"), ("right", "<i4"), ("tri_offset", "<i4"), ("tri_count", "<i4")])
MAT_DTYPE = np.dtype([("type", "<i4"), ("albedo_tex", "<i4"), ("_pad", "<i4", 2), ("albedo", "<f4", 3), ("emissive", "<f4", 3), ("fuzz", "<f4"), ("ref_idx", "<f4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material_id", "<i4"), ("_pad", "<i4")])
TEXHDR_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<i4")])
assert TRI_DTYPE.itemsize == 116 and NODE_DTYPE.itemsize == 40 and MAT_DTYPE.itemsize == 48 and SPHERE_DTYPE.itemsize == 24

# The ABI version THESE hand-written structs were laid out for (include/dsrt.h, DSRT_ABI_VERSION).  load() requires library == header == this, and compares the sizes
# of the structs above with the library's own (dsrt_sizeof): a header and library bumped without this file are refused, not mis-laid.
ABI_VERSION = 7

EXPORTS = [
    "dsrt_last_error", "dsrt_abi_version", "dsrt_sizeof", "dsrt_microbench_copy", "dsrt_dev_set_experiment", "dsrt_selftest_poke_node_word", "dsrt_ctx_set_certified_tree", "dsrt_ctx_has_certified_tree", "dsrt_dropin_has_certified_tree", "dsrt_host_scene_second_tree_probe",
    "dsrt_host_scene_create", "dsrt_host_scene_destroy", "dsrt_host_scene_add_obj", "dsrt_host_scene_add_world_file",
    "dsrt_host_scene_add_arrays", "dsrt_host_scene_add_texture_file", "dsrt_host_scene_build_bvh", "dsrt_host_scene_build_bvh_sah", "dsrt_host_scene_build_bvh_gpu", "dsrt_host_scene_view", "dsrt_host_scene_bvh_stack_need", "dsrt_host_scene_texture_failures",
    "dsrt_scene_set_frame", "dsrt_read_pose_file", "dsrt_pose_to_frame", "dsrt_camera_look_at", "dsrt_decode_image_file", "dsrt_write_ppm", "dsrt_write_png",
    "dsrt_device_count", "dsrt_ctx_create", "dsrt_ctx_destroy", "dsrt_ctx_clone", "dsrt_ctx_device",
    "dsrt_multi_create", "dsrt_multi_destroy", "dsrt_multi_count", "dsrt_multi_uses_rccl", "dsrt_selftest_rccl_gather", "dsrt_multi_scene_upload", "dsrt_multi_render_frame", "dsrt_multi_render_sequence", "dsrt_scene_upload", "dsrt_scene_upload_device",
    "dsrt_scene_set_camera_sun", "dsrt_shard_layout", "dsrt_ctx_scene_bounds", "dsrt_render", "dsrt_render_batch", "dsrt_render_batch_to_host", "dsrt_deinterleave_tiles", "dsrt_deinterleave_batch", "dsrt_render_to_host",
    "dsrt_selftest_math", "dsrt_selftest_devkat", "dsrt_selftest_philox", "dsrt_microbench_gather", "dsrt_microbench_valu", "dsrt_microbench_valu_kinds", "dsrt_microbench_valu_kind_name", "gpu_render_scene", "dsrt_build_gpu_scene", "dsrt_free_gpu_scene",
]


def header_abi_version():
    """DSRT_ABI_VERSION as written in include/dsrt.h -- the one place the number lives; the library returns what it was compiled with."""
    import re
    try:
        with open(os.path.join(os.path.dirname(_HERE), "include", "dsrt.h")) as f:
            m = re.search(r"^#define\s+DSRT_ABI_VERSION\s+(\d+)", f.read(), re.M)
    except OSError as e:
        raise ImportError(f"include/dsrt.h cannot be read ({e}): this binding checks the library against the header it was written for") from e
    if not m:
        raise ImportError("include/dsrt.h does not define DSRT_ABI_VERSION")
    return int(m.group(1))


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()). "
            "The render path has no fallback; it is the HIP library or nothing.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p

    def sig(name, res, args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args

    sig("dsrt_last_error", C.c_char_p, [])
    sig("dsrt_abi_version", C.c_int, [])
    sig("dsrt_sizeof", C.c_size_t, [C.c_int])
    sig("dsrt_dev_set_experiment", C.c_int, [C.c_uint32])
    sig("dsrt_selftest_poke_node_word", C.c_int, [vp, C.c_size_t, C.c_uint32, P(C.c_uint32)])
    sig("dsrt_microbench_copy", C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, P(C.c_float), P(C.c_double)])
    sig("dsrt_host_scene_create", vp, [])
    sig("dsrt_host_scene_destroy", None, [vp])
    sig("dsrt_host_scene_add_obj", C.c_int, [vp, C.c_char_p, C.c_double])
    sig("dsrt_host_scene_add_world_file", C.c_int, [vp, C.c_char_p])
    sig("dsrt_host_scene_add_arrays", C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int])
    sig("dsrt_host_scene_add_texture_file", C.c_int, [vp, C.c_char_p, C.c_int])
    sig("dsrt_host_scene_build_bvh", C.c_int, [vp])
    sig("dsrt_host_scene_build_bvh_sah", C.c_int, [vp])
    sig("dsrt_host_scene_build_bvh_gpu", C.c_int, [vp, C.c_int, P(C.c_float), P(C.c_float)])
    sig("dsrt_host_scene_view", C.c_int, [vp, P(GPUScene)])
    sig("dsrt_host_scene_bvh_stack_need", C.c_int, [vp])
    sig("dsrt_host_scene_texture_failures", C.c_int, [vp, C.c_char_p, C.c_size_t])
    sig("dsrt_scene_set_frame", None, [P(GPUScene), P(GPUCamera), P(C.c_float)])
    sig("dsrt_read_pose_file", C.c_int, [C.c_char_p, P(DsrtPose), C.c_int, P(C.c_int)])
    sig("dsrt_pose_to_frame", C.c_int, [P(DsrtPose), P(DsrtFrame)])
    sig("dsrt_camera_look_at", C.c_int, [P(GPUCamera), P(C.c_float), P(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int, C.c_int])
    sig("dsrt_decode_image_file", C.c_int, [C.c_char_p, C.c_int, P(C.c_int), P(C.c_int), vp, C.c_size_t])
    sig("dsrt_write_ppm", C.c_int, [C.c_char_p, vp, C.c_int, C.c_int])
    sig("dsrt_write_png", C.c_int, [C.c_char_p, vp, C.c_int, C.c_int])
    sig("dsrt_device_count", C.c_int, [])
    sig("dsrt_ctx_create", C.c_int, [C.c_int, P(vp)])
    sig("dsrt_ctx_destroy", None, [vp])
    sig("dsrt_ctx_clone", C.c_int, [vp, P(vp)])
    sig("dsrt_ctx_device", C.c_int, [vp])
    sig("dsrt_ctx_set_certified_tree", C.c_int, [vp, C.c_int])
    sig("dsrt_ctx_has_certified_tree", C.c_int, [vp])
    sig("dsrt_dropin_has_certified_tree", C.c_int, [])
    sig("dsrt_host_scene_second_tree_probe", C.c_int, [vp, P(C.c_int), P(C.c_float), vp, vp, vp, C.c_int, vp, C.c_int])
    sig("dsrt_multi_create", C.c_int, [P(C.c_int), C.c_int, C.c_int, P(vp)])
    sig("dsrt_multi_destroy", None, [vp])
    sig("dsrt_multi_count", C.c_int, [vp])
    sig("dsrt_multi_uses_rccl", C.c_int, [vp])
    sig("dsrt_selftest_rccl_gather", C.c_int, [C.c_int, C.c_size_t])
    sig("dsrt_multi_scene_upload", C.c_int, [vp, P(GPUScene)])
    sig("dsrt_multi_render_frame", C.c_int, [vp, P(DsrtRenderDesc), P(GPUCamera), P(C.c_float), vp, P(C.c_float), P(C.c_double)])
    sig("dsrt_multi_render_sequence", C.c_int, [vp, P(DsrtRenderDesc), P(GPUCamera), P(C.c_float), C.c_int, P(vp), P(C.c_double)])
    sig("dsrt_scene_upload", C.c_int, [vp, P(GPUScene)])
    sig("dsrt_scene_upload_device", C.c_int, [vp, P(GPUScene)])
    sig("dsrt_scene_set_camera_sun", C.c_int, [vp, P(GPUCamera), P(C.c_float)])
    sig("dsrt_shard_layout", C.c_int, [P(DsrtRenderDesc), P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_size_t)])
    sig("dsrt_ctx_scene_bounds", C.c_int, [vp, P(C.c_float), P(C.c_float)])
    sig("dsrt_render", C.c_int, [vp, P(DsrtRenderDesc), vp, vp, vp, P(DsrtStats)])
    sig("dsrt_render_batch", C.c_int, [vp, P(DsrtRenderDesc), C.c_int, P(GPUCamera), P(C.c_float), vp, vp, vp, P(DsrtStats)])
    sig("dsrt_render_batch_to_host", C.c_int, [vp, P(DsrtRenderDesc), C.c_int, P(GPUCamera), P(C.c_float), vp, P(DsrtStats)])
    sig("dsrt_deinterleave_tiles", C.c_int, [vp, P(DsrtRenderDesc), vp, vp, vp])
    sig("dsrt_deinterleave_batch", C.c_int, [vp, P(DsrtRenderDesc), C.c_int, vp, vp, vp])
    sig("dsrt_render_to_host", C.c_int, [vp, P(DsrtRenderDesc), vp, vp, P(DsrtStats)])
    sig("dsrt_selftest_math", C.c_int, [vp, C.c_int, vp, C.c_float, vp, C.c_int])
    sig("dsrt_selftest_devkat", C.c_int, [vp, C.c_int, vp, vp, C.c_int])
    sig("dsrt_selftest_philox", C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp])
    sig("dsrt_microbench_gather", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, P(C.c_float), P(C.c_double)])
    sig("dsrt_microbench_valu", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, P(C.c_float), P(C.c_double), P(C.c_double)])
    sig("dsrt_microbench_valu_kinds", C.c_int, [])
    sig("dsrt_microbench_valu_kind_name", C.c_char_p, [C.c_int])
    sig("gpu_render_scene", None, [P(GPUScene), C.c_int, C.c_int])
    sig("dsrt_build_gpu_scene", C.c_int, [vp, P(GPUCamera), P(C.c_float), P(GPUScene)])
    sig("dsrt_free_gpu_scene", None, [P(GPUScene)])
    have, want = lib.dsrt_abi_version(), header_abi_version()
    if not (have == want == ABI_VERSION):
        raise ImportError(f"ABI mismatch: {LIB_PATH} was built for {have}, include/dsrt.h says {want}, capi.py's structs are laid out for {ABI_VERSION}: "
                          "rebuild with `make lib` / update capi.py")
    for which, struct in enumerate((DsrtRenderDesc, DsrtStats, GPUScene, GPUCamera, DsrtPose, DsrtFrame)):      # DSRT_SIZEOF_* of include/dsrt.h, in order
        if lib.dsrt_sizeof(which) != C.sizeof(struct):
            raise ImportError(f"capi.py lays {struct.__name__} out in {C.sizeof(struct)} bytes, the library in {lib.dsrt_sizeof(which)}: update capi.py")
    return lib
