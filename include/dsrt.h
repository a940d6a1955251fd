/*
 * dsrt.h -- C ABI of libdsrt_hip.so, the MI355X-native renderer behind the Deep-Space-Ray-Tracer
 * host API.  Plain C types only: pointers, sizes, PODs from dsrt_scene_abi.h.  No torch types, no
 * C++ classes.  Every entry point names the reference interface it stands in for
 * (file:line relative to the reference repository).
 *
 * Error model: every int-returning function returns DSRT_OK (0) or a negative DSRT_ERR_*; the text
 * of the last failure on the calling thread is available from dsrt_last_error().  The reference
 * prints to stderr and returns void (src/gpu_render.cu:1053-1094, src/gpu_scene_builder.cpp:27-34);
 * the drop-in wrappers at the bottom keep that behaviour on top of these functions.
 *
 * Threading: a DsrtHostScene or DsrtContext may be used by one thread at a time.
 */
#ifndef DSRT_H
#define DSRT_H

#include "dsrt_scene_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DSRT_OK               0
#define DSRT_ERR_INVALID     -1   /* bad argument                                              */
#define DSRT_ERR_IO          -2   /* file could not be read / written                          */
#define DSRT_ERR_HIP         -3   /* a HIP runtime call failed                                 */
#define DSRT_ERR_NO_DEVICE   -4   /* no usable gfx950 device                                   */
#define DSRT_ERR_BVH_DEPTH   -5   /* BVH deeper than the 64-entry stack of src/gpu_render.cu:399 */
#define DSRT_ERR_NO_SCENE    -6   /* render before upload                                      */
#define DSRT_ERR_DEVICE_FLAG -7   /* the kernel raised its checked-mode status word            */
#define DSRT_ERR_NOMEM       -8   /* host allocation failed (no C++ exception crosses this ABI) */
#define DSRT_ERR_COMM        -9   /* an RCCL call failed                                         */

const char* dsrt_last_error(void);
/* ABI version: THE one place it is written.  Bumped on any signature, struct or flag change (3 = round 2: DsrtStats grew,
 * dsrt_render_batch, dsrt_multi_*; 4 = round 3: DsrtRenderDesc.tune[3] pruned to the switches a host may need, reserved bits
 * refused; dsrt_selftest_devkat, dsrt_microbench_valu; 5 = DsrtRenderDesc.math_mode appended;
 * 6 = dsrt_host_scene_add_texture_file; 7 = round 4: dsrt_microbench_copy, dsrt_sizeof, dsrt_dev_set_experiment, dsrt_selftest_poke_node_word, dsrt_ctx_set_certified_tree, DsrtStats grew).  dsrt_abi_version() returns the value the library was compiled with;
 * bindings parse this line (capi.header_abi_version) and compare. */
#define DSRT_ABI_VERSION 7
int dsrt_abi_version(void);
/* sizeof of the structs that cross this ABI, as the LIBRARY was compiled: a binding that mirrors them by hand (ctypes, cgo, JNA ...) compares its own sizes with
 * these at load time, so that a struct that grew in the header and the library but not in the binding is refused instead of silently mis-laid.  0 = unknown. */
#define DSRT_SIZEOF_RENDER_DESC 0
#define DSRT_SIZEOF_STATS       1
#define DSRT_SIZEOF_GPU_SCENE   2
#define DSRT_SIZEOF_GPU_CAMERA  3
#define DSRT_SIZEOF_POSE        4
#define DSRT_SIZEOF_FRAME       5
size_t dsrt_sizeof(int which);

/* ===================================================================================== */
/* Host scene assembly -- no GPU involved.                                               */
/* Replaces the flattening half of build_gpu_scene (src/gpu_scene_builder.cpp:464-601):  */
/* collect_world :252-317, upsert_material :71-139, HostTextureRegistry :199-246,        */
/* build_bvh_for_triangles :343-459, and the OBJ/MTL loader inc/triangle_mesh.h:75-255.  */
/* ===================================================================================== */
typedef struct DsrtHostScene DsrtHostScene;

DsrtHostScene* dsrt_host_scene_create(void);
void           dsrt_host_scene_destroy(DsrtHostScene* hs);

/* Append the triangles of an OBJ file (with its MTL materials / map_Kd textures), exactly as
 * `world.add(make_shared<triangle_mesh>(path, lambertian(0.73), scale))` followed by the builder's
 * collect step would (src/main.cpp:238-245, src/gpu_scene_builder.cpp:259-283). */
int dsrt_host_scene_add_obj(DsrtHostScene* hs, const char* obj_path, double scale);

/* Append objects described in the small "world description" text format used by the tests and the
 * CLI (mat / sphere / tri / obj lines).  Objects are flattened in file order. */
int dsrt_host_scene_add_world_file(DsrtHostScene* hs, const char* world_path);

/* Append already-flattened primitives (reference layouts).  Material ids are rebased onto the
 * scene's material table. */
int dsrt_host_scene_add_arrays(DsrtHostScene* hs, const GPUTriangle* tris, int num_tris, const GPUSphere* spheres,
                               int num_spheres, const GPUMaterial* mats, int num_mats);

/* Decode an image file into the scene's texture pool the way the reference's builder does (HostTextureRegistry::get_or_load,
 * src/gpu_scene_builder.cpp:205-246: forced RGB, powf(c / 255, 2.2) per channel, one slot per distinct path, a file that cannot be
 * decoded becomes one white texel and is listed by dsrt_host_scene_texture_failures) and return its slot: the value a GPUTriangle's
 * albedo_tex must carry when triangles are added with dsrt_host_scene_add_arrays.  `flip_vertically` is the state of the reference's
 * global stb flag at that moment (image_texture::load sets it, inc/texture.h:133; SURVEY.md note T): 1 for any scene whose MTL named a map.
 * Returns the slot (>= 0) or a negative DSRT_ERR_*. */
int dsrt_host_scene_add_texture_file(DsrtHostScene* hs, const char* path, int flip_vertically);

/* Median-split BVH over all triangles added so far: leaf <= 4, std::nth_element on the centroid along
 * the widest centroid axis, pre-order node numbering (src/gpu_scene_builder.cpp:343-459). */
int dsrt_host_scene_build_bvh(DsrtHostScene* hs);

/* NOT the reference's tree: a binned surface-area-heuristic BVH in the same node format (leaf <= 4, pre-order), the
 * "non-parity fast mode" of SURVEY.md 8(f) n4.  Rays visit far fewer nodes; the frame is a statistically equivalent image
 * (ties between equal-distance hits and float grazing cases resolve differently), not the reference's bytes.  There is no
 * reference interface this replaces -- the reference has one builder (src/gpu_scene_builder.cpp:343-459). */
int dsrt_host_scene_build_bvh_sah(DsrtHostScene* hs);

/* NOT the reference's tree either: a linear BVH (Morton order + Karras' radix-tree construction) built ON THE GPU `device`, in the same
 * node format (leaf <= 4), copied into the host scene.  The "GPU-side BVH build" of SURVEY.md 8(f) n4: milliseconds instead of the
 * seconds of the host builders, for hosts that rebuild per frame as the reference does (src/main.cpp:405).  Same parity status as the
 * SAH tree: a statistically equivalent image, exact agreement between the kernel and the oracle on this tree.  `build_ms` (optional)
 * receives the device time of the construction kernels, `total_ms` the whole call including the triangle upload and the copy back. */
int dsrt_host_scene_build_bvh_gpu(DsrtHostScene* hs, int device, float* build_ms, float* total_ms);

/* Fill `out` with HOST pointers into the scene's arrays (valid until the scene is modified or
 * destroyed).  Camera / params / sun fields of `out` are zeroed; set them with dsrt_scene_set_frame. */
int dsrt_host_scene_view(const DsrtHostScene* hs, GPUScene* out);

/* Textures that could not be decoded while objects were added.  Like the reference (stbi_load failure, src/gpu_scene_builder.cpp:216-221)
 * the builder goes on with a 1x1 white texel for such a map and prints a warning -- but the reference's stb_image also decodes BMP,
 * TGA, GIF, PSD, HDR, PIC, interlaced PNG and CMYK JPEG, while this library decodes PNM, non-interlaced PNG and 1- or 3-component JPEG only.  A scene for which this returns
 * non-zero therefore does NOT render like the reference would; hosts that care should refuse it (dsrt_render --strict-textures does,
 * bench.py labels the run).  Returns the number of failed maps; if `names` is given, their paths, newline-separated, as far as `cap` allows. */
int dsrt_host_scene_texture_failures(const DsrtHostScene* hs, char* names, size_t cap);

/* Greatest number of entries the reference traversal's stack can hold for this BVH (= height - 1). */
int dsrt_host_scene_bvh_stack_need(const DsrtHostScene* hs);

/* Defaults of build_gpu_scene for everything that is not geometry (src/gpu_scene_builder.cpp:560-598):
 * camera, params {gamma 2, exposure 50, use_bvh 1, rng_mode 0, tile_size 0}, black sky, seed 1337,
 * sun enabled with radiance (1e5, 9.5e4, 9e4). */
void dsrt_scene_set_frame(GPUScene* scene, const GPUCamera* cam, const float sun_dir_model[3]);

/* ===================================================================================== */
/* Pose file and camera -- src/main.cpp:139-173 (reader), :310-357 (world->model), :178-187 and
 * inc/camera.h:91-133 (camera).                                                          */
/* ===================================================================================== */
typedef struct DsrtPose {
    double cam_pos_world[3];
    double model_pos_world[3];
    float  model_euler_deg[3];      /* yaw, pitch, roll -- stored as float like PoseEntry, main.cpp:95-99 */
} DsrtPose;

typedef struct DsrtFrame {
    float  cam_in_model[3];
    float  sun_dir_model[3];
    double sep_m;
    int    skipped;                 /* sep_m < 1.0: the reference skips the frame, main.cpp:342-345 */
} DsrtFrame;

/* Reads up to `cap` poses; *count receives the number of valid lines in the file (may exceed cap).
 * Returns DSRT_ERR_IO if the file cannot be opened or holds no valid pose (read_pose_file returns false). */
int dsrt_read_pose_file(const char* path, DsrtPose* out, int cap, int* count);
int dsrt_pose_to_frame(const DsrtPose* pose, DsrtFrame* out);
/* point_camera_at + camera::initialize + toGPUCamera: vup (0,1,0), aperture 0, focus = |from - at|. */
int dsrt_camera_look_at(GPUCamera* out, const float from[3], const float at[3], float vfov_deg, int width, int height,
                        int spp, int max_depth);

/* Decode a texture file exactly as the scene builder does (PNM, non-interlaced PNG, baseline / progressive JPEG; forced to 3 channels like the
 * reference's stbi_load(..., 3), src/gpu_scene_builder.cpp:215; `flip_vertically` as stbi_set_flip_vertically_on_load).  Pass rgb = NULL to ask
 * for the size only.  DSRT_ERR_IO if the file cannot be read or is in a format this library does not decode. */
int dsrt_decode_image_file(const char* path, int flip_vertically, int* width, int* height, uint8_t* rgb, size_t cap);

/* P6 writer, as the tail of gpu_render_scene (src/gpu_render.cu:1099-1107). */
int dsrt_write_ppm(const char* path, const uint8_t* rgb, int width, int height);
/* 8-bit RGB PNG: what the reference obtains by shelling out to ImageMagick on the PPM (src/main.cpp:28-36). */
int dsrt_write_png(const char* path, const uint8_t* rgb, int width, int height);

/* ===================================================================================== */
/* Device side.                                                                          */
/* ===================================================================================== */
typedef struct DsrtContext DsrtContext;

/* The first call of either of these two sets the environment variable GPU_MAX_HW_QUEUES to 16 if the host has not set it (frames in flight
 * on separate streams only overlap on the device if each stream has a hardware queue of its own, and the HIP runtime reads the variable
 * when it initialises): a host that minds sets the variable itself, or initialises HIP, before calling.  Loading the library changes nothing. */
int  dsrt_device_count(void);
int  dsrt_ctx_create(int device, DsrtContext** out);
void dsrt_ctx_destroy(DsrtContext* ctx);

/* A second context on the same device that SHARES `src`'s resident scene (no copy, no re-upload) and has its own camera / sun
 * and its own working buffers: what a host needs to keep several frames of one scene in flight on separate streams.  A context
 * serves one render at a time (a second dsrt_render on it waits for the first, on any stream).  Uploading a new scene into
 * either context afterwards does not affect the other.  No reference counterpart (the reference renders one frame at a time). */
int  dsrt_ctx_clone(const DsrtContext* src, DsrtContext** out);
int  dsrt_ctx_device(const DsrtContext* ctx);

/* THE CERTIFIED SECOND TREE (round 4).  The reference's answer to a BVH query depends on its own median-split tree only in three narrow ways: triangles under
 * a zero-thickness box are never reached (src/gpu_render.cu:312), of two accepted triangles at exactly the same t the one tested later wins (:353), and a hit
 * computed in front of its own leaf box's entry distance is found or not depending on what was found before.  Everywhere else the answer is simply the accepted
 * triangle of smallest t -- whatever tree found it.  With this option the next dsrt_scene_upload also builds a binned-SAH tree over the reachable triangles
 * (boxes widened by 2^-16 of the scene's extent); a ray walks THAT tree (a third fewer node visits on a mesh of long thin members), the kernel then checks the
 * three conditions exactly -- no tie, the hit inside its REFERENCE-leaf box's slab interval as the reference's own arithmetic computes it, no zero direction
 * component -- and any ray that fails one is walked again on the reference tree (a handful per million).  The image is the reference's, byte for byte, provided
 * Moller-Trumbore's computed t of every accepted triangle is accurate to 2^-10 relative (the margin by which the second tree's distance culling is relaxed;
 * a wider margin costs 2 % per factor 16): what could slip through is a triangle hit at a grazing angle below ~3e-4 rad whose computed t lands IN FRONT of a nearer
 * triangle the second walk has already found -- the reference would then show the farther triangle, the second tree the nearer one; estimated at well under one ray
 * per 1080p x 1000-sample frame, observed in none (the headline frame in both math modes: 7.4e9 rays).  tests/test_gpu_certified_tree.py compares whole frames,
 * the headline frame included, with the reference kernel's own images, and DsrtRenderDesc.collect_counters = 3 AUDITS a launch: every answer of the second tree
 * is also walked on the reference tree and compared (DsrtStats.certificate_audit_mismatches), for a host that wants the check on its own mesh.  Scenes with
 * spheres reaching beyond 30 extents of the mesh, and cameras farther than that, use the reference tree only.  Off by default; the
 * environment variable DSRT_CERTIFIED_TREE=1 switches it on for contexts created afterwards (the drop-in gpu_render_scene included).  Costs one more tree in HBM
 * (about as much again as the scene) and the SAH build at upload (0.2 s per million triangles).  DSRT_TUNE_REFERENCE_WALK renders without it. */
int  dsrt_ctx_set_certified_tree(DsrtContext* ctx, int on);
int  dsrt_ctx_has_certified_tree(const DsrtContext* ctx);
/* Test hook, no GPU involved: what the upload would prepare for the certified second tree of this host scene (which must carry the reference's tree).
 * counts = {triangles, triangles the reference tree can never reach, triangles in the second tree, its nodes, its height, 1 if no sphere is too far away};
 * *pad = the widening of its boxes; the arrays (each may be NULL) receive the per-triangle flags and reference-leaf boxes and the tree itself. */
int  dsrt_host_scene_second_tree_probe(const DsrtHostScene* hs, int counts[6], float* pad, uint8_t* unreachable, float* leaf_box, GPUBVHNode* nodes, int max_nodes,
                                       int* order, int max_order);
int  dsrt_dropin_has_certified_tree(void);     /* the same question about the scene the drop-in gpu_render_scene converted last (it looks at DSRT_CERTIFIED_TREE on every call) */

/* Upload + re-layout for the GPU (once per scene, not per frame).  `scene` holds HOST pointers in the
 * reference layouts (as from dsrt_host_scene_view); its camera/params/sun are recorded as the current frame.
 * Replaces the upload half of build_gpu_scene (src/gpu_scene_builder.cpp:322-331, 475-546). */
int dsrt_scene_upload(DsrtContext* ctx, const GPUScene* scene);
/* Same, for a GPUScene whose array members are DEVICE pointers (what the reference's own builder
 * produces and what gpu_render_scene receives, src/gpu_render.cu:1037-1066). */
int dsrt_scene_upload_device(DsrtContext* ctx, const GPUScene* scene);
/* Per-frame update: only camera and sun change between frames (src/main.cpp:399-405). */
int dsrt_scene_set_camera_sun(DsrtContext* ctx, const GPUCamera* cam, const float sun_dir_model[3]);

/* PARITY DOMAIN.  rng_mode 0 renders the reference's bytes (as the chosen math_mode defines them) for scenes whose coordinates keep the reference's own
 * intermediates out of the subnormal and overflow ranges: for every box centre c, ray origin o and direction d the walk meets, |c - o| * |d| is zero or lies
 * in [2^-100, 2^100] (metres-scale scenes are 20 orders of magnitude inside; tested from 1e-15 to 1e9 times the station's size, tests/test_gpu_parity.py
 * ::test_scaled_scenes_match_the_oracle_bit_for_bit).  The kernel orders a node's children by 2 d where the reference compares d (render_kernel.hip): the
 * same comparison exactly when no intermediate is subnormal or overflows.  Outside that range images are still valid renders, but near/far ties may
 * resolve differently from the reference's. */
typedef struct DsrtRenderDesc {
    int      width, height;         /* gpu_render_scene(scene, width, height)                     */
    int      spp;                   /* <1 -> 1, as src/gpu_render.cu:987-988                      */
    int      max_depth;             /* <=0 -> 12, as :723-725                                     */
    float    gamma;                 /* <=0 -> 1, as :1043                                         */
    uint64_t seed;
    int      rng_mode;              /* 0 = reference LCG stream per pixel (parity mode, bit-exact);
                                       1 = rocRAND Philox4x32-10, one sub-sequence per (pixel, sample): samples become
                                           independent work items (statistically equivalent image, not bit-identical to mode 0);
                                           a pixel's samples are summed as integers in units of 2^-20, so the image is a function
                                           of (scene, camera, seed) alone -- not of sharding, scheduling or which lane drew what */
    int      tile_size;             /* screen-tile edge in pixels, multiple of 8; 0 -> 8          */
    int      shard_rank;            /* this process renders tiles t with t % shard_count == shard_rank */
    int      shard_count;           /* 0 or 1 -> whole image                                      */
    int      collect_counters;      /* 1 -> counting build of the kernel (fills DsrtStats); 2 -> counting build WITHOUT the any-hit
                                       shadow-ray early-out, whose counters equal the reference traversal's exactly; 3 -> counting build
                                       with the CERTIFICATE AUDIT: when the certified second tree is in use, every one of its answers is also walked
                                       on the reference tree and compared (DsrtStats.certificate_audited / certificate_audit_mismatches) */
    int      checked;               /* 1 -> bounds-checked build of the kernel (tests / first runs) */
    int      stack_entries;         /* LDS short-stack entries per lane: 0 or 8 (the only size built)  */
    int      tune[4];               /* scheduling knobs, 0 = default: {min_walk_iters, advance_budget, leaf_ratio4, flags}.  None of them
                                       changes a pixel (tested against the oracle in every combination).  Flags, DSRT_TUNE_* below: the low
                                       two bits choose the pre-pass, the others switch one scheduling measure off each.  Any other bit is
                                       refused (DSRT_ERR_INVALID): development switches are not part of a render's description -- see
                                       dsrt_dev_set_experiment below */
    int      math_mode;             /* where sinf / cosf / powf come from -- the three library functions of the path (src/gpu_render.cu:104-106, 157-158, 211,
                                       1019-1021).  0 (default): include/dsrt_detmath.h, built from correctly rounded operations only and shared with the CPU
                                       oracle: the image is a function of the inputs alone, the same on the GPU and on a CPU, and what every parity test
                                       against the oracle uses.  1: the device math library's own (what the reference's source gets when hipcc compiles it
                                       for this GPU WITH FLOATING-POINT CONTRACTION OFF): the image is then, byte for byte, the one the reference's own kernel
                                       renders on the same GPU when built that way (oracle/_ref/ref_gpu: hipify-perl + hipcc -ffp-contract=off;
                                       tests/golden/ref_gpu_images.json, tests/test_gpu_reference_fixtures.py).  hipcc's and nvcc's DEFAULT builds contract
                                       a * b + c into one rounding; against such a build (or libdevice's math) either mode is a statistical match only
                                       (oracle/_ref/ref_gpu_fma).  Mode 0 has the same standing against the reference's kernel built with dsrt_detmath.h in
                                       place of those three functions (oracle/_ref/ref_gpu_detmath, tests/golden/ref_gpu_detmath_images.json).  The two modes differ in the last place of those three
                                       functions and therefore, one LCG stream per pixel being what it is, in individual pixels; statistically they are
                                       the same picture.  rng_mode and math_mode are independent */
} DsrtRenderDesc;

#define DSRT_TUNE_NATURAL_ORDER   1    /* tiles in natural order, no empty-tile culling (no pre-pass at all)                   */
#define DSRT_TUNE_NO_CULLING      2    /* costliest-first order, but no tile is dropped as provably empty                     */
#define DSRT_TUNE_NO_HELPERS      4    /* idle lanes do not trace shadow rays for busy lanes of their wave                    */
#define DSRT_TUNE_NO_PROBE        8    /* no probe launch to refine the tile order (rng_mode 0)                               */
#define DSRT_TUNE_NO_STEALING    16    /* rng_mode 1: idle lanes do not take over samples of busy lanes                       */
#define DSRT_TUNE_NO_PRIORITY    32    /* rng_mode 0: waves holding a heavy tile's pixel do not raise their issue priority    */
#define DSRT_TUNE_REFERENCE_WALK 64    /* every ray walks the reference tree even when the certified second tree is resident   */
#define DSRT_TUNE_FLAG_MASK      127

typedef struct DsrtStats {
    float    kernel_ms;             /* HIP events around the render kernel on the given stream (0 if timing off) */
    int      waves_launched;
    uint32_t device_flags;          /* checked-mode status word (0 = clean)                       */
    int      lds_stack_entries;
    uint64_t samples, rays, primary_hits, box_fetches, nodes_entered, internal_entered, tri_tests, hit_updates,
             sphere_tests, shaded_hits, tex_fetches, stack_spills, max_stack;
    /* lane-slot accounting of the counting build: every lane of a wave adds 1 per wave iteration of the node loop / the
     * triangle loop / the advance loop, so active / slots is the SIMD utilisation of that loop */
    uint64_t node_slots, tri_slots, adv_slots, adv_active;
    /* where the idle lanes of the node loop were: parked at a leaf / waiting for their state machine / out of work */
    uint64_t idle_at_leaf, idle_waiting, idle_done;
    /* node visits at BVH depth < 6 / 9 / 12 (root = 0) */
    uint64_t visits_depth_lt6, visits_depth_lt9, visits_depth_lt12;
    /* tiles of this shard / tiles the pre-pass proved empty and left out (their pixels are exactly black) */
    uint64_t tiles_total, tiles_culled;
    /* counting build: sum over the waves of the time each spent in the kernel, in ticks of the 100 MHz wall clock; divided by
     * (waves_launched x kernel time) it is the fraction of the launch the average wave was resident */
    uint64_t wave_ticks;
    /* counting build, certified second tree in use: BVH queries whose certificate failed and which were walked again on the reference tree */
    uint64_t certificate_fallbacks;
    /* collect_counters = 3 (the CERTIFICATE AUDIT): answers of the second tree that were also walked on the reference tree, and how many of them differed (must be 0) */
    uint64_t certificate_audited, certificate_audit_mismatches;
    /* counting build, ms after the first wave started: the heavy / the light work queue handed out its last item, the last wave left */
    float    heavy_queue_empty_ms, light_queue_empty_ms, last_wave_exit_ms;
    int      certified_tree_used;   /* 1: the rays of this launch started on the certified second tree (dsrt_ctx_set_certified_tree) */
} DsrtStats;

/* Number of bytes of the compact per-shard output of dsrt_render for this desc (rgb8) and the number of
 * tiles this shard owns / the padded per-shard tile count (equal on every rank, for a gather). */
int dsrt_shard_layout(const DsrtRenderDesc* desc, int* tiles_total, int* tiles_this_shard, int* tiles_per_shard_padded,
                      size_t* rgb8_bytes_padded);

/*
 * Render.  Asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream) unless
 * `stats` is non-NULL, in which case the call synchronises the stream and fills `stats`.
 *   d_rgb8 : DEVICE buffer.  shard_count <= 1: width*height*3 bytes in image order (top row first), what
 *            the reference copies back and writes after the P6 header (src/gpu_render.cu:1087-1106).
 *            shard_count  > 1: tiles_per_shard_padded * tile*tile*3 bytes, tile-major, local tile k =
 *            global tile k*shard_count + shard_rank, rows of a tile top first.
 *   d_f32  : optional DEVICE buffer, same indexing, 3 floats per pixel: the value multiplied by 255.99
 *            (:1028), for the L-infinity parity check.  May be NULL.
 */
/* Bounding box of the resident scene's BVH root (all zeros without a BVH). */
int dsrt_ctx_scene_bounds(const DsrtContext* ctx, float lo[3], float hi[3]);

int dsrt_render(DsrtContext* ctx, const DsrtRenderDesc* desc, uint8_t* d_rgb8, float* d_f32, void* stream, DsrtStats* stats);

/*
 * Render `frames` views of the resident scene as ONE launch: the reference's frame loop (src/main.cpp:310-431) with the scene kept and
 * only camera and sun changing, but with all frames' pixels in a single pool of work.  In rng_mode 0 a pixel is a serial chain of spp
 * samples and a frame rendered alone ends in a tail of a few such chains on an otherwise empty chip; here a lane that finishes a pixel
 * of frame f goes straight on to frame f + 1, so the chains of one frame run under the bulk of the next ones.  Every frame's image is the
 * image dsrt_render gives for that camera and sun, byte for byte (rng_mode 0: the reference's; rng_mode 1: seed-determined).
 *   cameras      : `frames` cameras (dsrt_camera_look_at / dsrt_pose_frame_camera), HOST array
 *   sun_dirs_xyz : 3 * frames floats, HOST array (radiance and the enabled flag are the context's, dsrt_scene_set_camera_sun)
 *   d_rgb8       : DEVICE buffer, frames * width*height*3 bytes: the images one after another, each in dsrt_render's layout
 *   d_f32        : optional DEVICE buffer, frames * width*height*3 floats
 * With shard_count > 1 every frame's part is this rank's compact tile buffer (dsrt_shard_layout: rgb8_bytes_padded each), i.e. a rank renders its
 * tiles of ALL the frames as one pool -- the split that scales a sequence over the GPUs of a node in either rng_mode, one gather for the lot.
 * Work is handed out frame after frame in the order given (put the costliest -- nearest -- frames first), every frame's first eighth of heavy tiles
 * before any frame's remainder; in rng_mode 0 at 256 samples and more each frame's tiles are first re-sorted by a probe launch, as in dsrt_render.
 * Production kernel only (no counters, not `checked`); frames * pixels per frame (x 16 in rng_mode 1) must stay below 2^32 -- split a longer
 * sequence into several calls.  Asynchronous on `stream` unless `stats` is given
 * (kernel_ms then covers the one launch; the per-frame pre-passes before it are not included).
 */
int dsrt_render_batch(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const GPUCamera* cameras, const float* sun_dirs_xyz,
                      uint8_t* d_rgb8, float* d_f32, void* stream, DsrtStats* stats);
/* The same with the images delivered to HOST memory (frames * width*height*3 bytes), synchronously: for hosts without device code of their own,
 * like the reference's main.cpp. */
int dsrt_render_batch_to_host(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const GPUCamera* cameras, const float* sun_dirs_xyz,
                              uint8_t* h_rgb8, DsrtStats* stats);

/* Root rank, after a gather: tile-major shards [shard][tile][tile*tile*3] -> image-order rgb8. */
int dsrt_deinterleave_tiles(DsrtContext* ctx, const DsrtRenderDesc* desc, const uint8_t* d_gathered, uint8_t* d_rgb8_image,
                            void* stream);
/* The same for the gather of SHARDED BATCH launches (dsrt_render_batch with shard_count > 1): d_gathered = the ranks' buffers one after another,
 * each holding its part of frame 0, of frame 1, ... (`frames` parts of rgb8_bytes_padded bytes); d_rgb8_images receives the `frames` whole images. */
int dsrt_deinterleave_batch(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const uint8_t* d_gathered, uint8_t* d_rgb8_images, void* stream);

/* Convenience for hosts without their own device buffers: render the whole image into HOST memory. */
int dsrt_render_to_host(DsrtContext* ctx, const DsrtRenderDesc* desc, uint8_t* h_rgb8, float* h_f32, DsrtStats* stats);

/* Device evaluation of the shared deterministic math (tests): out[i] = f(x[i]) on the GPU;
 * fn 0 = sin, 1 = cos, 2 = pow(x[i], y).  Host pointers. */
int dsrt_selftest_math(DsrtContext* ctx, int fn, const float* x, float y, float* out, int n);

/* Device evaluation of the render kernel's own material / frame helpers (csrc/device_math.h: reflect, refract, scatter_metal,
 * scatter_dielectric, build_onb, schlick -- src/gpu_render.cu:112-118, 195-212, 603-661) on explicit inputs, for comparison with known
 * answers produced by the reference's host code (tests/golden/ref_matkat.json).  12 words in and 12 words out per case, host pointers;
 * the packing per `fn` (0..5) is documented at dsrt_devkat_kernel in csrc/render_kernel.hip. */
int dsrt_selftest_devkat(DsrtContext* ctx, int fn, const float* in12, float* out12, int n);

/* Device check that the kernel's stateless Philox4x32-10 equals rocRAND's engine: the first n 32-bit words of
 * (seed, subsequence, offset 0) from both.  Host pointers. */
int dsrt_selftest_philox(DsrtContext* ctx, uint64_t seed, uint64_t subsequence, int n, uint32_t* ours, uint32_t* rocrand_words);

/* ===================================================================================== */
/* All GPUs of a node from ONE host process (the reference's main() is one process calling */
/* gpu_render_scene per frame, src/main.cpp:310-431; it has no multi-GPU code).           */
/* ===================================================================================== */
typedef struct DsrtMulti DsrtMulti;

/* `devices[r]` is the HIP device of rank r.  Distinct devices: an RCCL communicator is created over them (ncclCommInitAll).
 * Ranks that share a device (a one-GPU box: tests) get the same code path with the gather done by device-to-device copies --
 * RCCL refuses duplicate devices.  `frames_in_flight` (>= 1) is used by dsrt_multi_render_sequence only. */
int  dsrt_multi_create(const int* devices, int n, int frames_in_flight, DsrtMulti** out);
void dsrt_multi_destroy(DsrtMulti* m);
int  dsrt_multi_count(const DsrtMulti* m);
int  dsrt_multi_uses_rccl(const DsrtMulti* m);
/* Self-test: a one-rank RCCL communicator on `device` and one ncclGather of `bytes` bytes through it, checked.  All of the collective
 * path that can run on a single GPU. */
int  dsrt_selftest_rccl_gather(int device, size_t bytes);

/* Development switches (scheduling experiments of the A/B tools under tools/; the bits are listed in csrc/device_api.hip).  One process-wide word; its initial
 * value is the environment variable DSRT_EXPERIMENT, read ONCE in the first dsrt_device_count / dsrt_ctx_create call -- never per render.  Undefined bits are
 * refused (DSRT_ERR_INVALID) and a non-zero word is announced on stderr.  None changes an image byte; bit 27 replaces the optional FLOAT image of a counting
 * build by timing words.  Not for production hosts. */
int  dsrt_dev_set_experiment(uint32_t word);
/* Test hook for the bounds-checked kernel build: overwrites 32-bit word `word_index` of the context's resident node-record array (device_layout.h: 16 words per
 * record, words 12 and 13 are the child references) and returns the previous value in *old_value.  A checked render of a scene corrupted this way must come
 * back with DSRT_ERR_DEVICE_FLAG, not hang. */
int  dsrt_selftest_poke_node_word(DsrtContext* ctx, size_t word_index, uint32_t value, uint32_t* old_value);
/* The scene (HOST pointers, reference layouts) is converted and made resident once per device. */
int  dsrt_multi_scene_upload(DsrtMulti* m, const GPUScene* host_scene);
/* ONE frame over all ranks: interleaved screen tiles (tile g -> rank g mod N), one ncclGather of the equal-sized compact
 * buffers to rank 0, de-interleave there, image (width*height*3, top row first) copied to `h_rgb8` (host).  desc's shard fields
 * are ignored.  Optional outputs: per-rank render time in ms (HIP events; N floats) and the wall time of the whole call. */
int  dsrt_multi_render_frame(DsrtMulti* m, const DsrtRenderDesc* desc, const GPUCamera* cam, const float sun_dir_model[3],
                             uint8_t* h_rgb8, float* kernel_ms_per_rank, double* seconds);
/* MANY frames of the resident scene (src/main.cpp's frame loop): every rank renders ITS interleaved tiles of EVERY frame as sharded batch
 * launches (dsrt_render_batch with a shard), nearest frames first, one gather per launch for all its frames and the de-interleave on rank 0.
 * cams[i] / sun_dirs[3 i ..] as from dsrt_camera_look_at / dsrt_pose_to_frame.  h_images may be NULL (timing only) or hold n_frames host
 * pointers (NULL entries are skipped). */
int  dsrt_multi_render_sequence(DsrtMulti* m, const DsrtRenderDesc* desc, const GPUCamera* cams, const float* sun_dirs, int n_frames,
                                uint8_t* const* h_images, double* seconds);

/* Calibration of the roofline the render kernel is measured against (bench.py): a kernel with the render
 * kernel's launch shape (256-thread workgroups, 4 waves per SIMD, grid = resident set) in which `live_lanes` of every 64 lanes gather
 * random aligned 64-byte records from a table of `table_bytes` bytes, `iters` records per lane, and do nothing else.
 *   mode 0: each lane reads its record as 4 x 16-byte loads (the render kernel's access shape)
 *   mode 1: the four lanes of a quad read one record per load instruction into an LDS tile (global_load_lds_dwordx4), 4 x ds_read_b128 back
 *   mode 2: as 1 through registers (global_load_dwordx4 + ds_write_b128)
 * dependent != 0: the next record index depends on the loaded data (a traversal step); pad_valu: dependent v_fma per record.
 * Returns the kernel time (HIP events) and the number of records gathered.  No reference interface: measurement only. */
int dsrt_microbench_gather(int device, int mode, int dependent, int live_lanes, int pad_valu, size_t table_bytes, int iters,
                           float* out_ms, double* out_records);

/* The other calibration: what a vector-ALU instruction costs to issue.  `waves_per_simd` (1..8) workgroups per CU (one wave per SIMD each) each
 * issue iters x 32 instructions from eight independent register streams: pattern 0 = the instruction of `kind` 32 times, 1 = alternating with
 * v_add_f32, 2 = in pairs between pairs of v_add_f32 (some kinds cost far more back to back), 3 = 16 of them then 16 v_add_f32, 4 / 5 = as 1 / 3
 * with v_pk_mul_f32 as the partner; only the lanes in `lane_mask` execute them.
 * Returns the kernel time (HIP events), the wave-instructions issued and the shader clock during the run (the waves' s_memtime against the
 * 100 MHz s_memrealtime): cycles per instruction per SIMD = SIMDs x clock x time / instructions.  dsrt_microbench_valu_kinds /
 * _kind_name enumerate the kinds.  No reference interface: measurement only. */
int dsrt_microbench_valu(int device, int kind, int pattern, int waves_per_simd, int iters, uint64_t lane_mask, float* out_ms, double* out_wave_instructions,
                         double* out_shader_clock_GHz);
int dsrt_microbench_valu_kinds(void);
const char* dsrt_microbench_valu_kind_name(int kind);

/* The third calibration: what this board's HBM delivers to a plain streaming copy -- a float4 grid-stride kernel (16 bytes per lane per access), `blocks_per_cu`
 * 256-thread workgroups per CU, `bytes` per buffer (take it far beyond the 256 MB Infinity Cache), `reps` launches back to back; `mode` 0 = copy, 1 = read only,
 * 2 = write only, 3 = copy with non-temporal loads and stores.  Returns the time (HIP events)
 * and the bytes moved (read + written).  The figure bench.py prints as extras.hbm_copy_GBps_measured, next to the 8 TB/s of the specification.
 * No reference interface: measurement only. */
int dsrt_microbench_copy(int device, int mode, size_t bytes, int blocks_per_cu, int reps, float* out_ms, double* out_bytes_moved);

/* ===================================================================================== */
/* Drop-in layer: the reference's own three entry points.                                */
/* ===================================================================================== */
/* src/gpu_render.cu:1037-1038, declared at its call site src/main.cpp:24-25.  `scene` holds DEVICE pointers.
 * Blocking; writes image_gpu.ppm into the CWD; failures print to stderr and return (no file).  math_mode 0 unless the environment variable
 * DSRT_MATH_MODE is 1 (the signature has no room for it): with it the file is, byte for byte, the one the reference's own gpu_render_scene
 * writes when its source is built for this GPU. */
void gpu_render_scene(const GPUScene* scene, int width, int height);

/* C forms of build_gpu_scene / free_gpu_scene (inc/gpu_scene_builder.h:72-73): the C++ overloads taking
 * hittable_list/camera/vec3 live in deep-space-ray-tracer_amd/host/scene_model.hpp and forward here. */
int  dsrt_build_gpu_scene(const DsrtHostScene* hs, const GPUCamera* cam, const float sun_dir_model[3], GPUScene* out);
void dsrt_free_gpu_scene(GPUScene* scene);

#ifdef __cplusplus
}
#endif
#endif /* DSRT_H */
