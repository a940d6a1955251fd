/*
 * dsrt_oracle.h -- CPU restatement of the reference's per-pixel x spp sampling loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (include/, deep-space-ray-tracer_amd/) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py -- always as the checker / the reported CPU baseline, never as
 * the thing measured or shipped.
 *
 * What it restates: src/gpu_render.cu of the reference (render_kernel :973-1031 and everything
 * it calls; each function in dsrt_oracle.c cites its lines).  Arithmetic is fp32, same operation
 * order, compiled with -ffp-contract=off.
 *
 * Pinning status (read before trusting it):
 *   - The reference has no tests, golden images or known-answer vectors, and its
 *     render kernel is a CUDA translation unit that cannot be built in this image (no nvcc, no
 *     libcudart; we do not write stand-ins).  What CAN be executed is pinned:
 *       * everything that FEEDS the loop (pose -> camera, OBJ/MTL -> triangles/materials, flattening,
 *         median-split BVH) against the reference's real host code compiled from /root/reference by
 *         oracle/Makefile into oracle/_ref/ (goldens in tests/golden/ref_*);
 *       * bbox_hit / hit_sphere / Moller-Trumbore against the reference's CPU classes aabb::hit (exact),
 *         sphere::hit, triangle::hit (to rounding), tests/golden/ref_hitkat.json;
 *       * rand01, the rejection loop of random_in_unit_sphere, the local cosine direction and the
 *         camera-ray set-up against the reference's own host-compilable DEVICE helpers, executed
 *         (inc/rtweekend.h:126-202, inc/camera.h:35-61; tests/golden/ref_devkat.json): bit for bit where
 *         the arithmetic is float or exact, to 4e-7 where the helper computes in double;
 *       * reflect, refract, scatter_metal (fuzz 0: direction and accept test), scatter_dielectric on the
 *         total-internal-reflection branch (direction, and that no random number is drawn), build_onb
 *         and schlick against the reference's host helpers and material classes, executed (inc/vec3.h:136-147,
 *         inc/material.h:28-32, 123-180, inc/onb.h:47-56; tests/golden/ref_matkat.json): bit for bit where
 *         the host arithmetic is float (everything but reflectance, which is double: to 3e-7); build_onb's u
 *         is the exact negative of the class's (cross(v, w) against cross(w, v)).
 *   - CONTROL FLOW AND ORDER (ray_color :715-936, scene_hit :509-551, the traversal order of bvh_hit_closest :387-473), i.e. THE LOOP AS A WHOLE, are pinned
 *     by images the reference's own kernel rendered, committed as data:
 *       * DIRECTLY: oracle/_ref/ref_gpu_detmath is the reference's renderer (src/gpu_render.cu and src/gpu_scene_builder.cpp translated CUDA -> HIP by the
 *         image's hipify-perl, compiled with hipcc -ffp-contract=off: oracle/Makefile) with its three libm names (cosf, sinf, powf) mapped onto
 *         include/dsrt_detmath.h by ref_gpu_detmath_prelude.h -- the functions THIS file uses.  Its images of the six parity scenes, 30 randomised views and the
 *         station on pose frames are in tests/golden/ref_gpu_detmath_images.json (made on an MI355X by tests/golden/make_ref_gpu_fixtures.py), and this
 *         restatement reproduces them byte for byte in the CPU suite (tests/test_oracle_reference_fixtures.py): one hop, no product code in between.
 *       * and through the product: oracle/_ref/ref_gpu (the same build with the device math library) == the HIP kernel in math_mode 1, byte for byte
 *         (tests/golden/ref_gpu_images.json, tests/test_gpu_reference_fixtures.py, live: tests/test_gpu_reference_kernel.py), whose math_mode 0 build equals
 *         THIS restatement bit for bit (tests/test_gpu_parity.py).
 *     What no fixture can cover: a real CUDA run (nvcc contracts to FMA by default, libdevice math; no NVIDIA GPU exists in this pipeline).
 *   - cosf/sinf/powf come from include/dsrt_detmath.h (shared with the HIP kernel), not from any
 *     libm: see that header.  Build with -DDSRT_ORACLE_LIBM to use the host libm instead (for the
 *     statistical comparison only).
 */
#ifndef DSRT_ORACLE_H
#define DSRT_ORACLE_H

#include "../include/dsrt_scene_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters, all per call (added to, not reset).  Definitions follow SURVEY.md section 8(d). */
typedef struct DsrtOracleCounters {
    uint64_t samples;          /* ray_color invocations                                         */
    uint64_t rays;             /* scene_hit invocations (primary + bounce + shadow)             */
    uint64_t primary_hits;     /* samples whose depth-0 ray hit something                       */
    uint64_t box_tests;        /* every bbox_hit call the reference makes (incl. its re-tests)  */
    uint64_t box_fetches;      /* n_box: distinct node boxes read = 1 per BVH ray + 2 per internal node entered */
    uint64_t nodes_entered;    /* n_enter: nodes whose own bbox test passed                     */
    uint64_t internal_entered; /* of those, internal nodes                                      */
    uint64_t tri_tests;        /* n_tri: Moller-Trumbore evaluations                            */
    uint64_t hit_updates;      /* n_upd: accepted closest-hit updates                           */
    uint64_t sphere_tests;
    uint64_t shaded_hits;      /* material fetches                                              */
    uint64_t tex_fetches;
    uint64_t max_stack;        /* max BVH stack depth seen                                      */
    uint64_t rng_draws;
} DsrtOracleCounters;

/*
 * Render rows [y0, y1) (kernel coordinates: y = 0 is the BOTTOM row of the image, exactly as
 * blockIdx.y*8+threadIdx.y in src/gpu_render.cu:984; the store flips to row H-1-y, :1027).
 * `scene` is a GPUScene whose array members are HOST pointers.
 *   rgb8    : W*H*3 bytes, image order (top row first), or NULL
 *   rgb_f32 : W*H*3 floats, same indexing, the value that is multiplied by 255.99 (:1028), or NULL
 * Returns 0, or a negative number for invalid arguments.
 */
int dsrt_oracle_render_rows(const GPUScene* scene, int W, int H, int y0, int y1,
                            uint8_t* rgb8, float* rgb_f32, DsrtOracleCounters* counters);

/* rand01 of src/gpu_render.cu:77-80, exposed for the known-answer test. */
float dsrt_oracle_rand01(uint32_t* state);

/* One scene_hit (src/gpu_render.cu:509-551) for ray-level tests.  out = {t, px,py,pz, nx,ny,nz, u, v};
 * ids = {mat_id, tri_tex_id, tri_index, front_face}.  Returns 1 on hit. */
int dsrt_oracle_scene_hit(const GPUScene* scene, const float orig[3], const float dir[3],
                          float t_min, float t_max, float out[9], int ids[4]);

/* bbox_hit of src/gpu_render.cu:285-315 on one box (same algorithm as the reference's CPU aabb::hit, inc/aabb.h:33-56). */
int dsrt_oracle_bbox_hit(const float lo[3], const float hi[3], const float orig[3], const float dir[3], float t_min, float t_max);

/* The shared deterministic math, exposed so tests can compare device results bit for bit. */
/* The oracle's rejection loop (:82-91), cosine direction in local coordinates (:99-109) and camera ray (:941-968) on their own,
 * for the known-answer vectors produced by the reference's host-compilable device helpers (tests/golden/ref_devkat.json). */
void dsrt_oracle_random_in_unit_sphere(uint32_t* state, float out[3]);
void dsrt_oracle_random_cosine_direction(uint32_t* state, float out[3]);
void dsrt_oracle_camera_ray(const GPUCamera* cam, int px, int py, int W, int H, float jx, float jy, float orig[3], float dir[3]);

/* reflect :195, refract :199-206 (normalises its argument first), f3_norm :51-56, scatter_metal :603-619 (returns "goes on"),
 * scatter_dielectric :621-661, build_onb :112-118, schlick :208-212 on their own, for the known answers produced by the reference's host
 * helpers and material classes (inc/vec3.h:136-147, inc/material.h:28-32, 123-180, inc/onb.h:47-56; tests/golden/ref_matkat.json). */
void dsrt_oracle_reflect(const float v[3], const float n[3], float out[3]);
void dsrt_oracle_refract(const float v[3], const float n[3], float eta, float out[3]);
void dsrt_oracle_normalize(const float v[3], float out[3]);
int  dsrt_oracle_scatter_metal(const float dir[3], const float n[3], float fuzz, uint32_t* state, float out_dir[3]);
void dsrt_oracle_scatter_dielectric(const float dir[3], const float n[3], int front_face, float ref_idx, uint32_t* state, float out_dir[3]);
void dsrt_oracle_build_onb(const float n[3], float u[3], float v[3], float w[3]);
float dsrt_oracle_schlick(float cosine, float ref_idx);

float dsrt_oracle_sinf(float x);
float dsrt_oracle_cosf(float x);
float dsrt_oracle_powf(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
