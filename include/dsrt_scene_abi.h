/*
 * dsrt_scene_abi.h -- the POD data contract between the host scene builder and the renderer.
 *
 * These structs are field-for-field, offset-for-offset the ones the reference declares in
 *   inc/gpu_scene.h:28-173 (GPUTextureHeader, GPUMaterial, GPUSphere, GPUTriangle, GPUBVHNode,
 *   GPURenderParams, GPUScene) and inc/camera.h:13-30 (GPUCamera),
 * so that a GPUScene assembled by the reference's own builder can be handed to this library's
 * gpu_render_scene() unchanged and vice versa.  The reference spells its 3-vectors `float3`
 * (from <cuda_runtime.h>) and `vec3` (class with float e[3], inc/vec3.h:14-22); both are
 * 12 bytes / 4-byte aligned, which is what DsrtF3 is.  Offsets are pinned by the static
 * assertions at the bottom (values: SURVEY.md section 8(b), re-measured in
 * tests/test_host_golden.py::test_abi_matches_reference_layout against the reference headers compiled in oracle/_ref).
 *
 * Plain C: usable from C, C++, HIP and (through ctypes) Python.
 */
#ifndef DSRT_SCENE_ABI_H
#define DSRT_SCENE_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DsrtF3 { float x, y, z; } DsrtF3;

/* inc/gpu_scene.h:21-26 */
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3 };
/* inc/gpu_scene.h:90-94 */
enum { SKY_SOLID = 0, SKY_GRADIENT = 1, SKY_ENV_MAP = 2 };

/* inc/gpu_scene.h:28-32 */
typedef struct GPUTextureHeader {
    int width;
    int height;
    int offset;            /* float index into texture_pool (RGB packed) */
} GPUTextureHeader;

/* inc/gpu_scene.h:34-45 */
typedef struct GPUMaterial {
    int    type;
    int    albedo_tex;
    int    _pad0;
    int    _pad1;
    DsrtF3 albedo;
    DsrtF3 emissive;
    float  fuzz;
    float  ref_idx;
} GPUMaterial;

/* inc/gpu_scene.h:50-55 */
typedef struct GPUSphere {
    DsrtF3 center;
    float  radius;
    int    material_id;
    int    _pad;
} GPUSphere;

/* inc/gpu_scene.h:57-72 */
typedef struct GPUTriangle {
    DsrtF3 v0, v1, v2;
    DsrtF3 n0, n1, n2;
    DsrtF3 uv0, uv1, uv2;  /* (u, v, 0) */
    int    material_id;
    int    albedo_tex;     /* -1 if none */
} GPUTriangle;

/* inc/gpu_scene.h:77-85 */
typedef struct GPUBVHNode {
    DsrtF3 bbox_min;
    DsrtF3 bbox_max;
    int    left;           /* -1 in a leaf */
    int    right;
    int    tri_offset;     /* into tri_indices */
    int    tri_count;      /* >0 => leaf */
} GPUBVHNode;

/* inc/gpu_scene.h:96-111 */
typedef struct GPURenderParams {
    int   img_width;
    int   img_height;
    int   samples_per_pixel;
    int   max_depth;
    int   use_bvh;
    int   rng_mode;        /* reference: set, never read.  Here: 0 = reference LCG stream (parity mode) */
    int   tile_size;       /* reference: set, never read.  Here: screen-tile edge for multi-GPU sharding (0 = default) */
    int   _pad0;
    float gamma;
    float exposure;        /* passed through and ignored, as in src/gpu_render.cu:979,1002 */
    float env_rotation;
    float _pad1;
} GPURenderParams;

/* inc/camera.h:13-30 */
typedef struct GPUCamera {
    DsrtF3 origin;
    DsrtF3 lower_left_corner;
    DsrtF3 horizontal;
    DsrtF3 vertical;
    DsrtF3 u, v, w;
    float  lens_radius;
    int    image_width;
    int    image_height;
    int    samples_per_pixel;
    int    max_depth;
} GPUCamera;

/* inc/gpu_scene.h:116-173.  A HOST struct whose array members are DEVICE pointers
 * (src/gpu_render.cu:1059-1066 copies the 384 bytes to the device before the launch). */
typedef struct GPUScene {
    const GPUSphere*        spheres;
    int                     num_spheres;
    int                     _pad_sph0;
    int                     _pad_sph1;

    const GPUTriangle*      triangles;
    const int*              tri_indices;
    int                     num_triangles;
    int                     _pad_geo;

    GPUBVHNode*             bvh_nodes;
    int                     num_bvh_nodes;

    int*                    bvh_tri_indices;   /* alias of tri_indices (src/gpu_scene_builder.cpp:500-503) */

    const GPUMaterial*      materials;
    int                     num_materials;
    int                     _pad_mat0;
    int                     _pad_mat1;

    const GPUTextureHeader* textures;
    int                     num_textures;
    int                     _pad_tex0;
    int                     _pad_tex1;

    const float*            texture_pool;
    int                     texture_pool_floats;
    int                     _pad_pool0;
    int                     _pad_pool1;

    GPUCamera               camera;

    int                     sky_type;
    int                     env_tex_id;
    int                     _pad_sky0;
    int                     _pad_sky1;
    DsrtF3                  sky_solid;
    DsrtF3                  sky_top;
    DsrtF3                  sky_bottom;

    GPURenderParams         params;

    uint64_t                seed;

    uint8_t                 sun_enabled;       /* C++ `bool` in the reference: 1 byte */
    uint8_t                 _pad_sun[3];
    DsrtF3                  sun_dir;           /* ISS -> Sun; the kernel negates it (src/gpu_render.cu:802-806) */
    DsrtF3                  sun_radiance;
} GPUScene;

#ifdef __cplusplus
}
#define DSRT_SA(cond) static_assert(cond, #cond)
#else
#define DSRT_SA(cond) _Static_assert(cond, #cond)
#endif

DSRT_SA(sizeof(DsrtF3) == 12);
DSRT_SA(sizeof(GPUTextureHeader) == 12);
DSRT_SA(sizeof(GPUMaterial) == 48 && offsetof(GPUMaterial, albedo) == 16 && offsetof(GPUMaterial, emissive) == 28 &&
        offsetof(GPUMaterial, fuzz) == 40 && offsetof(GPUMaterial, ref_idx) == 44);
DSRT_SA(sizeof(GPUSphere) == 24);
DSRT_SA(sizeof(GPUTriangle) == 116 && offsetof(GPUTriangle, n0) == 36 && offsetof(GPUTriangle, uv0) == 72 &&
        offsetof(GPUTriangle, material_id) == 108 && offsetof(GPUTriangle, albedo_tex) == 112);
DSRT_SA(sizeof(GPUBVHNode) == 40 && offsetof(GPUBVHNode, left) == 24 && offsetof(GPUBVHNode, tri_count) == 36);
DSRT_SA(sizeof(GPURenderParams) == 48);
DSRT_SA(sizeof(GPUCamera) == 104 && offsetof(GPUCamera, lens_radius) == 84 && offsetof(GPUCamera, max_depth) == 100);
DSRT_SA(sizeof(GPUScene) == 384);
DSRT_SA(offsetof(GPUScene, num_spheres) == 8 && offsetof(GPUScene, triangles) == 24 && offsetof(GPUScene, tri_indices) == 32);
DSRT_SA(offsetof(GPUScene, num_triangles) == 40 && offsetof(GPUScene, bvh_nodes) == 48 && offsetof(GPUScene, num_bvh_nodes) == 56);
DSRT_SA(offsetof(GPUScene, bvh_tri_indices) == 64 && offsetof(GPUScene, materials) == 72 && offsetof(GPUScene, num_materials) == 80);
DSRT_SA(offsetof(GPUScene, textures) == 96 && offsetof(GPUScene, num_textures) == 104 && offsetof(GPUScene, texture_pool) == 120);
DSRT_SA(offsetof(GPUScene, texture_pool_floats) == 128 && offsetof(GPUScene, camera) == 140 && offsetof(GPUScene, sky_type) == 244);
DSRT_SA(offsetof(GPUScene, env_tex_id) == 248 && offsetof(GPUScene, sky_solid) == 260 && offsetof(GPUScene, sky_top) == 272);
DSRT_SA(offsetof(GPUScene, sky_bottom) == 284 && offsetof(GPUScene, params) == 296 && offsetof(GPUScene, seed) == 344);
DSRT_SA(offsetof(GPUScene, sun_enabled) == 352 && offsetof(GPUScene, sun_dir) == 356 && offsetof(GPUScene, sun_radiance) == 368);

#endif /* DSRT_SCENE_ABI_H */
