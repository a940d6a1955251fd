// device_layout.h -- how the scene lives in HBM, and the argument block of the render kernel.
//
// The reference keeps the scene as it came off the host: 116-byte AoS triangles, 40-byte AoS BVH nodes, an
// index indirection between them (inc/gpu_scene.h:57-85), and tests three 24-byte boxes per internal node from
// three different 40-byte records (src/gpu_render.cu:408-431).  Here the scene is re-laid-out ONCE at upload
// for how a wave64 traversal actually touches it:
//
//   pairs      one 64-byte record per INTERNAL node holding BOTH children's boxes and both child references: a node
//              visit is one aligned 64-byte gather per lane (4 x dwordx4), not three scattered ones.  The two boxes are
//              interleaved so that every (left, right) pair of like coordinates sits in adjacent registers after the load
//              and the slab arithmetic runs on packed fp32 instructions:
//                q0 = (L.lo.x, R.lo.x, L.hi.x, R.hi.x)   q1 = (L.lo.y, R.lo.y, L.hi.y, R.hi.y)
//                q2 = (L.lo.z, R.lo.z, L.hi.z, R.hi.z)   q3 = (left_ref, right_ref as int bits, L.lo.x + L.hi.x, R.lo.x + R.hi.x)
//              (the last two: the x sums of the child-ordering test, formed in float on the host exactly as the kernel would -- one packed add per visit)
//              Records are in depth-first order (a left child sits right behind its parent).
//   child ref  >= kRefBias (64): index + kRefBias of an internal node in `pairs` (below)
//              <  0: a leaf: bit31 | code<<28 | payload.  code 0..6: count = code+1 triangles whose first PAIR record is
//                    `payload`; code 7: payload indexes `big_leaves` {first pair, count} (leaves of more than 7 triangles
//                    only arise from coincident centroids, builder :408-414).
//   tri_pairs  a leaf's triangles in leaf order (tri_indices order; the indirection is gone), TWO per 80-byte record and
//              interleaved like the boxes, so Moller-Trumbore runs on packed fp32 pairs, two triangles per step:
//                (v0x.A, v0x.B, v0y.A, v0y.B) (v0z.A, v0z.B, e1x.A, e1x.B) (e1y.A, e1y.B, e1z.A, e1z.B)
//                (e2x.A, e2x.B, e2y.A, e2y.B) (e2z.A, e2z.B, -, -)        e1 = v1 - v0, e2 = v2 - v0 in float,
//              exactly the differences hit_triangle_index forms per test (src/gpu_render.cu:336-337).  A leaf with an odd
//              count ends in a half-empty record whose B triangle is all zeros (det = 0: rejected by the |det| test).
//              "Slot" of a triangle = 2 * pair + (0 for A, 1 for B); slots of padding triangles are never hit.
//   tri_shade  indexed by slot, 48 bytes, read once per closest hit, not per candidate:
//                (n0.x, n0.y, n0.z, n1.x) (n1.y, n1.z, n2.x, n2.y) (n2.z, material_id, albedo_tex, original index)
//   tri_uv     indexed by slot, 32 bytes, only allocated when the scene has textures: (uv0.xy, uv1.xy) (uv2.xy, 0, 0)
//   materials  the reference's 48-byte record viewed as 3 x float4.
//   spheres / texture headers / texture pool: unchanged (few, wave-uniform or rarely touched).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsrt_scene_abi.h"

namespace dsrt {

constexpr int kLeafBit = (int)0x80000000u;
// Node references as a lane's `cur` holds them.  Every sentinel and every class boundary is an INLINE constant of the ISA (integers -16 .. 64 cost no
// register and no literal): the traversal loop's compares and selects name them directly, where 0x7FFFFFFE cost a move per iteration and a register.
//   cur <  0            a leaf (bit 31 | code << 28 | payload)
//   cur == kRefNone     no node: traversal finished / no BVH
//   cur == kRefPop      take the next postponed child from the stack
//   cur >= kRefBias     internal node number (cur - kRefBias) of `pairs`
constexpr int kRefNone = 62;
constexpr int kRefPop = 63;
constexpr int kRefBias = 64;

__host__ __device__ inline bool ref_is_leaf(int r) { return r < 0; }
__host__ __device__ inline bool ref_is_internal(int r) { return r >= kRefBias; }
__host__ __device__ inline int leaf_code(int r) { return (r >> 28) & 7; }
__host__ __device__ inline int leaf_payload(int r) { return r & 0x0FFFFFFF; }
inline int make_leaf_ref(int code, int payload) { return kLeafBit | (code << 28) | payload; }

struct DeviceScene {
    const float4* pairs;
    const uint8_t* pair_depth;  // depth of every record of `pairs` (root 0): read by the counting build only
    const char*   pairs_biased; // pairs - kRefBias records, as bytes: an internal reference r is the record at pairs_biased + (r << 6)
    const float4* tri_pairs;
    const float4* tri_shade;
    const float4* tri_uv;
    const int2*   big_leaves;
    const float4* materials;
    const GPUSphere* spheres;
    const GPUTextureHeader* tex_headers;
    const float*  tex_pool;
    int   num_pairs, num_tri_pairs, num_big_leaves, num_materials;
    int   num_spheres, num_lights, num_textures, tex_pool_floats;
    float root_lo[3];
    float root_hi[3];
    int   root_ref;            // kRefNone when there is no BVH
    int   stack_need;          // deepest the traversal stack can get for this BVH (the deeper of the two trees when there are two)
    // The certified second tree (device_api.hip: pack_scene; path_machine.h: the certificate).  Both trees live in the SAME arrays -- the second tree's node
    // records, pair records and shading records first, the reference tree's behind them -- so a reference is a reference whichever tree a lane is on.
    const float4* tri_cert;    // per slot OF THE SECOND TREE, 2 x float4: the box of the triangle's leaf ON THE REFERENCE TREE (lo.xyz, hi.x) (hi.yz, -, -); null = no second tree
    int   accel_root_ref;      // the second tree's root (kRefNone: none)
    float accel_root_lo[3];
    float accel_root_hi[3];
};

constexpr int kCamOrigin = 0, kCamLlc = 3, kCamHorizontal = 6, kCamVertical = 9;     // offsets into FrameParams::cam / BatchFrame::cam

struct FrameParams {
    float cam[12];             // origin, lower-left corner, horizontal, vertical (kCamOrigin ... below), as GPUCamera has them
    float sun_dir[3], sun_radiance[3];
    int   sun_enabled;
    int   width, height, spp, max_depth;
    float inv_gamma;
    uint32_t seed32;
    uint32_t seed_hi;          // rng_mode 1: upper half of the 64-bit seed (Philox key.y)
    int   chunks, chunk_len;   // rng_mode 1: a pixel's spp samples are split into `chunks` work items of `chunk_len` samples
    int   light_chunk_len;     // rng_mode 1: samples per work item of a background pixel (path_machine.h, ST_FETCH)
    int   tile;                // tile edge in pixels (multiple of 8)
    int   tiles_x, tiles_y;
    int   shard_rank, shard_count;
    int   local_tiles;         // tiles owned by this shard
    uint32_t total_items;      // local_tiles * tile * tile
    int   compact_output;      // 1: tile-major shard buffer, 0: image order
    const uint32_t* tile_order; // processing order of this shard's tiles (local tile numbers, costliest first); null = natural order
};

// One entry of a batch launch's table (dsrt_render_batch): a contiguous part of ONE frame's tile order, with what differs from frame to
// frame.  Every frame has two entries -- the first eighth of its heavy tiles (its longest chains), and the rest -- and the table lists all
// frames' first parts before all frames' second parts, so that every frame's longest chains start at the beginning of the launch.  Camera,
// sun and image slot are filled by the host (the same in both entries of a frame), the counts by dsrt_batch_table_kernel from what each
// frame's pre-pass left in its `sched` words.
struct BatchFrame {
    float    cam[12];          // as FrameParams::cam
    float    sun_dir[3];
    uint32_t item_end;         // work items of frames 0 .. this one (exclusive end of this frame's range in the batch queue)
    uint32_t order_base;       // where this frame's tile order starts in RenderArgs::batch_order
    uint32_t n_heavy, n_live;  // tiles that see geometry / tiles not proven empty
    uint32_t slices, chunk_len; // rng_mode 1: work items per heavy pixel and their length
    uint32_t image_slot;       // which of the launch's output images this entry's pixels belong to
    uint32_t pad[2];
};
static_assert(sizeof(BatchFrame) == 96, "BatchFrame is shared with the host as an array");

struct RenderArgs {
    DeviceScene scene;
    FrameParams frame;
    uint8_t*  out_rgb8;
    float*    out_f32;
    uint32_t* queue;           // next work item of the HEAVY queue (pixels, or sample slices, of tiles that see geometry)
    uint32_t* queue_light;     // next work item of the LIGHT queue (pixels of the other live tiles); its own cache line
    const uint32_t* sched;     // written by the pre-pass, read-only during the render (path_machine.h, ST_FETCH):
                               //   [0] n_heavy  tiles that see geometry (the first n_heavy entries of tile_order)
                               //   [1] n_live   tiles in tile_order: the shard's tiles minus those proven empty
                               //   [2] spread   lanes per wave (1..64) that serve the heavy queue first; the others start on the light one
    uint32_t* probe_queue;     // probe launch only: 64 queue words, 64 bytes apart (path_machine.h, ST_FETCH)
    int       hot;             // rng_mode 0: 1 = a wave that holds a pixel of a heavy tile raises its issue priority (render_body)
    const BatchFrame* batch;   // batch launch only: the table, batch_frames entries (two per frame)
    const uint32_t* batch_order; //   the frames' tile orders, BatchFrame::order_base apart
    uint32_t  batch_frames, batch_frame_pixels;   // table entries in the launch; output pixels per frame (W*H, or a shard's padded tile buffer)
    uint32_t* tile_work;       // probe launch only (null otherwise): rays traced per local tile, the measured cost the order is refined by
    uint64_t* counters;        // kNumCounters entries (counting build only)
    uint32_t* flags;           // checked-mode status word
    unsigned long long* accum_fixed; // rng_mode 1: [output pixel][3] sample sums in units of 2^-20, added with integer atomics (exact, so the
                               //   image does not depend on how a pixel's samples were split between work items)
    uint2*    spill;           // stack overflow area: [(entry - K) * spill_stride + global lane]
    uint32_t  spill_stride;
    int       spill_entries;
    int       min_walk_iters;  // x10: the traverse phase yields to ADVANCE once (waiting lane-slots wasted) >= this/10 * walking lanes
    int       advance_budget;  // state transitions per lane per advance phase
    int       helpers;         // 1: idle lanes trace shadow rays for busy lanes of their wave (path_machine.h)
    int       steal;           // rng_mode 1: 1 = a lane that is out of work takes over half the remaining samples of a busy lane of its wave
    int       leaf_ratio4;     // x10: the node loop yields to the leaf pass once (parked lane-slots wasted) >= this/10 * descending lanes
    int       deal_leaves;     // 1: a parked leaf's second pair record is evaluated by a lane that is not at a leaf (render_kernel.hip, phase L)
    int       accel;           // 1: rays start on the certified second tree and fall back to the reference tree when the certificate fails (path_machine.h)
    int       audit;           // counting build: 1 = every answer of the second tree is also walked on the reference tree and compared (path_machine.h, CERTIFICATE AUDIT)
};

// order matches the DsrtStats tail in include/dsrt.h
enum Counter { C_SAMPLES, C_RAYS, C_PRIMARY_HITS, C_BOX_FETCHES, C_NODES_ENTERED, C_INTERNAL_ENTERED, C_TRI_TESTS, C_HIT_UPDATES,
               C_SPHERE_TESTS, C_SHADED_HITS, C_TEX_FETCHES, C_STACK_SPILLS, C_MAX_STACK,
               C_NODE_SLOTS, C_TRI_SLOTS, C_ADV_SLOTS, C_ADV_ACTIVE,
               C_IDLE_AT_LEAF, C_IDLE_WAITING, C_IDLE_DONE, C_VISITS_LT6, C_VISITS_LT9, C_VISITS_LT12, C_CERT_FALLBACKS, C_AUDITED, C_AUDIT_MISMATCHES, C_WAVE_TICKS,
               C_T_FIRST, C_T_HEAVY_EMPTY, C_T_LIGHT_EMPTY, C_T_LAST,      // wall-clock marks (100 MHz ticks), kept as maxima: the first three of ~t
               kNumCounters };

// status bits raised by the checked build
constexpr uint32_t kFlagBadNodeRef = 1u, kFlagBadTriSlot = 2u, kFlagBadMaterial = 4u, kFlagStackOverflow = 8u,
                   kFlagStepCap = 16u, kFlagBadBigLeaf = 32u;

}  // namespace dsrt
