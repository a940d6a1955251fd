// render_kernel.hip -- the per-pixel x spp sampling loop as a persistent wave64 kernel for gfx950.
//
// What it computes is src/gpu_render.cu:973-1031 of the reference (render_kernel and everything below it:
// rand01 :77-80, camera ray :941-968, ray_color :715-936, scene_hit :509-551, bvh_hit_closest :387-473,
// bbox_hit :285-315, hit_triangle_index :322-380, hit_sphere :478-504, the scatter functions :603-661, the
// sampling helpers :82-189, tex2D :232-259, tone map + store :1003-1030) -- same fp32 operations in the same
// order, so that with rng_mode 0 the bytes written are the bytes the reference's arithmetic defines.
//
// How it computes it is not the reference's one-thread-per-pixel loop:
//   * Persistent lanes.  A lane owns one pixel at a time (the reference's LCG makes the samples of a pixel one
//     serial stream, :990-999) and pulls the next from a global queue when it finishes, so a wave never idles behind
//     its slowest pixel.  Work items are 8x8-pixel blocks, handed out costliest tile first (scheduling pre-pass
//     below); with rng_mode 1 (a Philox sub-sequence per sample) the items are slices of a pixel's samples.
//   * A lane is a small state machine (path_machine.h).  The wave alternates between an ADVANCE phase (all lanes
//     that are not walking a ray step their state machine) and a TRAVERSE phase (all lanes with a live ray walk the
//     BVH together, "while-while": node visits together, parked leaves together); phases are left when the lane-slots
//     wasted by the lanes waiting for the other phase exceed what switching costs (ballot + popcount per iteration),
//     which keeps the loops about half full although bounce depths diverge from 1 to 50.
//   * Node visit = ONE 64-byte record with both child boxes, interleaved so the slab arithmetic runs on packed fp32
//     pairs (device_layout.h); the chosen child is not re-tested on entry and a postponed child carries its slab
//     entry distance on the stack, so a pop is a single compare.  Both are bit-identical to the reference's re-tests:
//     see `slab()` in device_math.h and the pop below.
//   * The traversal stack is a short stack in LDS, [entry][lane] so the 64 lanes of a wave hit 64 different banks;
//     entries beyond the LDS part spill to a per-lane strip in global memory (rare, behind a wave-uniform guard).
//     The continuation a lane postpones while its shadow ray is in flight also waits in LDS, not in registers.
//   * The hit record is assembled once per ray from (slot, t, u, v) instead of on every accepted candidate,
//     and shadow rays stop at the first accepted triangle (same boolean as the reference's closest-hit search).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math: IEEE div/sqrt are part of the contract).
#include <rocrand/rocrand_kernel.h>

#include "path_machine.h"

namespace dsrt {
#ifdef DSRT_DEVICE_LIBM
// Second compilation of this file (Makefile: hip_render_kernel_devlibm.o): the kernels whose arithmetic involves sinf / cosf / powf, built with the device math
// library's versions (device_math.h) and put in a namespace of their own -- DsrtRenderDesc.math_mode 1.  Everything that does not depend on those three functions
// (pre-pass, table, de-interleave, hash, test hooks) is compiled once, in the first pass.
namespace devlibm {
#endif
constexpr int kWavesPerBlock = 4;
// Register budget: 4 waves per SIMD (= 4 blocks per CU, which is also what the blocks' 31 KB of LDS allow).
#ifndef DSRT_WAVES_ATTR
#define DSRT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#endif

// Moller-Trumbore :336-353 on one pair record (two triangles, packed fp32), evaluated in full; the reference's early returns become one
// predicate per triangle.  Each `if (x) return false` is kept as `!(x)` so that NaNs fall the same way.  The test against `closest`
// (:353) is NOT part of this: it is applied, in order, by apply_pair.
__device__ __forceinline__ void moller_trumbore_pair(const float4* __restrict__ tp, F3 ro, F3 rd, v2f& t, v2f& u, v2f& v, bool& ok_a, bool& ok_b) {
    const float4 f0 = tp[0], f1 = tp[1], f2 = tp[2], f3 = tp[3];
    const float2 f4 = *reinterpret_cast<const float2*>(tp + 4);
    const v2f v0x = {f0.x, f0.y}, v0y = {f0.z, f0.w}, v0z = {f1.x, f1.y};
    const v2f e1x = {f1.z, f1.w}, e1y = {f2.x, f2.y}, e1z = {f2.z, f2.w};
    const v2f e2x = {f3.x, f3.y}, e2y = {f3.z, f3.w}, e2z = {f4.x, f4.y};
    const v2f pvx = rd.y * e2z - rd.z * e2y, pvy = rd.z * e2x - rd.x * e2z, pvz = rd.x * e2y - rd.y * e2x;   // cross(rd, e2)
    const v2f det = (e1x * pvx + e1y * pvy) + e1z * pvz;
    const v2f inv_det = {1.0f / det.x, 1.0f / det.y};
    const v2f tvx = ro.x - v0x, tvy = ro.y - v0y, tvz = ro.z - v0z;
    u = ((tvx * pvx + tvy * pvy) + tvz * pvz) * inv_det;
    const v2f qvx = tvy * e1z - tvz * e1y, qvy = tvz * e1x - tvx * e1z, qvz = tvx * e1y - tvy * e1x;          // cross(tvec, e1)
    v = ((rd.x * qvx + rd.y * qvy) + rd.z * qvz) * inv_det;
    t = ((e2x * qvx + e2y * qvy) + e2z * qvz) * inv_det;
    const v2f uv = u + v;
    ok_a = !(fabsf(det.x) < 1e-8f) && !(u.x < 0.0f) && !(u.x > 1.0f) && !(v.x < 0.0f) && !(uv.x > 1.0f) && !(t.x < kTMin);
    ok_b = !(fabsf(det.y) < 1e-8f) && !(u.y < 0.0f) && !(u.y > 1.0f) && !(v.y < 0.0f) && !(uv.y > 1.0f) && !(t.y < kTMin);
}

// The accepts of one pair record in the reference's order (:353, :371-379): A, then B against the `closest` A may have just lowered.
// `slot_a` = slot of triangle A.  Returns true when the walk is over (an any-hit shadow ray found its blocker).
template <bool COUNT, bool ANYHIT>
__device__ __forceinline__ bool apply_pair(Lane& ln, uint32_t* c, int slot_a, v2f t, v2f u, v2f v, bool ok_a, bool ok_b, bool has_b) {
    bool stop = false;
    if (COUNT) c[C_TRI_TESTS]++;
    if (ok_a && !(t.x > ln.closest)) {
        // (certified second tree, path_machine.h) an equality accept over an earlier hit is a TIE: the reference's answer would depend on its own order
        ln.aux = (t.x == ln.closest && ln.hit_slot >= 0) ? (ln.aux | kTie) : (ln.aux & ~kTie);
        ln.closest = t.x; ln.cull = t.x * ln.relax; ln.hit_slot = slot_a; ln.hit_u = u.x; ln.hit_v = v.x;
        if (COUNT) c[C_HIT_UPDATES]++;
        stop = ANYHIT && ln.state == ST_TRAV_SHADOW;
    }
    if (COUNT && has_b && !stop) c[C_TRI_TESTS]++;
    if (!stop && ok_b && !(t.y > ln.closest)) {          // an absent B is all zeros: det == 0, never ok
        ln.aux = (t.y == ln.closest && ln.hit_slot >= 0) ? (ln.aux | kTie) : (ln.aux & ~kTie);
        ln.closest = t.y; ln.cull = t.y * ln.relax; ln.hit_slot = slot_a + 1; ln.hit_u = u.y; ln.hit_v = v.y;
        if (COUNT) c[C_HIT_UPDATES]++;
        stop = ANYHIT && ln.state == ST_TRAV_SHADOW;
    }
    return stop;
}

// Node-loop iterations per look at the loop's votes (1: +1.4 % time, 3: no better than 2; profiles/r03/ab_node_loop_unroll.jsonl).
constexpr int kNodeUnroll = 2;

template <int K, bool COUNT, bool CHECKED, bool ANYHIT, int RNGMODE, bool PROBE, bool BATCH = false, bool LEAN = false>
__device__ __forceinline__ void render_body(const RenderArgs& args) {
    const DeviceScene& S = args.scene;
    __shared__ uint2 lds_stack[kWavesPerBlock][K + 1][64];       // entry K is a dump slot, see the node visit

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;

    __shared__ float lds_pend[kPendWords][kPendStride];
    Lane ln;
    ln.pend = &lds_pend[0][threadIdx.x];
    ln.aux = (uint32_t)lane;
    int& state = ln.state; int& cur = ln.cur; int& sp = ln.sp;
    uint32_t& steps = ln.steps;
    F3& ro = ln.ro; F3& rd = ln.rd; F3& rinv = ln.rinv;
    uint32_t c[kNumCounters];
#pragma unroll
    for (int i = 0; i < kNumCounters; ++i) c[i] = 0;
    uint32_t flags = 0;
    const unsigned long long t_enter = COUNT ? wall_clock64() : 0ull;
    if (lane == 0) mark_time<COUNT>(args, C_T_FIRST);

    for (;;) {
        // =====================================================================================
        // ADVANCE phase
        // =====================================================================================
        for (int budget = 0; budget < args.advance_budget; ++budget) {
            // lanes whose delegated shadow ray (path_machine.h) has not answered yet step aside; those whose answer came are back
            if (wave_any((ln.aux & kAwait) != 0u)) {
                if (ln.aux & kAwait) {
                    lane_handoff_acquire();
                    const bool answered = ln.pend[12 * kPendStride] != 0.0f;
                    if (state >= kParked) { if (answered) state -= kParked; }
                    else if (state < ST_TRAV_CLOSEST && !answered) state += kParked;
                }
            }
            if (!wave_any(state < ST_TRAV_CLOSEST)) break;
            if (COUNT) { c[C_ADV_SLOTS]++; if (state < ST_TRAV_CLOSEST) c[C_ADV_ACTIVE]++; }
            // idle lanes go through the step too: that is where they pick up shadow rays
            if (state < ST_TRAV_CLOSEST || state == ST_DONE) {
                // The pass reads its launch constants (camera, sun, sizes, scene pointers) from the kernel-argument segment WHERE IT USES THEM -- scalar loads through the
                // constant cache -- instead of keeping ~60 scalar registers alive across the whole kernel for them: the pointer goes through an empty asm, so the
                // compiler cannot hoist the loads out of the loop and then spill their results to vector lanes (105 spilled scalars, 170 v_readlane / v_writelane in this
                // pass, before; none now, and 112 VGPRs instead of 128.  Time: unchanged within the noise, profiles/r04/ab_kernarg_reload_*.jsonl).
                typedef const RenderArgs __attribute__((address_space(4)))* KernargPtr;
                KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kp));
                advance_step<COUNT, CHECKED, ANYHIT, RNGMODE, PROBE, BATCH, LEAN>(ln, *(const RenderArgs*)kp, c, flags);
            }
        }

        if (wave_all(state == ST_DONE)) break;

        if constexpr (!PROBE && RNGMODE == 0) {
            // Waves that hold a pixel of a heavy tile ask the SIMD's arbiter for priority over waves that only hold background pixels
            // (levels: device_api.hip).  Scheduling only.
            if (args.hot) {
                if (wave_any((ln.aux & kHot) != 0u)) __builtin_amdgcn_s_setprio(3);
                else __builtin_amdgcn_s_setprio(0);
            }
        }

        // =====================================================================================
        // TRAVERSE phase ("while-while"), written flat: per wave iteration every lane that is descending performs at most
        // one POP ATTEMPT and one NODE VISIT, both as straight predicated code (no inner loops, no nested branches); lanes
        // that reached a leaf PARK there until enough lanes are parked, then all parked leaves are intersected together,
        // one triangle index at a time.  Each lane still performs exactly the reference's sequence of box tests,
        // triangle tests and pops -- only WHEN a lane runs changes, never what it does.
        //   cur >= kRefBias: internal node to visit          cur == kRefPop: take the next postponed child
        //   cur <  0            : parked at a leaf          cur == kRefNone: no ray
        // =====================================================================================
        // Leaving a phase is decided by accumulated waste, in lane-slots: every wave iteration spent here costs the lanes
        // that are waiting for the OTHER phase one slot each; switching costs the lanes that are busy HERE one pass of the
        // other phase.  Switch when the first exceeds the second (the ratios are the relative lengths of the phases' code).
        // Invariant used for the votes below: a lane that is not walking a ray has cur == kRefNone, so "walking" and its
        // refinements are single compares on `cur` (a vote on one compare is the compare's own mask, no extra VALU work).
        for (int wait_waste = 0;;) {
            const int n_walk = __popcll(wave_ballot(cur != kRefNone));
            if (n_walk == 0) break;
            const int n_wait = __popcll(wave_ballot(state < ST_TRAV_CLOSEST));
            if (wait_waste * 10 >= n_walk * args.min_walk_iters) break;
            int leaf_waste = 0;

            // ---------------- phase I: pops and internal nodes ----------------
            for (;;) {
                const bool descending = cur > kRefNone;                                              // kRefPop or an internal node
                const int n_desc = __popcll(wave_ballot(descending));
                const int n_leaf = __popcll(wave_ballot(cur < 0));
                if (n_desc == 0 || leaf_waste * 10 >= n_desc * args.leaf_ratio4) break;
                // Two iterations per look at the votes: the loop's own bookkeeping (two compares, seven scalar instructions, two branches) is paid every other time.
                leaf_waste += kNodeUnroll * n_leaf;
                wait_waste += kNodeUnroll * n_wait;
#pragma unroll
                for (int rep = 0; rep < kNodeUnroll; ++rep) {           // (body not re-indented: it is the iteration described above)
                // (checked build) a reference in 0 .. kRefNone - 1 belongs to no class: such a lane would be counted as walking for ever.  Library-built records never hold one;
                // a corrupted record or stack word might.
                if (CHECKED && cur >= 0 && cur < kRefNone) { flags |= kFlagBadNodeRef; cur = kRefNone; }
                if (COUNT) { c[C_NODE_SLOTS]++; if (cur < 0) c[C_IDLE_AT_LEAF]++; if (state < ST_TRAV_CLOSEST || (cur == kRefNone && state <= ST_TRAV_SHADOW)) c[C_IDLE_WAITING]++; if (state == ST_DONE) c[C_IDLE_DONE]++; }

                // pop attempt: a postponed child is entered iff its entry distance is still in front of `closest`, which is
                // bbox_hit(node, ray, t_min, closest) for a box already known to be hit (src/gpu_render.cu:422-424, 462-468)
                {
                    const bool popping = cur == kRefPop;
                    const bool finished = popping && sp == 0;
                    // a ray that has emptied its stack: its state leaves ST_TRAV_* when the node loop is left (below), not here, every iteration
                    if (finished) cur = kRefNone;
                    if (popping != finished) {                 // popping with something on the stack (one compare of sp, not two)
                        --sp;
                        uint2 e = lds_stack[wave][sp < K ? sp : K][lane];
                        if (wave_any(sp >= K)) {               // wave-uniform guard: keeps the common path a plain ds_read_b64 (without it: +3 %)
                            if (sp >= K) e = args.spill[(size_t)(sp - K) * args.spill_stride + glane];
                        }
                        if (ln.cull > __uint_as_float(e.y)) cur = (int)e.x;                // (cull == closest on the reference tree: the reference's own test)
                    }
                }

                // node visit: both child boxes from one 64-byte record
                if (cur > kRefPop) {
                    if (CHECKED && (cur - kRefBias >= S.num_pairs || ++steps > kStepCap)) {
                        flags |= cur - kRefBias >= S.num_pairs ? kFlagBadNodeRef : kFlagStepCap;
                        cur = kRefNone; state -= (ST_TRAV_CLOSEST - ST_SHADE);
                    } else {
                        const float4* rec = reinterpret_cast<const float4*>(S.pairs_biased + ((uint32_t)cur << 6));     // 32-bit offset from a scalar base (num_pairs < 2^26 - 64, checked at upload)
                        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                        const int ref_l = __float_as_int(q3.x), ref_r = __float_as_int(q3.y);
                        if (COUNT) { c[C_NODES_ENTERED]++; c[C_INTERNAL_ENTERED]++; c[C_BOX_FETCHES] += 2; const int dpt = S.pair_depth[cur - kRefBias]; if (dpt < 6) c[C_VISITS_LT6]++; if (dpt < 9) c[C_VISITS_LT9]++; if (dpt < 12) c[C_VISITS_LT12]++; }
                        // Both boxes at once: every quantity below is a (left, right) pair in two adjacent registers, so the
                        // subtractions / multiplications are packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32: IEEE per component,
                        // same results as the scalar forms).  Record layout: q0 = (L.lo.x, R.lo.x, L.hi.x, R.hi.x), q1 = y, q2 = z.
                        // (Doing the swap of :305-307 by address -- six 8-byte loads at near/far offsets -- saves the twelve selects
                        // but costs three more memory instructions per visit and was 11 % slower: profiles/r01/README.md.)
                        const v2f lox = {q0.x, q0.y}, hix = {q0.z, q0.w}, loy = {q1.x, q1.y}, hiy = {q1.z, q1.w}, loz = {q2.x, q2.y}, hiz = {q2.z, q2.w};
                        const v2f ax = (lox - ro.x) * rinv.x, bx = (hix - ro.x) * rinv.x;      // bbox_hit :303-304
                        const v2f ay = (loy - ro.y) * rinv.y, by = (hiy - ro.y) * rinv.y;
                        const v2f az = (loz - ro.z) * rinv.z, bz = (hiz - ro.z) * rinv.z;
                        const bool nx = rinv.x < 0.0f, ny = rinv.y < 0.0f, nz = rinv.z < 0.0f;       // the swap of :305-307
                        const float t0xl = nx ? bx.x : ax.x, t1xl = nx ? ax.x : bx.x, t0xr = nx ? bx.y : ax.y, t1xr = nx ? ax.y : bx.y;
                        const float t0yl = ny ? by.x : ay.x, t1yl = ny ? ay.x : by.x, t0yr = ny ? by.y : ay.y, t1yr = ny ? ay.y : by.y;
                        const float t0zl = nz ? bz.x : az.x, t1zl = nz ? az.x : bz.x, t0zr = nz ? bz.y : az.y, t1zr = nz ? az.y : bz.y;
                        const float tl = fmaxf(fmaxf(kTMin, t0xl), fmaxf(t0yl, t0zl)), tr = fmaxf(fmaxf(kTMin, t0xr), fmaxf(t0yr, t0zr));
                        const bool hl = !(fminf(fminf(ln.cull, t1xl), fminf(t1yl, t1zl)) <= tl);
                        const bool hr = !(fminf(fminf(ln.cull, t1xr), fminf(t1yr, t1zr)) <= tr);
                        // nearer child by box centre along the ray :433-453 (only matters when both are hit).  The reference compares
                        //   d = ((c.x - o.x) * dir.x + (c.y - o.y) * dir.y) + (c.z - o.z) * dir.z,   c = 0.5f * (lo + hi)
                        // of the two children.  Computed here is 2 d, with the reference's roundings: s = lo + hi is the reference's sum (the x sums come with the record); the product
                        // by 0.5f is exact, and fma(-2, o, s) = fl(s - 2 o) = 2 fl(0.5 s - o), because scaling by two commutes with rounding; so do the
                        // products and sums that follow.  dL < dR <=> 2 dL < 2 dR, and three packed multiplications per visit are gone.  (The
                        // identity needs the reference's intermediates to be zero or normal numbers below 1.7e38: coordinates in metres are.)
                        const v2f m2 = {-2.0f, -2.0f}, ox2 = {ro.x, ro.x}, oy2 = {ro.y, ro.y}, oz2 = {ro.z, ro.z};
                        const v2f ux = __builtin_elementwise_fma(m2, ox2, (v2f){q3.z, q3.w}), uy = __builtin_elementwise_fma(m2, oy2, loy + hiy), uz = __builtin_elementwise_fma(m2, oz2, loz + hiz);
                        const v2f dc = (ux * rd.x + uy * rd.y) + uz * rd.z;
                        const bool left_near = dc.x < dc.y;
                        const bool both = hl && hr;
                        // the far child goes to stack[sp]; written unconditionally (slot sp is above the top, slot K is a dump
                        // slot for sp >= K), the stack only grows when both children were hit
                        const uint2 far = make_uint2((uint32_t)(left_near ? ref_r : ref_l), __float_as_uint(left_near ? tr : tl));
                        lds_stack[wave][sp < K ? sp : K][lane] = far;
                        if (both && sp >= K) {                             // the rare spill store (a vote shared by the two sites, at the top of the iteration, was 1 % slower)
                            if (sp - K < args.spill_entries) {
                                args.spill[(size_t)(sp - K) * args.spill_stride + glane] = far;
                                if (COUNT) c[C_STACK_SPILLS]++;
                            } else flags |= kFlagStackOverflow;
                        }
                        sp += both ? 1 : 0;
                        if (COUNT && (uint32_t)sp > c[C_MAX_STACK]) c[C_MAX_STACK] = (uint32_t)sp;
                        // next node, without branches: the left child if both were hit and it is the nearer one, or if it alone was hit
                        const bool take_left = hl && !(hr && !left_near);            // (written as mask logic: a select between two predicates compiles to five VALU instructions)
                        const int child = take_left ? ref_l : ref_r;
                        cur = (hl || hr) ? child : kRefPop;
                    }
                }
                }                                                       // rep
            }

            if (CHECKED && cur >= 0 && cur < kRefNone) { flags |= kFlagBadNodeRef; cur = kRefNone; }       // (a bad reference taken in the loop's last iteration)
            // rays that emptied their stack in the node loop: TRAV_CLOSEST -> SHADE, TRAV_SHADOW -> SHADOW_DONE (a lane in ST_TRAV_* without a node has just ended)
            if (cur == kRefNone && state >= ST_TRAV_CLOSEST && state <= ST_TRAV_SHADOW) state -= (ST_TRAV_CLOSEST - ST_SHADE);

            // ---------------- phase L: every lane parked at a leaf intersects it, triangle by triangle :413-420 ----------------
            const bool at_leaf = cur < 0;
            const unsigned long long leaf_mask = wave_ballot(at_leaf);
            if (leaf_mask != 0ull) {
                wait_waste += n_wait;
                int first = 0, count = 0;
                if (at_leaf) {
                    first = leaf_payload(cur);
                    count = leaf_code(cur) + 1;
                    if (count == 8) {
                        if (CHECKED && first >= S.num_big_leaves) { flags |= kFlagBadBigLeaf; first = 0; count = 0; }
                        else { const int2 bl = S.big_leaves[first]; first = bl.x; count = bl.y; }
                    }
                    if (CHECKED && (first < 0 || first + ((count + 1) >> 1) > S.num_tri_pairs || ++steps > kStepCap)) { flags |= kFlagBadTriSlot; count = 0; }
                    if (COUNT) c[C_NODES_ENTERED]++;
                    cur = kRefPop;
                }
                // Two triangles (one pair record) per step.  Every quantity is an (A, B) pair in adjacent registers; both
                // tests are evaluated in full against nothing but the ray, then APPLIED in the reference's order: A first, and
                // B against the `closest` A may have just lowered -- the reference's sequence of accepts, bit for bit.
                //
                // Second records are DEALT to the lanes that are not at a leaf.  The median-split builder stops at four triangles, so
                // nearly every leaf of a large mesh is two pair records, while at most a third of the wave is parked when this pass
                // runs: the loop used to run twice at a quarter of its lanes.  Evaluating a record needs the ray and the record, not
                // `closest`; only the accepts are ordered.  So the r-th lane that is not at a leaf evaluates record 1 of the r-th
                // parked lane that has one (the ray comes over by ds_bpermute; its own registers are untouched), hands (t, u, v, ok)
                // of both triangles back the same way, and the owner applies A, B of record 0 and then A, B of record 1 -- the
                // reference's order.  Taken only when EVERY second record finds a lane (otherwise the second trip is paid anyway).
                // Leaves of more than four triangles (coincident centroids) finish in the plain loop below.
                int done = 0;                                              // triangles of this lane's leaf dealt with so far
                if (args.deal_leaves) {
                    const bool want = count > 2;
                    const unsigned long long want_mask = wave_ballot(want);
                    const int n_want = __popcll(want_mask);
                    if (n_want > 0 && n_want <= 64 - __popcll(leaf_mask)) {
                        const uint32_t rank_w = __builtin_amdgcn_mbcnt_hi((uint32_t)(want_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_mask, 0u));
                        const uint32_t rank_i = __builtin_amdgcn_mbcnt_hi((uint32_t)(~leaf_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~leaf_mask, 0u));
                        float* const table = ln.pend + 13 * kPendStride - lane;          // the wave's 64-entry table (free outside advance_step)
                        const bool helping = !at_leaf && rank_i < (uint32_t)n_want;
                        if (want) table[rank_w] = __int_as_float(lane);
                        lane_handoff_release();
                        int src = lane;                                      // whose ray this lane evaluates: its own, or its owner's
                        if (helping) { lane_handoff_acquire(); src = __float_as_int(table[rank_i]); table[rank_i] = __int_as_float(lane); }
                        lane_handoff_release();
                        const int src4 = src << 2;
                        const F3 lro = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(ro.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(ro.y))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(ro.z))));
                        const F3 lrd = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(rd.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(rd.y))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(rd.z))));
                        const int lpair = __builtin_amdgcn_ds_bpermute(src4, first) + (helping ? 1 : 0);
                        v2f t, u, v;                                         // meaningful only where `ok` says so (unset elsewhere: six moves per pass)
                        int ok = 0;                                          // bit 0: A passed every test but the one against `closest`, bit 1: B
                        if (COUNT) c[C_TRI_SLOTS] += 2;
                        if (count > 0 || helping) {                          // (count is 0 in the lanes that are not at a leaf: two plain compares, not a select between predicates)
                            bool ok_a, ok_b;
                            moller_trumbore_pair(S.tri_pairs + (size_t)lpair * 5, lro, lrd, t, u, v, ok_a, ok_b);
                            ok = (ok_a ? 1 : 0) | (ok_b ? 2 : 0);
                        }
                        // the owner fetches its helper's results (a lane without one reads its own and ignores them)
                        int from = lane;
                        if (want) { lane_handoff_acquire(); from = __float_as_int(table[rank_w]); }
                        const int from4 = from << 2;
                        const v2f t2 = {__int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(t.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(t.y)))};
                        const v2f u2 = {__int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(u.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(u.y)))};
                        const v2f v2 = {__int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(v.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(from4, __float_as_int(v.y)))};
                        const int ok2 = __builtin_amdgcn_ds_bpermute(from4, ok);
                        if (at_leaf && count > 0) {
                            bool stop = apply_pair<COUNT, ANYHIT>(ln, c, first * 2, t, u, v, (ok & 1) != 0, (ok & 2) != 0, count > 1);
                            if (!stop && want) stop = apply_pair<COUNT, ANYHIT>(ln, c, first * 2 + 2, t2, u2, v2, (ok2 & 1) != 0, (ok2 & 2) != 0, count > 3);
                            done = want ? 4 : 2;
                            if (stop) { count = 0; cur = kRefNone; state = ST_SHADOW_DONE; }
                        }
                    }
                }
                for (int i = done; wave_any(i < count); i += 2) {
                    if (COUNT) c[C_TRI_SLOTS] += 2;
                    if (i < count) {
                        const int pair = first + (i >> 1);
                        v2f t, u, v;
                        bool ok_a, ok_b;
                        moller_trumbore_pair(S.tri_pairs + (size_t)pair * 5, ro, rd, t, u, v, ok_a, ok_b);
                        if (apply_pair<COUNT, ANYHIT>(ln, c, pair * 2, t, u, v, ok_a, ok_b, i + 1 < count)) { count = 0; cur = kRefNone; state = ST_SHADOW_DONE; }
                    }
                }
            }
        }
    }

    if (COUNT && lane == 0) c[C_WAVE_TICKS] = (uint32_t)(wall_clock64() - t_enter);
    if (lane == 0) mark_time<COUNT>(args, C_T_LAST);
    flush_counters<COUNT>(args, c);
    if (flags) atomicOr(args.flags, flags);
}

// LEAN (path_machine.h): the instantiation for scenes of Lambertian triangles only, chosen by the host from what upload found in the scene.  Production builds only:
// the counting and checked builds are the general code.
template <int K, bool COUNT, bool CHECKED, bool ANYHIT, int RNGMODE, bool LEAN = false>
__global__ void __launch_bounds__(64 * kWavesPerBlock) DSRT_WAVES_ATTR dsrt_render_kernel(const RenderArgs args) {
    render_body<K, COUNT, CHECKED, ANYHIT, RNGMODE, false, false, LEAN>(args);
}

// Batch launch: many frames of one scene as one pool of work (path_machine.h, ST_FETCH).
template <int RNGMODE, bool LEAN = false>
__global__ void __launch_bounds__(64 * kWavesPerBlock) DSRT_WAVES_ATTR dsrt_render_batch_kernel(const RenderArgs args) {
    render_body<8, false, false, true, RNGMODE, false, true, LEAN>(args);
}

// Fills the count fields of the batch table (entries f and frames + f belong to frame f) from the sched words every frame's pre-pass
// wrote (sched_stride apart), and the running item totals.
#ifndef DSRT_DEVICE_LIBM
__global__ void dsrt_batch_table_kernel(BatchFrame* __restrict__ table, const uint32_t* __restrict__ sched, uint32_t sched_stride, uint32_t frames,
                                        uint32_t tt, int rng_mode, int spp, int light_chunk_len, uint32_t* __restrict__ total_items) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long acc = 0;
    const uint32_t light_slices = rng_mode == 1 ? (uint32_t)((spp + light_chunk_len - 1) / light_chunk_len) : 1u;
    for (uint32_t e = 0; e < 2u * frames; ++e) {
        const uint32_t f = e < frames ? e : e - frames;
        const uint32_t* s = sched + (size_t)f * sched_stride;
        const uint32_t n_heavy = s[0], n_live = s[1];
        const uint32_t top = n_heavy ? (n_heavy >> 3 ? n_heavy >> 3 : 1u) : 0u;           // the first eighth of the heavy tiles, at least one
        BatchFrame& b = table[e];
        if (e < frames) { b.n_heavy = top; b.n_live = top; }
        else { b.n_heavy = n_heavy - top; b.n_live = n_live - top; b.order_base += top; }
        b.slices = rng_mode == 1 ? s[3] : 1u; b.chunk_len = rng_mode == 1 ? s[4] : (uint32_t)spp;
        acc += (unsigned long long)b.n_heavy * tt * b.slices + (unsigned long long)(b.n_live - b.n_heavy) * tt * light_slices;
        b.item_end = acc > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)acc;           // (the host bounds frames x pixels x 16 below 2^32)
    }
    *total_items = acc > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)acc;
}

#endif  // !DSRT_DEVICE_LIBM

// The probe launch of the pre-pass: the same body at a couple of samples per pixel, adding the rays every pixel needed to its
// tile's entry of args.tile_work.  Its own kernel symbol, so that profiles keep it apart from the frame's launch.
template <bool LEAN>
__global__ void __launch_bounds__(64 * kWavesPerBlock) DSRT_WAVES_ATTR dsrt_probe_kernel(const RenderArgs args) {
    render_body<8, false, false, true, 0, true, false, LEAN>(args);
}

#ifndef DSRT_DEVICE_LIBM
// ---------------------------------------------------------------------------------------------------------------
// Scheduling pre-pass: how expensive is each screen tile?  One wave per 8x8 pixel block traces the un-jittered centre
// ray of every pixel with an any-hit walk; the tile's cost is the number of pixels that see geometry.  The render
// kernel then hands tiles out costliest first, so the pixels that are 1000-sample serial chains start early and the
// end of the frame is filled with background pixels instead of a long tail.  This changes only the ORDER in which
// pixels are rendered, never a pixel's value (each pixel depends on nothing but its own coordinates and the scene).
//
// Empty tiles.  With `cull` set, a block whose sample footprint cannot reach the root box is left out of the order
// altogether and its pixels keep the zeros the output was cleared to.  That is exact, not approximate: every sample of such
// a pixel fails the root test of bvh_hit_closest (:394-410), ray_color returns black (:744-747), the sum is 0 and the tone
// map of 0 is byte 0 -- whatever the pixel's random numbers were, and no other pixel reads them.  "Cannot reach" is decided
// conservatively: the pyramid of rays through the block GROWN BY ONE PIXEL on every side (the jitter of :995-996 keeps a
// sample inside its pixel, at most on its far edge after rounding) is tested in double precision against the eight corners
// of the root box; a block is dropped only if one side plane of that pyramid has the whole box outside it, or the box is
// behind the camera.  One pixel is 5e-4 of the image width; the float rounding of the reference's ray set-up and slab test
// is 1e-7 relative, so a ray the reference would let into the root box is never culled.  Not applied when the scene has
// spheres (they are tested outside the BVH, :527-548) and not in counting builds (their counters include these samples).
// cost word: bit 31 = the tile has at least one block that is not provably empty; low bits = pixels that see geometry.
// ---------------------------------------------------------------------------------------------------------------
__device__ bool block_footprint_misses_root(const DeviceScene& S, const FrameParams& P, int x0, int ky0) {
    const double o[3] = {P.cam[kCamOrigin + 0], P.cam[kCamOrigin + 1], P.cam[kCamOrigin + 2]};
    const double ulo = (double)(x0 - 1) / (double)(P.width - 1), uhi = (double)(x0 + 9) / (double)(P.width - 1);
    const double vlo = (double)(ky0 - 1) / (double)(P.height - 1), vhi = (double)(ky0 + 9) / (double)(P.height - 1);
    const double us[4] = {ulo, uhi, uhi, ulo}, vs[4] = {vlo, vlo, vhi, vhi};
    double dir[4][3], mid[3] = {0, 0, 0};
    for (int c = 0; c < 4; ++c)
        for (int a = 0; a < 3; ++a) {
            dir[c][a] = ((double)P.cam[kCamLlc + a] + us[c] * (double)P.cam[kCamHorizontal + a] + vs[c] * (double)P.cam[kCamVertical + a]) - o[a];
            mid[a] += dir[c][a];
        }
    double rel[8][3];
    for (int j = 0; j < 8; ++j)
        for (int a = 0; a < 3; ++a) rel[j][a] = (double)((j >> a) & 1 ? S.root_hi[a] : S.root_lo[a]) - o[a];
    bool behind = true;
    for (int j = 0; j < 8; ++j) behind = behind && (mid[0] * rel[j][0] + mid[1] * rel[j][1] + mid[2] * rel[j][2] <= 0.0);
    if (behind) return true;
    for (int c = 0; c < 4; ++c) {
        const double* a = dir[c];
        const double* b = dir[(c + 1) & 3];
        double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double s = n[0] * mid[0] + n[1] * mid[1] + n[2] * mid[2];          // orient the side plane: inside is positive
        if (s == 0.0) continue;                                                   // degenerate pyramid: never cull on it
        if (s < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        bool outside = true;
        for (int j = 0; j < 8; ++j) outside = outside && (n[0] * rel[j][0] + n[1] * rel[j][1] + n[2] * rel[j][2] < 0.0);
        if (outside) return true;
    }
    return false;
}

// (batch launches: blockIdx.y is the frame; its camera comes from the batch table and its cost words lie `stride` further on)
__global__ void __launch_bounds__(256) dsrt_tile_cost_kernel(const DeviceScene S, const FrameParams P0, uint32_t* __restrict__ cost, int cull,
                                                             const BatchFrame* __restrict__ batch, uint32_t stride) {
    FrameParams P = P0;
    if (batch) {
        for (int a = 0; a < 12; ++a) P.cam[a] = batch[blockIdx.y].cam[a];
        cost += (size_t)blockIdx.y * stride;
    }
    const uint32_t blocks_per_tile = (uint32_t)((P.tile >> 3) * (P.tile >> 3));
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave_id >= (uint32_t)P.local_tiles * blocks_per_tile) return;
    const uint32_t k = wave_id / blocks_per_tile, sub = wave_id % blocks_per_tile, per_row = (uint32_t)P.tile >> 3;
    const uint32_t g = k * (uint32_t)P.shard_count + (uint32_t)P.shard_rank;
    const uint32_t tx = g % (uint32_t)P.tiles_x, ty = g / (uint32_t)P.tiles_x;
    const int x = (int)(tx * (uint32_t)P.tile + (sub % per_row) * 8u + (lane & 7u));
    const int row = (int)(ty * (uint32_t)P.tile + (sub / per_row) * 8u + (lane >> 3));
    bool hit = false;
    {
        // wave-uniform: this 8x8 block's first column and its lowest kernel row (ky = H - 1 - row; the block's rows are row0..row0+7)
        const int bx0 = (int)(tx * (uint32_t)P.tile + (sub % per_row) * 8u), brow0 = (int)(ty * (uint32_t)P.tile + (sub / per_row) * 8u);
        const bool empty = cull && S.num_spheres == 0 && (S.root_ref == kRefNone || block_footprint_misses_root(S, P, bx0, P.height - 1 - (brow0 + 7)));
        if (empty) return;
        if (lane == 0) atomicOr(&cost[k], 0x80000000u);
    }
    if (x < P.width && row < P.height) {
        const int ky = P.height - 1 - row;
        const float u = ((float)x + 0.5f) / (float)(P.width - 1), v = ((float)ky + 0.5f) / (float)(P.height - 1);
        const F3 ro = ld3(P.cam + kCamOrigin);
        const F3 rd = ((ld3(P.cam + kCamLlc) + (ld3(P.cam + kCamHorizontal) * u)) + (ld3(P.cam + kCamVertical) * v)) - ro;
        const F3 rinv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        for (int i = 0; i < S.num_spheres && !hit; ++i) { float t; F3 n; hit = hit_sphere(S.spheres[i], ro, rd, kTMax, t, n); }
        float t_entry;
        if (!hit && S.root_ref != kRefNone && slab(ld3(S.root_lo), ld3(S.root_hi), ro, rinv, kTMax, t_entry)) {
            int stack[64];
            int sp = 0, cur = S.root_ref;
            for (int guard = 0; guard < (1 << 20) && !hit; ++guard) {
                if (cur >= 0) {
                    const float4* rec = S.pairs + (size_t)(cur - kRefBias) * 4;
                    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                    float tl, tr;
                    const bool hl = slab(mk(q0.x, q1.x, q2.x), mk(q0.z, q1.z, q2.z), ro, rinv, kTMax, tl);
                    const bool hr = slab(mk(q0.y, q1.y, q2.y), mk(q0.w, q1.w, q2.w), ro, rinv, kTMax, tr);
                    const int rl = __float_as_int(q3.x), rr = __float_as_int(q3.y);
                    if (hl && hr) { if (sp < 64) stack[sp++] = rr; cur = rl; }
                    else if (hl) cur = rl;
                    else if (hr) cur = rr;
                    else { if (sp == 0) break; cur = stack[--sp]; }
                } else {
                    int first = leaf_payload(cur), count = leaf_code(cur) + 1;
                    if (count == 8) { const int2 bl = S.big_leaves[first]; first = bl.x; count = bl.y; }
                    for (int i = 0; i < count && !hit; ++i) {
                        const float* tp = reinterpret_cast<const float*>(S.tri_pairs + (size_t)(first + (i >> 1)) * 5) + (i & 1);
                        const F3 v0 = mk(tp[0], tp[2], tp[4]), e1 = mk(tp[6], tp[8], tp[10]), e2 = mk(tp[12], tp[14], tp[16]);
                        const F3 pvec = cross(rd, e2);
                        const float det = dot(e1, pvec);
                        const float inv_det = 1.0f / det;
                        const F3 tvec = ro - v0;
                        const float uu = dot(tvec, pvec) * inv_det;
                        const F3 qvec = cross(tvec, e1);
                        const float vv = dot(rd, qvec) * inv_det;
                        const float t = dot(e2, qvec) * inv_det;
                        hit = !(fabsf(det) < 1e-8f) && !(uu < 0.0f) && !(uu > 1.0f) && !(vv < 0.0f) && !(uu + vv > 1.0f) && !(t < kTMin) && !(t > kTMax);
                    }
                    if (sp == 0) break;
                    cur = stack[--sp];
                }
            }
        }
    }
    const uint32_t n = (uint32_t)__popcll(wave_ballot(hit));
    if (lane == 0 && n) atomicAdd(&cost[k], n);
}

// One block: counting sort of the shard's live tiles by cost, costliest first (65 bins; order inside a bin does not matter).
// (batch launches: one block per frame, the frames' arrays `stride` apart)
__global__ void __launch_bounds__(1024) dsrt_tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, int n, int tile,
                                                               uint32_t* __restrict__ sched, uint32_t items_per_pixel, uint32_t resident_lanes, int spp, uint32_t stride) {
    cost += (size_t)blockIdx.x * stride; order += (size_t)blockIdx.x * stride; sched += (size_t)blockIdx.x * stride;
    __shared__ uint32_t bins[65], cursor[65];
    const uint32_t full = (uint32_t)(tile * tile);
    for (int b = threadIdx.x; b < 65; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x)
        if (cost[t]) atomicAdd(&bins[64u - min(64u, (cost[t] & 0x7FFFFFFFu) * 64u / full)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int b = 0; b < 65; ++b) { cursor[b] = acc; acc += bins[b]; }
        const uint32_t n_heavy = acc - bins[64];
        // lanes per wave that start on the heavy queue: all of them once the heavy items outnumber the resident lanes, otherwise
        // just enough to deal the heavy items out over every resident wave (see ST_FETCH in path_machine.h)
        // rng_mode 1 (items_per_pixel > 1): how many slices a heavy pixel is cut into.  8 when the heavy pixels alone fill the chip; with
        // fewer of them (a far frame) up to 64, so that their work spreads over more WAVES -- sample stealing only evens out a wave
        uint32_t slices = items_per_pixel, chunk_len = spp;
        if (items_per_pixel > 1u) {
            const unsigned long long heavy_pixels = (unsigned long long)n_heavy * full;
            while (slices < 64u && slices * 2u <= (uint32_t)spp && heavy_pixels * slices * 2ull <= resident_lanes) slices *= 2u;
            chunk_len = ((uint32_t)spp + slices - 1u) / slices;
            slices = ((uint32_t)spp + chunk_len - 1u) / chunk_len;
        }
        const unsigned long long heavy_items = (unsigned long long)n_heavy * full * slices;
        uint32_t spread = 64;
        if (heavy_items < resident_lanes) spread = (uint32_t)((heavy_items * 64ull + resident_lanes - 1) / resident_lanes);
        sched[0] = n_heavy; sched[1] = acc; sched[2] = spread < 1 ? 1u : spread; sched[3] = slices; sched[4] = chunk_len;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x)
        if (cost[t]) order[atomicAdd(&cursor[64u - min(64u, (cost[t] & 0x7FFFFFFFu) * 64u / full)], 1u)] = (uint32_t)t;
}

// Refinement of the order by measured cost: a probe launch of the render kernel itself (a few samples per pixel, same state
// machine) has left the number of rays it traced for every tile in `work`; the tiles that see geometry -- the first sched[0]
// entries of `order` -- are re-sorted by it, most work first, so the pixels with the longest serial sample chains start at
// once instead of wherever their tile's coverage count put them.  Scheduling only.
__global__ void __launch_bounds__(1024) dsrt_tile_reorder_kernel(const uint32_t* __restrict__ work, uint32_t* __restrict__ order, uint32_t* __restrict__ tmp,
                                                                 const uint32_t* __restrict__ sched) {
    __shared__ uint32_t bins[256], cursor[256], wmax;
    const uint32_t n = sched[0];
    // Fewer heavy pixels than resident lanes (a mid-distance frame; one rank's share): every heavy pixel starts in the first instant
    // whatever the order, and the order only decides which waves -- fetching at the same moment, so mostly neighbours on a CU --
    // hold the long chains.  Sorted, the costliest tiles sit side by side; dealt costliest, cheapest, second costliest, ... a long
    // chain's neighbours finish early and leave it the SIMD.  Interleaved medians at 1080p x 1000 (sorted / dealt / coverage order /
    // no probe): frame 75 387 / 361 / 374 / 374 ms, frame 85 400 / 388 / 368 / 373, frame 90 505 / 500 / 512 / 479, one of 8 shares of
    // the near frame 564 / 563 / 625 / 610 (profiles/r02/ab_small_regime_order.jsonl); one setting spreads +-5 % on such frames.
    const bool all_start_at_once = sched[2] < 64u;
    for (int b = threadIdx.x; b < 256; b += blockDim.x) bins[b] = 0;
    if (threadIdx.x == 0) wmax = 1u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicMax(&wmax, work[order[i]]);
    __syncthreads();
    const uint32_t top = wmax;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        tmp[i] = order[i];
        atomicAdd(&bins[255u - (uint32_t)((unsigned long long)work[order[i]] * 255ull / top)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int b = 0; b < 256; ++b) { cursor[b] = acc; acc += bins[b]; } }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t t = tmp[i];
        order[atomicAdd(&cursor[255u - (uint32_t)((unsigned long long)work[t] * 255ull / top)], 1u)] = t;
    }
    if (all_start_at_once) {             // costliest, cheapest, second costliest, second cheapest, ...
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) tmp[i] = order[i];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) order[i] = (i & 1u) ? tmp[n - 1u - (i >> 1)] : tmp[i >> 1];
    }
}

hipError_t launch_tile_reorder(const uint32_t* work, uint32_t* order, uint32_t* tmp, const uint32_t* sched, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_tile_reorder_kernel, dim3(1), dim3(1024), 0, stream, work, order, tmp, sched);
    return hipGetLastError();
}

hipError_t launch_tile_order(const DeviceScene& S, const FrameParams& P, uint32_t* cost, uint32_t* order, uint32_t* sched, uint32_t items_per_pixel,
                             uint32_t resident_lanes, bool cull, hipStream_t stream, const BatchFrame* batch = nullptr, uint32_t frames = 1, uint32_t stride = 0) {
    const uint32_t waves = (uint32_t)P.local_tiles * (uint32_t)((P.tile >> 3) * (P.tile >> 3));
    for (uint32_t f = 0; f < frames; ++f) {                           // (the cost words only: the arrays carry the sched words behind them)
        hipError_t e = hipMemsetAsync(cost + (size_t)f * stride, 0, (size_t)P.local_tiles * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(dsrt_tile_cost_kernel, dim3((waves + 3) / 4, frames), dim3(256), 0, stream, S, P, cost, cull ? 1 : 0, batch, stride);
    hipLaunchKernelGGL(dsrt_tile_order_kernel, dim3(frames), dim3(1024), 0, stream, (const uint32_t*)cost, order, P.local_tiles, P.tile, sched, items_per_pixel, resident_lanes, P.spp, stride);
    return hipGetLastError();
}

// Tile-major shards -> image order, on the root after the gather.
__global__ void dsrt_deinterleave_kernel(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ image, int W, int H, int tile, int tiles_x,
                                         int shard_count, size_t shard_stride_bytes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int x = (int)(i % W), row = (int)(i / W);
    const int g = (row / tile) * tiles_x + (x / tile);
    const int rank = g % shard_count, k = g / shard_count;
    const size_t src = (size_t)rank * shard_stride_bytes + ((size_t)k * tile * tile + (size_t)(row % tile) * tile + (x % tile)) * 3;
    image[i * 3 + 0] = gathered[src + 0];
    image[i * 3 + 1] = gathered[src + 1];
    image[i * 3 + 2] = gathered[src + 2];
}

__global__ void dsrt_math_kernel(int fn, const float* __restrict__ x, float y, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? dsrt_sinf(x[i]) : (fn == 1 ? dsrt_cosf(x[i]) : dsrt_powf(x[i], y));
}

// Test hook (dsrt_selftest_devkat): the render kernel's own material / frame helpers of device_math.h on explicit inputs, 12 words in and
// 12 words out per case, so that they can be compared with known answers produced by the reference's host code (tests/golden/ref_matkat.json).
//   fn 0 reflect            in: v[3] n[3]                              out: r[3]
//   fn 1 refract            in: v[3] n[3] eta                          out: r[3]
//   fn 2 scatter_metal      in: dir[3] n[3] fuzz, LCG state (bits)     out: dir[3], went on (1.0 / 0.0), LCG state after (bits)
//   fn 3 scatter_dielectric in: dir[3] n[3] ref_idx, state, front face out: dir[3], -, LCG state after (bits)
//   fn 4 build_onb          in: n[3]                                   out: u[3] v[3] w[3]
//   fn 5 schlick            in: cosine, ratio                          out: reflectance
__global__ void dsrt_devkat_kernel(int fn, const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * 12;
    float* o = out + (size_t)i * 12;
    const F3 v = ld3(a), nn = ld3(a + 3);
    uint32_t rng = __float_as_uint(a[7]);
    if (fn == 0) { const F3 r = reflect(v, nn); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
    else if (fn == 1) { const F3 r = refract(v, nn, a[6]); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
    else if (fn == 2) { F3 d; const bool ok = scatter_metal(v, nn, a[6], rng, d); o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = ok ? 1.0f : 0.0f; o[4] = __uint_as_float(rng); }
    else if (fn == 3) { const F3 d = scatter_dielectric(v, nn, __float_as_uint(a[8]) != 0u, a[6], rng); o[0] = d.x; o[1] = d.y; o[2] = d.z; o[4] = __uint_as_float(rng); }
    else if (fn == 4) { F3 u, vv, w; build_onb(v, u, vv, w); o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = vv.x; o[4] = vv.y; o[5] = vv.z; o[6] = w.x; o[7] = w.y; o[8] = w.z; }
    else { o[0] = schlick(a[0], a[1]); }
}

hipError_t launch_devkat(int fn, const float* in, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_devkat_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, fn, in, out, n);
    return hipGetLastError();
}

#endif  // !DSRT_DEVICE_LIBM

// rng_mode 1: a pixel's samples were summed as integers in units of 2^-20 (path_machine.h, end_sample); here the mean, the
// reference's tone map and the 8-bit store (:1003-1030).  Pixels nobody sampled (culled tiles, padding) hold zero sums: black.
__global__ void dsrt_resolve_kernel(const unsigned long long* __restrict__ sums, int spp, float inv_gamma, size_t n_pixels,
                                    uint8_t* __restrict__ out_rgb8, float* __restrict__ out_f32) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const double unit = 1.0 / 1048576.0 / (double)spp;
    F3 col = mk((float)((double)sums[i * 3 + 0] * unit), (float)((double)sums[i * 3 + 1] * unit), (float)((double)sums[i * 3 + 2] * unit));
    col = mk(fmaxf(col.x, 0.0f), fmaxf(col.y, 0.0f), fmaxf(col.z, 0.0f));
    col = mk(fminf(col.x, 10.0f), fminf(col.y, 10.0f), fminf(col.z, 10.0f));
    col = mk(dsrt_powf(col.x, inv_gamma), dsrt_powf(col.y, inv_gamma), dsrt_powf(col.z, inv_gamma));
    col = clamp01(col);
    out_rgb8[i * 3 + 0] = (unsigned char)(255.99f * col.x);
    out_rgb8[i * 3 + 1] = (unsigned char)(255.99f * col.y);
    out_rgb8[i * 3 + 2] = (unsigned char)(255.99f * col.z);
    if (out_f32) { out_f32[i * 3 + 0] = col.x; out_f32[i * 3 + 1] = col.y; out_f32[i * 3 + 2] = col.z; }
}

// ---- launchers (called from device_api.hip) -----------------------------------------------------------
template <int K, int RNGMODE>
static hipError_t launch_k(const RenderArgs& a, int blocks, bool count, bool checked, bool anyhit, bool lean, hipStream_t stream) {
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    if (!count && !checked && lean) {
        hipLaunchKernelGGL((dsrt_render_kernel<K, false, false, true, RNGMODE, true>), grid, block, 0, stream, a);
    } else if (count) {
        if (anyhit) hipLaunchKernelGGL((dsrt_render_kernel<K, true, true, true, RNGMODE>), grid, block, 0, stream, a);
        else        hipLaunchKernelGGL((dsrt_render_kernel<K, true, true, false, RNGMODE>), grid, block, 0, stream, a);
    } else if (checked) {
        hipLaunchKernelGGL((dsrt_render_kernel<K, false, true, true, RNGMODE>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((dsrt_render_kernel<K, false, false, true, RNGMODE>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_render_batch(const RenderArgs& a, int rng_mode, int blocks, bool lean, hipStream_t stream) {
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    if (rng_mode == 0) { if (lean) hipLaunchKernelGGL((dsrt_render_batch_kernel<0, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((dsrt_render_batch_kernel<0, false>), grid, block, 0, stream, a); }
    else if (rng_mode == 1) { if (lean) hipLaunchKernelGGL((dsrt_render_batch_kernel<1, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((dsrt_render_batch_kernel<1, false>), grid, block, 0, stream, a); }
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

#ifndef DSRT_DEVICE_LIBM
hipError_t launch_batch_table(BatchFrame* table, const uint32_t* sched, uint32_t sched_stride, uint32_t frames, uint32_t tt, int rng_mode, int spp, int light_chunk_len,
                              uint32_t* total_items, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_batch_table_kernel, dim3(1), dim3(64), 0, stream, table, sched, sched_stride, frames, tt, rng_mode, spp, light_chunk_len, total_items);
    return hipGetLastError();
}

#endif

hipError_t launch_probe(const RenderArgs& a, int blocks, bool lean, hipStream_t stream) {
    if (lean) hipLaunchKernelGGL(dsrt_probe_kernel<true>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, stream, a);
    else hipLaunchKernelGGL(dsrt_probe_kernel<false>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs& a, int lds_entries, int rng_mode, int blocks, bool count, bool checked, bool anyhit, bool lean, hipStream_t stream) {
    if (lds_entries != 8) return hipErrorInvalidValue;     // the only short-stack size built
    if (rng_mode == 0) return launch_k<8, 0>(a, blocks, count, checked, anyhit, lean, stream);
    if (rng_mode == 1) return launch_k<8, 1>(a, blocks, count, checked, anyhit, lean, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_resolve(const unsigned long long* sums, int spp, float inv_gamma, size_t n_pixels, uint8_t* out_rgb8, float* out_f32, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_resolve_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, sums, spp, inv_gamma, n_pixels, out_rgb8, out_f32);
    return hipGetLastError();
}

#ifndef DSRT_DEVICE_LIBM
// Test hook: the first n words of Philox sub-sequence `sub` from our stateless form and from rocRAND's own device engine.
__global__ void dsrt_philox_kernel(unsigned long long seed, unsigned long long sub, int n, uint32_t* __restrict__ ours, uint32_t* __restrict__ theirs) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub, 0, &st);
    for (int i = 0; i < n; ++i) {
        ours[i] = philox4x32_10_word((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)i >> 2, 0u, (uint32_t)sub, (uint32_t)(sub >> 32), (uint32_t)i & 3u);
        theirs[i] = rocrand(&st);
    }
}

hipError_t launch_philox(unsigned long long seed, unsigned long long sub, int n, uint32_t* ours, uint32_t* theirs, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_philox_kernel, dim3(1), dim3(64), 0, stream, seed, sub, n, ours, theirs);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const uint8_t* gathered, uint8_t* image, int W, int H, int tile, int tiles_x, int shard_count,
                               size_t shard_stride_bytes, hipStream_t stream) {
    const size_t n = (size_t)W * H;
    hipLaunchKernelGGL(dsrt_deinterleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, image, W, H, tile, tiles_x,
                       shard_count, shard_stride_bytes);
    return hipGetLastError();
}

// 128-bit position-dependent content hash of a word array (the drop-in layer's "same scene as last frame?" test, device_api.hip):
// two independent 64-bit mixes of (word, index, salt), summed -- addition commutes, so blocks may finish in any order.
__global__ void __launch_bounds__(256) dsrt_content_hash_kernel(const uint32_t* __restrict__ words, size_t n, uint64_t salt, uint64_t* __restrict__ out2) {
    uint64_t a = 0, b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = ((uint64_t)words[i] << 32 | (uint64_t)(uint32_t)i) ^ salt ^ ((uint64_t)(i >> 32) * 0xD6E8FEB86659FD93ull);
        x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
        a += x;
        uint64_t y = x ^ 0xA0761D6478BD642Full;
        y ^= y >> 29; y *= 0xBF58476D1CE4E5B9ull; y ^= y >> 32;
        b += y * 0x94D049BB133111EBull;
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd((unsigned long long*)&out2[0], (unsigned long long)a); atomicAdd((unsigned long long*)&out2[1], (unsigned long long)b); }
}

hipError_t launch_content_hash(const uint32_t* words, size_t n_words, uint64_t salt, uint64_t* d_hash2, hipStream_t stream) {
    if (!n_words) return hipSuccess;
    const size_t want = (n_words + 255) / 256;
    hipLaunchKernelGGL(dsrt_content_hash_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, stream, words, n_words, salt, d_hash2);
    return hipGetLastError();
}

hipError_t launch_math(int fn, const float* x, float y, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, x, y, out, n);
    return hipGetLastError();
}

int kernel_waves_per_block() { return kWavesPerBlock; }

#endif  // !DSRT_DEVICE_LIBM

#ifdef DSRT_DEVICE_LIBM
}  // namespace devlibm
#endif
}  // namespace dsrt
