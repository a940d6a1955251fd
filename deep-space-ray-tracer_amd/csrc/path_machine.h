// path_machine.h -- one pixel's path as a small state machine; the render kernels differ only in how they schedule it.
//
// States (what the lane needs next):
//   ST_FETCH        a pixel from the work queue                  ST_GEN          start the next sample / finish the pixel
//   ST_BOUNCE       top of ray_color's depth loop (:727-744)     ST_SHADE        closest-hit result is in (hit_slot, closest, u, v)
//   ST_SHADOW_DONE  shadow-ray result is in hit_slot             ST_TRAV_*       a ray is ready / being walked through the BVH
// advance_step() takes a lane whose state is < ST_TRAV_CLOSEST as far as it can go without walking a ray.  The state blocks are
// laid out in the order paths flow through them (SHADOW_DONE -> SHADE -> BOUNCE -> FETCH -> GEN -> ray start), each executed
// once per call by all lanes that are in that state when control reaches it, so one call normally ends with every lane either
// walking a new ray or done; the ray start (reciprocals, root-box test) is shared by all three kinds of ray.
#pragma once

#include <type_traits>

#include "device_math.h"

namespace dsrt {

enum : int {
    ST_FETCH = 0, ST_GEN = 1, ST_BOUNCE = 2, ST_SHADE = 3, ST_SHADOW_DONE = 4, ST_ENDING = 5,       // advance-phase states
    ST_TRAV_CLOSEST = 8, ST_TRAV_SHADOW = 9,                                                         // traverse-phase states
    ST_DONE = 16,                                                                                    // out of work: free to trace shadow rays for others
    kParked = 32                                                                                     // added to an advance-phase state: waiting for a helper's answer
};

// Shadow rays by idle lanes.  A lane that is out of work (ST_DONE) does not leave: when a busy lane of its wave reaches a
// Lambertian hit that needs a sun shadow ray (:800-836), the idle lane takes that ray and the busy lane goes straight on to its
// next bounce, so the two walks of a bounce overlap instead of following each other.  Nothing about the arithmetic changes: the
// shadow ray draws no random numbers; its contribution waits in the owner's LDS strip and is added to L before anything else
// touches L (the next hit's terms, the end of the sample), i.e. in the reference's order; at most one delegated ray per path
// is outstanding.  This shortens the one thing rng_mode 0 cannot parallelise -- a pixel's serial chain of samples -- whenever
// the chip is not full: far frames, the tail of a frame, one rank's share of a multi-GPU job.
//   Lane::aux  bits 0-5 lane number; bit 8 kAwait: a delegated shadow ray of this path is outstanding; bit 9 kHot: pixel of a heavy tile;
//              bits 16-22: (owner lane + 1) while this lane traces a shadow ray for `owner`;
//              bits 24-31: probe launch only: rays traced for the current pixel (saturating)
//   pend strip while a ray is delegated: [0..2] the contribution, [3..5] shadow origin, [6..8] shadow direction,
//              [12] the answer: 0 pending, 1 blocked, 2 clear;  row 13: the wave's request table (owner lane per request rank)
constexpr uint32_t kAwait = 1u << 8;
// The certified second tree (args.accel; device_api.hip: pack_scene).  A ray starts on the second tree with its distance culling relaxed by kCullRelax; when its walk
// ends, the CERTIFICATE decides whether the answer is provably the reference walk's (advance_step), and if not the same ray walks the reference tree.
//              bit 10 kOnRef: this ray is (re-)walking the reference tree: its answer is final;  bit 11 kTie: the last accept on the second tree was an equality accept
constexpr uint32_t kOnRef = 1u << 10, kTie = 1u << 11, kAudit = 1u << 12;   // (kAudit: counting build, RenderArgs::audit -- this ray's certified answer is being checked against the reference walk)
#ifndef DSRT_CULL_RELAX_DIV
#define DSRT_CULL_RELAX_DIV 1024.0f
#endif
constexpr float kCullRelax = 1.0f + 1.0f / DSRT_CULL_RELAX_DIV;
constexpr uint32_t kHot = 1u << 9;           // rng_mode 0: the pixel this lane is working on belongs to a heavy tile (its wave asks for issue priority)

// Hand-off between two LANES of one wave through LDS (request table, ray, answer word).  The lanes of a wave execute in lockstep and
// the DS unit retires a wave's operations in order, so no hardware instruction is needed; what IS needed is that the compiler
// keeps the writer's stores ahead of, and the reader's loads behind, the ballot that separates them.  A wavefront-scope fence
// says exactly that and costs no ISA.
__device__ __forceinline__ void lane_handoff_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
__device__ __forceinline__ void lane_handoff_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

constexpr uint32_t kStepCap = 1u << 22;     // no ray walks more node/leaf steps than this (guards against corrupt input)

struct Lane {
    int state = ST_FETCH;
    int px = 0, ky = 0, sample = 0, depth = 0;
    uint32_t out_index = 0, rng = 0;             // rng: LCG state (rng_mode 0) or draws taken in the current sample (rng_mode 1)
    int sample_end = 0;                          // rng_mode 1: this work item covers samples [.., sample_end) of the pixel
    uint32_t chunk = 0;                          // rng_mode 1: which slice of the pixel's samples
    uint32_t aux = 0;                            // lane number inside the wave + helper / await bits (see above)
    uint32_t frame = 0;                          // batch launches (dsrt_render_batch): the table entry (camera, sun) this lane's pixel belongs to
    uint32_t t0 = 0;                             // counting build: wall clock (100 MHz, low 32 bits) when this lane fetched its current work item
    F3 accum = {0, 0, 0}, thr = {1, 1, 1}, L = {0, 0, 0};
    F3 ro = {0, 0, 0}, rd = {0, 0, 1}, rinv = {0, 0, 0};
    int cur = kRefNone, sp = 0, hit_slot = -1;
    float closest = kTMax, hit_u = 0.0f, hit_v = 0.0f;
    int audit_orig = -1; float audit_t = 0.0f, audit_u = 0.0f, audit_v = 0.0f;     // counting build, certificate audit: the second tree's answer while the reference tree is walked
    float cull = kTMax;                          // what BOXES are culled against: `closest` itself on the reference tree, closest * kCullRelax on the second tree
    float relax = 1.0f;                          // cull = closest * relax
    uint32_t steps = 0;
    // The continuation postponed while a shadow ray is in flight (sun term, next throughput, next ray, "path ends") is
    // 13 words that nothing touches during the walk: it lives in LDS, one column per lane, not in registers.
    float* pend = nullptr;                      // &strip[0][lane of block], element i at pend[i * kPendStride]
};

constexpr int kPendStride = 256;                // = threads per block of the render kernel
constexpr int kPendWords = 14;                  // 13 words of continuation + one row for the shadow-ray request table
__device__ __forceinline__ void pend_put(const Lane& ln, int i, F3 v) { ln.pend[(i + 0) * kPendStride] = v.x; ln.pend[(i + 1) * kPendStride] = v.y; ln.pend[(i + 2) * kPendStride] = v.z; }
__device__ __forceinline__ F3 pend_get(const Lane& ln, int i) { return mk(ln.pend[(i + 0) * kPendStride], ln.pend[(i + 1) * kPendStride], ln.pend[(i + 2) * kPendStride]); }

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const RenderArgs& args, uint32_t* c) {
    if (COUNT) {
#pragma unroll
        for (int i = 0; i < kNumCounters; ++i) {
            if (i >= C_T_FIRST) continue;                       // time marks: written where they happen (mark_time)
            if (i == C_MAX_STACK) atomicMax((unsigned long long*)&args.counters[i], (unsigned long long)c[i]);
            else if (c[i]) atomicAdd((unsigned long long*)&args.counters[i], (unsigned long long)c[i]);
            if (i != C_MAX_STACK) c[i] = 0;
        }
    }
}

// Counting build: the earliest (first three marks, stored inverted) or latest moment any lane passed a point of the launch.
template <bool COUNT>
__device__ __forceinline__ void mark_time(const RenderArgs& args, int which) {
    if (COUNT) {
        const unsigned long long t = wall_clock64();
        atomicMax((unsigned long long*)&args.counters[which], which == C_T_LAST ? t : ~t);
    }
}

template <int RNGMODE> struct RngOf;
template <> struct RngOf<0> { using type = uint32_t&; };
template <> struct RngOf<1> { using type = PhiloxStream; };
template <int RNGMODE>
__device__ __forceinline__ typename RngOf<RNGMODE>::type make_rng(Lane& ln, const FrameParams& P) {
    if constexpr (RNGMODE == 0) {
        return ln.rng;
    } else {
        const unsigned long long sub = (unsigned long long)(uint32_t)(ln.px + ln.ky * P.width) * (unsigned long long)(uint32_t)P.spp + (unsigned long long)(uint32_t)ln.sample;
        return PhiloxStream{P.seed32, P.seed_hi, (uint32_t)sub, (uint32_t)(sub >> 32), ln.rng};
    }
}

// LEAN: instantiation for scenes that have no spheres (hence no sphere lights), no textures and only Lambertian materials -- decided once, at upload
// (device_api.hip: pack_scene) -- which is what an OBJ mesh with plain Kd materials is, the headline mesh included.  The blocks for the other material
// classes, the sphere loops, the mixture branch and the texture fetch are then not compiled at all: exact by construction (they are the branches such a scene
// never takes) and ~20 scalar registers the advance pass no longer keeps alive.
template <bool COUNT, bool CHECKED, bool ANYHIT, int RNGMODE, bool PROBE, bool BATCH = false, bool LEAN = false>
__device__ __forceinline__ void advance_step(Lane& ln, const RenderArgs& args, uint32_t* c, uint32_t& flags) {
    const DeviceScene& S = args.scene;
    const int num_spheres = LEAN ? 0 : S.num_spheres;
    const FrameParams& P = args.frame;
    int& state = ln.state; int& px = ln.px; int& ky = ln.ky; int& sample = ln.sample; int& depth = ln.depth;
    uint32_t& out_index = ln.out_index;
    F3& accum = ln.accum; F3& thr = ln.thr; F3& L = ln.L; F3& ro = ln.ro; F3& rd = ln.rd; F3& rinv = ln.rinv;
    int& cur = ln.cur; int& sp = ln.sp; int& hit_slot = ln.hit_slot;
    float& closest = ln.closest; float& hit_u = ln.hit_u; float& hit_v = ln.hit_v;
    uint32_t& steps = ln.steps;
    const int spp = P.spp;
    const int W = P.width, H = P.height;
    // the generator the body below draws from: the lane's LCG word itself, or a Philox stream rebuilt from (pixel, sample, draws)
    typename RngOf<RNGMODE>::type rng = make_rng<RNGMODE>(ln, P);
    auto restream = [&]() {                      // rng_mode 1: point the generator at the current (pixel, sample), draw 0
        if constexpr (RNGMODE == 1) {
            const unsigned long long sub = (unsigned long long)(uint32_t)(px + ky * W) * (unsigned long long)(uint32_t)spp + (unsigned long long)(uint32_t)sample;
            rng.sub0 = (uint32_t)sub; rng.sub1 = (uint32_t)(sub >> 32); rng.n = 0;
        }
    };

    // ray_color's return and the accumulate in render_kernel: clamp the SAMPLE to [0,1] (:935), add (:999), next sample.
    auto end_sample = [&]() {
        if (ln.aux & kAwait) { state = ST_ENDING; return; }     // L is not complete until the delegated shadow ray has answered
        if constexpr (RNGMODE == 1) {
            // rng_mode 1 sums a pixel's samples as integers (units of 2^-20, round to nearest): integer addition is exact and
            // associative, so the pixel's value is the same however its samples were cut into work items or handed between lanes.
            // The three words of `accum` hold the running sums as bit patterns; an item has at most 4095 samples (host), so no overflow.
            const F3 s01 = clamp01(L);
            accum.x = __uint_as_float(__float_as_uint(accum.x) + (uint32_t)(s01.x * 1048576.0f + 0.5f));
            accum.y = __uint_as_float(__float_as_uint(accum.y) + (uint32_t)(s01.y * 1048576.0f + 0.5f));
            accum.z = __uint_as_float(__float_as_uint(accum.z) + (uint32_t)(s01.z * 1048576.0f + 0.5f));
        } else {
            accum = accum + clamp01(L);
        }
        sample++;
        restream();
        state = ST_GEN;
    };
    const uint32_t my_lane = ln.aux & 63u;
    auto strip_of = [&](uint32_t other_lane) -> float* { return ln.pend + ((int)other_lane - (int)my_lane); };

    // Join: a runnable lane with a delegated shadow ray has its answer (the kernel parks it otherwise): add the sun term now, before
    // this step can touch L, exactly where the reference adds it (:816-834).
    if (ln.aux & kAwait) {
        lane_handoff_acquire();
        if (ln.pend[12 * kPendStride] == 2.0f) L = L + pend_get(ln, 0);
        ln.aux &= ~kAwait;
    }
    if (state == ST_ENDING) end_sample();

    int launch = 0;         // set by a block that leaves a new ray in (ro, rd): 1 = closest-hit ray, 2 = shadow ray

    // THE CERTIFICATE (include/dsrt.h, dsrt_ctx_set_certified_tree).  A walk of the second tree has ended with a triangle T: the accepted triangle of smallest t (closest-hit
    // rays) or some accepted triangle (any-hit shadow rays).  The reference's walk of ITS tree returns the same T -- same t, u, v: Moller-Trumbore knows no tree -- if
    //   (1) T can be reached there at all            -- triangles under a zero-thickness box were left out of the second tree;
    //   (2) no other accepted triangle has EXACTLY T's t  (the reference keeps whichever it tests last, :353)                             -- kTie, kept by apply_pair;
    //   (3) T's leaf box ON THE REFERENCE TREE, in the reference's own slab arithmetic, is passed with t_entry < t_T                       -- slab() on tri_cert, below:
    //       every ancestor's box contains the leaf's and the slab arithmetic is monotone in the box bounds (float subtraction and multiplication by one fixed 1/d
    //       round monotonically), so every ancestor is entered no later and left no earlier; `closest` never falls below t_T, the minimum; hence every box on the
    //       root-to-leaf path passes whenever the reference tests it, T is tested against a closest >= t_T and accepted, and nothing accepted later has t <= t_T;
    //   (4) no direction component is zero (0 * inf in the slab arithmetic is outside that argument).
    // A shadow ray only asks whether ANY accepted triangle is reachable: (3) for the one found says so (either the reference reaches it, or it has already found another).
    // A ray that found nothing found nothing: the second tree's walk is conservative (boxes widened at upload, distance culling relaxed by kCullRelax).
    // A ray that fails any condition walks the reference tree -- the same ray, from the root, with the reference's own culling -- and that answer is final.
    if (args.accel) {
        // CERTIFICATE AUDIT (counting build, DsrtRenderDesc.collect_counters = 3): every answer of the second tree -- certified hits and misses alike -- is ALSO walked on the
        // reference tree, the two answers are compared (triangle, and the bit patterns of t, u, v; for an any-hit shadow ray: blocked or not), differences are counted
        // (DsrtStats.certificate_audit_mismatches: must be 0) and the reference walk's answer is the one used.  What the certificate claims, checked ray by ray.
        const bool ended = state == ST_SHADE || state == ST_SHADOW_DONE;
        if (COUNT && ended && (ln.aux & kAudit) && (ln.aux & kOnRef)) {          // the reference walk of an audited ray has ended: compare
            ln.aux &= ~kAudit;
            bool same;
            if (ANYHIT && state == ST_SHADOW_DONE) same = (hit_slot >= 0) == (ln.audit_orig >= 0);
            else if (hit_slot < 0) same = ln.audit_orig < 0;
            else same = __float_as_int(S.tri_shade[(size_t)hit_slot * 3 + 2].w) == ln.audit_orig && __float_as_uint(closest) == __float_as_uint(ln.audit_t) &&
                        __float_as_uint(hit_u) == __float_as_uint(ln.audit_u) && __float_as_uint(hit_v) == __float_as_uint(ln.audit_v);
            if (!same) c[C_AUDIT_MISMATCHES]++;
        } else if (ended && !(ln.aux & kOnRef)) {
            bool certified = true;
            if (hit_slot >= 0) {
                const float4* cb = S.tri_cert + (size_t)hit_slot * 2;
                const float4 c0 = cb[0], c1 = cb[1];
                float t_entry;
                certified = slab(mk(c0.x, c0.y, c0.z), mk(c0.w, c1.x, c1.y), ro, rinv, closest, t_entry) && !(ln.aux & kTie) &&
                            rd.x != 0.0f && rd.y != 0.0f && rd.z != 0.0f;
            }
            const bool audit = COUNT && args.audit != 0 && certified;
            if (!certified || audit) {
                if (COUNT && !certified) c[C_CERT_FALLBACKS]++;
                if (COUNT && audit) {
                    c[C_AUDITED]++;
                    ln.audit_orig = hit_slot >= 0 ? __float_as_int(S.tri_shade[(size_t)hit_slot * 3 + 2].w) : -1;
                    ln.audit_t = closest; ln.audit_u = hit_u; ln.audit_v = hit_v;
                    ln.aux |= kAudit;
                }
                ln.aux = (ln.aux | kOnRef) & ~kTie;
                closest = kTMax; ln.cull = kTMax; ln.relax = 1.0f;
                hit_slot = -1;
                sp = 0;
                steps = 0;
                float t_entry;
                // the head of bvh_hit_closest :394-410 on the reference tree; a miss of its root box leaves the ray in ST_SHADE / ST_SHADOW_DONE without a hit
                if (S.root_ref != kRefNone && slab(ld3(S.root_lo), ld3(S.root_hi), ro, rinv, closest, t_entry)) { cur = S.root_ref; state += ST_TRAV_CLOSEST - ST_SHADE; }
                else if (COUNT && (ln.aux & kAudit)) { ln.aux &= ~kAudit; if (ln.audit_orig >= 0) c[C_AUDIT_MISMATCHES]++; }      // (the reference misses its root box: its answer is "no hit", now)
            }
        }
    }
    if (state == ST_SHADOW_DONE) {
        // blocked = scene_hit(shadow_ray) :816: BVH result, then the spheres
        bool blocked = hit_slot >= 0;
        if (!blocked || !ANYHIT) {
            for (int i = 0; i < num_spheres; ++i) {
                if (COUNT) c[C_SPHERE_TESTS]++;
                float t_hit; F3 n_hit;
                if (hit_sphere(S.spheres[i], ro, rd, closest, t_hit, n_hit)) { blocked = true; closest = t_hit; }
            }
        } else if (COUNT) {
            c[C_SPHERE_TESTS] += (uint32_t)num_spheres;
        }
        const uint32_t owner_plus1 = (ln.aux >> 16) & 0x7Fu;
        if (owner_plus1) {                                    // traced for another lane: hand the answer over, be free again
            strip_of(owner_plus1 - 1u)[12 * kPendStride] = blocked ? 1.0f : 2.0f;
            lane_handoff_release();                            // the answer word is read by ANOTHER lane of this wave (poll in render_body)
            ln.aux &= 0xFF00FFFFu;
            state = ST_DONE;
        } else {
            if (!blocked) L = L + pend_get(ln, 0);
            if (ln.pend[12 * kPendStride] != 0.0f) end_sample();
            else { thr = pend_get(ln, 3); ro = pend_get(ln, 6); rd = pend_get(ln, 9); depth++; state = ST_BOUNCE; }
        }
    }
    // idle lanes of this wave, counted where all lanes that entered this step are together again
    const unsigned long long free_mask = args.helpers ? wave_ballot(state == ST_DONE) : 0ull;
    const uint32_t n_free = (uint32_t)__popcll(free_mask);
    bool delegated = false;
    if (state == ST_SHADE) {
        // ---- finish scene_hit :516-551: triangle record from (slot, t, u, v), then the spheres ----
        bool hit_any = false;
        F3 hp = mk(0, 0, 0), hn = mk(0, 0, 0);
        int mat_id = 0, tex_id = -1;
        bool front = true;
        if (hit_slot >= 0) {
            const float4* sh = S.tri_shade + (size_t)hit_slot * 3;
            const float4 a0 = sh[0], a1 = sh[1], a2 = sh[2];
            const float t = closest;
            hp = mk(ro.x + t * rd.x, ro.y + t * rd.y, ro.z + t * rd.z);
            const float wgt = 1.0f - hit_u - hit_v;                                          // :359-369
            F3 n = ((mk(a0.x, a0.y, a0.z) * wgt) + (mk(a0.w, a1.x, a1.y) * hit_u)) + (mk(a1.z, a1.w, a2.x) * hit_v);
            n = normalize(n);
            front = dot(rd, n) < 0.0f;
            hn = front ? n : (n * -1.0f);
            mat_id = __float_as_int(a2.y);
            tex_id = __float_as_int(a2.z);
            hit_any = true;
        }
        for (int i = 0; i < num_spheres; ++i) {
            if (COUNT) c[C_SPHERE_TESTS]++;
            const GPUSphere sph = S.spheres[i];
            float t_hit; F3 n_hit;
            if (hit_sphere(sph, ro, rd, closest, t_hit, n_hit)) {
                hit_any = true;
                closest = t_hit;
                hp = mk(ro.x + t_hit * rd.x, ro.y + t_hit * rd.y, ro.z + t_hit * rd.z);
                front = dot(rd, n_hit) < 0.0f;
                hn = front ? n_hit : (n_hit * -1.0f);
                mat_id = sph.material_id;
                tex_id = -1;
            }
        }
        if (!hit_any) {
            end_sample();                                                                     // :744-747
        } else {
            if (COUNT) { c[C_SHADED_HITS]++; if (depth == 0) c[C_PRIMARY_HITS]++; }
            if (CHECKED && (unsigned)mat_id >= (unsigned)S.num_materials) { flags |= kFlagBadMaterial; mat_id = 0; }
            const float4* mp = S.materials + (size_t)mat_id * 3;
            const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
            const int mtype = LEAN ? (int)MAT_LAMBERTIAN : __float_as_int(m0.x);
            if (mtype == MAT_DIFFUSE_LIGHT) {                                                 // :754-758
                L = L + (thr * mk(m1.w, m2.x, m2.y));
                end_sample();
            } else {
                F3 albedo = mk(m1.x, m1.y, m1.z);                                             // :763-774
                if (!LEAN && tex_id >= 0 && S.tri_uv) {
                    const float4* uvp = S.tri_uv + (size_t)hit_slot * 2;
                    const float4 u0 = uvp[0], u1 = uvp[1];
                    const float wgt = 1.0f - hit_u - hit_v;
                    const float u_tex = wgt * u0.x + hit_u * u0.z + hit_v * u1.x;
                    const float v_tex = wgt * u0.y + hit_u * u0.w + hit_v * u1.y;
                    albedo = albedo * tex2d(S, tex_id, u_tex, v_tex, c[C_TEX_FETCHES]);
                }
                if (mtype == MAT_DIELECTRIC) {                                               // scatter_dielectric :621-661
                    const F3 dir = scatter_dielectric(rd, hn, front, m2.w, rng);
                    ro = hp; rd = dir;                      // attenuation is (1,1,1): throughput unchanged
                    depth++;
                    state = ST_BOUNCE;
                } else if (mtype == MAT_METAL) {                                             // scatter_metal :603-619
                    F3 dir;
                    if (scatter_metal(rd, hn, m2.z, rng, dir)) {
                        thr = thr * albedo;
                        ro = hp; rd = dir;
                        depth++;
                        state = ST_BOUNCE;
                    } else {
                        end_sample();
                    }
                } else {
                    // ---- Lambertian: sun next-event estimation :800-836 ----
                    bool need_shadow = false;
                    F3 sh_o = mk(0, 0, 0), sh_d = mk(0, 0, 0);
                    if (P.sun_enabled) {
                        // (Forming Ldir once per frame on the host -- same float operations -- was tried in round 3: 30 instructions fewer per shaded hit and the
                        //  frame 1 % SLOWER, 1056 -> 1067 ms in an interleaved A/B of the two builds: the register allocation around this block moved.  Left here.)
                        const float* sun = BATCH ? args.batch[ln.frame].sun_dir : P.sun_dir;
                        const F3 Ldir = normalize(mk(-sun[0], -sun[1], -sun[2]));
                        const float cos_t = fmaxf(0.0f, dot(hn, Ldir));
                        if (cos_t > 0.0f) {
                            sh_o = hp + (hn * 1e-3f);
                            sh_d = Ldir;
                            const float pdf_brdf = cos_t / kPi;
                            const float pdf_mix = 0.5f * 1.0f + 0.5f * pdf_brdf;
                            const float weight = (cos_t / kPi) / pdf_mix;
                            pend_put(ln, 0, thr * (albedo * (ld3(P.sun_radiance) * weight)));
                            need_shadow = true;
                        }
                    }
                    // ---- next direction.  The shadow ray draws no random numbers, so sampling the bounce
                    //      before tracing it leaves the LCG stream exactly as the reference's order does. ----
                    bool end_after = false;
                    F3 ndir = mk(0, 0, 1), nthr = thr;
                    if (LEAN || S.num_lights == 0) {                                         // :852-866
                        float pdf;
                        ndir = sample_cosine_hemisphere(hn, rng, pdf);
                        if (pdf <= 0.0f) end_after = true;
                        else {
                            const float cos_t = fmaxf(0.0f, dot(ndir, hn));
                            const float spdf = cos_t / kPi;
                            nthr = thr * (albedo * (spdf / pdf));
                        }
                    } else {                                                                 // :871-932
                        float pdf_val = 0.0f;
                        const float choose = rand01(rng);
                        if (choose < 0.5f) {
                            int k = (int)(rand01(rng) * (float)S.num_lights);
                            if (k >= S.num_lights) k = S.num_lights - 1;
                            int found = 0, light_idx = 0;
                            for (int i = 0; i < S.num_spheres; ++i) {
                                const float4* lm = S.materials + (size_t)S.spheres[i].material_id * 3;
                                const float4 l0 = lm[0], l1 = lm[1], l2 = lm[2];
                                if (__float_as_int(l0.x) == MAT_DIFFUSE_LIGHT && (l1.w > 0 || l2.x > 0 || l2.y > 0)) {
                                    if (found == k) { light_idx = i; break; }
                                    found++;
                                }
                            }
                            float pdf_lc = 0.0f;
                            sample_sphere_light(S.spheres[light_idx], hp, rng, ndir, pdf_lc);
                            if (pdf_lc <= 0.0f) end_after = true;
                            else {
                                const float cos_t = fmaxf(0.0f, dot(ndir, hn));
                                if (cos_t <= 0.0f) end_after = true;
                                else {
                                    const float pdf_light = pdf_lc / (float)S.num_lights;
                                    const float pdf_brdf = cos_t / kPi;
                                    pdf_val = 0.5f * pdf_light + 0.5f * pdf_brdf;
                                }
                            }
                        } else {
                            float pdf_brdf = 0.0f;
                            ndir = sample_cosine_hemisphere(hn, rng, pdf_brdf);
                            if (pdf_brdf <= 0.0f) end_after = true;
                            else pdf_val = 0.5f * pdf_brdf;
                        }
                        if (!end_after) {
                            const float cos_t = fmaxf(0.0f, dot(ndir, hn));
                            const float spdf = cos_t / kPi;
                            nthr = thr * (albedo * (spdf / pdf_val));
                        }
                    }
                    if (need_shadow && n_free > 0) {
                        // hand the shadow ray to an idle lane if there is one for this request (requests of this step, in lane order)
                        const unsigned long long req = wave_ballot(true);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(req >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)req, 0u));
                        if (rank < n_free) {
                            ln.pend[13 * kPendStride + ((int)rank - (int)my_lane)] = __uint_as_float(my_lane);
                            pend_put(ln, 3, sh_o); pend_put(ln, 6, sh_d);
                            ln.pend[12 * kPendStride] = 0.0f;
                            lane_handoff_release();            // request table, ray and cleared answer word: read by the helper lane below
                            ln.aux |= kAwait;
                            delegated = true;
                            need_shadow = false;
                        }
                    }
                    if (need_shadow) {
                        ln.pend[12 * kPendStride] = end_after ? 1.0f : 0.0f; pend_put(ln, 3, nthr); pend_put(ln, 6, hp); pend_put(ln, 9, ndir);
                        ro = sh_o; rd = sh_d;
                        launch = 2;
                    } else if (end_after) {
                        end_sample();
                    } else {
                        thr = nthr; ro = hp; rd = ndir;
                        depth++;
                        state = ST_BOUNCE;
                    }
                }
            }
        }
    }
    if (n_free > 0) {
        // the idle lanes pick up the shadow rays handed over in this step: the r-th idle lane serves the r-th request
        const unsigned long long handed = wave_ballot(delegated);
        if (handed != 0ull && state == ST_DONE) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
            if (r < (uint32_t)__popcll(handed)) {
                lane_handoff_acquire();
                const uint32_t owner = __float_as_uint(ln.pend[13 * kPendStride + ((int)r - (int)my_lane)]);
                const float* theirs = strip_of(owner);
                ro = mk(theirs[3 * kPendStride], theirs[4 * kPendStride], theirs[5 * kPendStride]);
                rd = mk(theirs[6 * kPendStride], theirs[7 * kPendStride], theirs[8 * kPendStride]);
                ln.aux |= (owner + 1u) << 16;
                launch = 2;
            }
        }
    }
    if (state == ST_BOUNCE) {
        // top of the depth loop :727-744
        bool go = depth < P.max_depth;
        if (go && depth >= 5) {
            float p = fmaxf(thr.x, fmaxf(thr.y, thr.z));
            p = fminf(p, 0.95f);
            if (rand01(rng) > p) go = false;
            else thr = thr * (1.0f / p);
        }
        if (!go) end_sample();
        else launch = 1;
    }
    if (state == ST_GEN && sample >= ln.sample_end) {       // sample_end == spp with rng_mode 0
        if constexpr (RNGMODE == 1) {
            // this work item's samples are done: its integer sums join the pixel's (dsrt_resolve_kernel tone-maps once all are in)
            unsigned long long* dst = args.accum_fixed + (size_t)out_index * 3;
            atomicAdd(dst + 0, (unsigned long long)__float_as_uint(accum.x));
            atomicAdd(dst + 1, (unsigned long long)__float_as_uint(accum.y));
            atomicAdd(dst + 2, (unsigned long long)__float_as_uint(accum.z));
        } else {
            // tone map + store :1003-1030
            float inv_spp = 1.0f / (float)spp;
            F3 col = accum * inv_spp;
            col = mk(fmaxf(col.x, 0.0f), fmaxf(col.y, 0.0f), fmaxf(col.z, 0.0f));
            col = mk(fminf(col.x, 10.0f), fminf(col.y, 10.0f), fminf(col.z, 10.0f));
            col = mk(dsrt_powf(col.x, P.inv_gamma), dsrt_powf(col.y, P.inv_gamma), dsrt_powf(col.z, P.inv_gamma));
            col = clamp01(col);
            const size_t o = (size_t)out_index * 3;
            if constexpr (!PROBE) {                             // (the probe launch measures, it has no image: in a batch its frame is not even the buffer's)
                args.out_rgb8[o + 0] = (unsigned char)(255.99f * col.x);
                args.out_rgb8[o + 1] = (unsigned char)(255.99f * col.y);
                args.out_rgb8[o + 2] = (unsigned char)(255.99f * col.z);
                if (args.out_f32) { args.out_f32[o + 0] = col.x; args.out_f32[o + 1] = col.y; args.out_f32[o + 2] = col.z; }
            }
            if (COUNT && (args.steal & 8) && args.out_f32) {
                // timing image (counting build, tune[3] bit 27; tools/chain_timeline.py): instead of the colour, the float image receives as
                // bit patterns when this pixel's chain of samples was fetched, when it ended, and which wave ran it
                args.out_f32[o + 0] = __uint_as_float(ln.t0);
                args.out_f32[o + 1] = __uint_as_float((uint32_t)wall_clock64());
                args.out_f32[o + 2] = __uint_as_float((uint32_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
            }
        }
        if (PROBE) {                                            // probe launch: what this pixel cost, added to its tile
            const uint32_t g = (uint32_t)((H - 1 - ky) / P.tile) * (uint32_t)P.tiles_x + (uint32_t)(px / P.tile);
            // a tile's place in the order is decided by its COSTLIEST pixel, not by the sum over its pixels: a tile's pixels go to 64
            // different lanes, so what the end of the frame waits for is the longest single chain, and that has to start first
            // (keyed by the sum, the near frame took 1176 instead of 1161 ms: profiles/r02/ab_probe_key_max_vs_sum.jsonl)
            atomicMax(&args.tile_work[g / (uint32_t)P.shard_count], ln.aux >> 24);
            ln.aux &= 0x00FFFFFFu;
        }
        flush_counters<COUNT>(args, c);
        state = ST_FETCH;
    }
    if (state == ST_FETCH) {
        // Two queues.  HEAVY: the pixels of tiles that see geometry, costliest tile first -- with rng_mode 1 cut into sample slices,
        // slice fastest, so a wave starts on few pixels.  LIGHT: the pixels of the remaining live tiles (they only see background, or
        // graze the root box), one work item per pixel.  `spread` lanes of every wave serve the heavy queue first, the rest the
        // light one first; a lane whose queue is empty moves to the other.  When there are more heavy items than resident lanes
        // spread is 64 and this is plain costliest-first.  When there are fewer (a far frame; one rank of a multi-GPU job), heavy
        // items are dealt out `spread` per wave over ALL resident waves instead of filling the first waves with 64 serial chains
        // each and leaving the rest of the chip with nothing to interleave: a wave's run time grows with the number of long
        // chains it holds, because lanes in different phases take turns.  Only the assignment of pixels to lanes changes.
        // The probe visits the heavy tiles only, every pixel (a tile's place is decided by its costliest pixel, so a sample of its
        // pixels misses it: all against one in four, near frame 1157 -> 1126 ms, spread 1141-1162 -> 1114-1131).
        const uint32_t tt = (uint32_t)(P.tile * P.tile);
        // Batch launch (dsrt_render_batch: many frames of one scene as ONE pool of work): a single queue runs through the frames in order,
        // each frame's heavy tiles first, then its light ones; the item number says which frame (its table entry carries that frame's
        // camera, sun, tile order and the counts its pre-pass found).  A lane that finishes a pixel of frame f simply goes on with whatever
        // comes next, so the serial chains of one frame run under the bulk of the following ones and no lane waits for a frame to end.
        uint32_t bitem = 0;
        const BatchFrame* bf = nullptr;
        if constexpr (BATCH) {
            bitem = atomicAdd(args.queue, 1u);
            uint32_t lo = 0, hi = args.batch_frames - 1u;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (bitem < args.batch[mid].item_end) hi = mid; else lo = mid + 1u; }
            bf = args.batch + lo;
            ln.frame = lo;
            if (bitem >= args.batch[args.batch_frames - 1u].item_end) bitem = 0xFFFFFFFFu;     // past the last frame: no work left
            else if (lo) bitem -= args.batch[lo - 1u].item_end;
        }
        const uint32_t n_heavy = BATCH ? bf->n_heavy : args.sched[0], n_live = BATCH ? bf->n_live : (PROBE ? args.sched[0] : args.sched[1]), spread = BATCH ? 64u : args.sched[2];
        const uint32_t per_pixel = RNGMODE == 1 ? (BATCH ? bf->slices : args.sched[3]) : 1u;          // slices per heavy pixel and their length: decided by the pre-pass
        const int chunk_len = RNGMODE == 1 ? (BATCH ? (int)bf->chunk_len : (int)args.sched[4]) : spp;
        // Pixels of the light tiles are cut into items of P.light_chunk_len samples (host: dsrt_render).  They were one item each until the
        // queue marks of the counting build -- DsrtStats.heavy_queue_empty_ms -- showed what that did: the light queue is served last, and
        // a 1000-sample item of cheap samples is a longer job than a 125-sample item of dear ones, so the biggest jobs came last.
        const int light_len = RNGMODE == 1 ? P.light_chunk_len : spp;
        const uint32_t per_pixel_light = RNGMODE == 1 ? (uint32_t)((spp + light_len - 1) / light_len) : 1u;
        const uint32_t heavy_items = n_heavy * tt * per_pixel, light_items = (n_live - n_heavy) * tt * per_pixel_light;
        bool heavy = PROBE || (ln.aux & 63u) < spread;
        uint32_t item;
        if constexpr (PROBE) {
            // The probe's work items are tiny (a few samples), so its queue word would be the bottleneck: one returning atomic on one address
            // costs about 13 ns, and half a million of them are the probe's whole run time.  64 queue words on lines of their own, picked by
            // wave number, each handing out the items congruent to its number modulo 64: the atomics of different waves no longer queue up.
            const uint32_t shard = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63u;
            item = shard + 64u * atomicAdd(args.probe_queue + shard * 16u, 1u);
        } else if constexpr (BATCH) {
            heavy = bitem < heavy_items;
            item = bitem == 0xFFFFFFFFu ? bitem : (heavy ? bitem : bitem - heavy_items);
        } else {
            if (heavy) item = atomicAdd(args.queue, 1u); else item = atomicAdd(args.queue_light, 1u);   // (uniform address per branch:
            if (item >= (heavy ? heavy_items : light_items)) {                                          //  one atomic per wave)
                mark_time<COUNT>(args, heavy ? C_T_HEAVY_EMPTY : C_T_LIGHT_EMPTY);
                heavy = !heavy;
                if (heavy) item = atomicAdd(args.queue, 1u); else item = atomicAdd(args.queue_light, 1u);
            }
        }
        const uint32_t pp = heavy ? per_pixel : per_pixel_light;
        const bool sliced = RNGMODE == 1 && pp > 1u;
        ln.chunk = 0;
        const bool none = item >= (heavy ? heavy_items : light_items);
        if constexpr (!PROBE && RNGMODE == 0) {
            // Issue priority for the long chains: render_body raises the wave's s_setprio while it holds a pixel of a heavy tile.
            ln.aux &= ~kHot;
            if (args.hot && heavy && !none) ln.aux |= kHot;
        }
        if (sliced) { ln.chunk = item % pp; item /= pp; }
        if (!heavy) item += n_heavy * tt;                     // position in tile_order x pixels per tile
        if (none) {
            state = ST_DONE;
        } else {
            const uint32_t within = item % tt;
            const uint32_t* order = BATCH ? args.batch_order + bf->order_base : P.tile_order;
            const uint32_t k = order ? order[item / tt] : item / tt;
            const uint32_t g = k * (uint32_t)P.shard_count + (uint32_t)P.shard_rank;
            const uint32_t tx = g % (uint32_t)P.tiles_x, ty = g / (uint32_t)P.tiles_x;
            const uint32_t per_row = (uint32_t)P.tile >> 3;
            const uint32_t sub = within >> 6;                                               // which 8x8 block of the tile
            const uint32_t lx = within & 7u, ly = (within >> 3) & 7u;
            const uint32_t in_x = (sub % per_row) * 8u + lx, in_y = (sub / per_row) * 8u + ly;
            const int x = (int)(tx * (uint32_t)P.tile + in_x), row = (int)(ty * (uint32_t)P.tile + in_y);
            if (x < W && row < H) {
                px = x;
                ky = H - 1 - row;                               // the kernel's y: 0 at the bottom (:984, :1027)
                out_index = P.compact_output ? (k * (uint32_t)(P.tile * P.tile) + in_y * (uint32_t)P.tile + in_x) : ((uint32_t)row * (uint32_t)W + (uint32_t)x);
                if constexpr (BATCH) out_index += bf->image_slot * args.batch_frame_pixels;     // the batch's images (or shard buffers) lie one after another
                accum = mk(0, 0, 0);
                if (COUNT) ln.t0 = (uint32_t)wall_clock64();
                if constexpr (RNGMODE == 0) {
                    rng = (uint32_t)(px + ky * W) ^ P.seed32;   // :990
                    sample = 0;
                    ln.sample_end = spp;
                } else {
                    const int len = heavy ? chunk_len : light_len;
                    sample = sliced ? (int)ln.chunk * len : 0;
                    ln.sample_end = sliced ? min(spp, sample + len) : spp;
                    restream();
                }
                state = ST_GEN;
            }
        }
    }
    if constexpr (RNGMODE == 1) {
        // Sample stealing.  With a Philox sub-sequence per (pixel, sample) any lane can compute any sample, and with integer sums it does
        // not matter which lane did.  So a lane that is out of work (the queues are empty) takes over the upper half of the remaining
        // samples of a lane of its wave that is about to start a sample and still has at least 2 * kStealMin to go: the wave then ends
        // when its WORK ends, not when its longest item does -- which is what one rank's share of a multi-GPU frame, a few items per
        // lane, is short of.  Matching as for the shadow-ray helpers: the r-th idle lane serves the r-th donor; the donor leaves
        // (pixel, output index, range) in its own LDS strip, which is free between two samples.
        constexpr int kStealMin = 4;
        const bool idle_here = state == ST_DONE && launch == 0 && ((ln.aux >> 16) & 0x7Fu) == 0u;
        const unsigned long long idle = (args.steal & 1) ? wave_ballot(idle_here) : 0ull;
        if (idle != 0ull) {
            const bool donor = state == ST_GEN && ln.sample_end - sample >= 2 * kStealMin;
            const unsigned long long donors = wave_ballot(donor);
            if (donors != 0ull) {
                const uint32_t pairs = min((uint32_t)__popcll(idle), (uint32_t)__popcll(donors));
                if (donor) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(donors >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donors, 0u));
                    if (rank < pairs) {
                        const int mid = sample + ((ln.sample_end - sample + 1) >> 1);
                        ln.pend[13 * kPendStride + ((int)rank - (int)my_lane)] = __uint_as_float(my_lane);
                        ln.pend[0 * kPendStride] = __uint_as_float((uint32_t)px | ((uint32_t)ky << 16));
                        ln.pend[1 * kPendStride] = __uint_as_float(out_index);
                        ln.pend[2 * kPendStride] = __uint_as_float((uint32_t)mid);
                        ln.pend[3 * kPendStride] = __uint_as_float((uint32_t)ln.sample_end);
                        if constexpr (BATCH) ln.pend[4 * kPendStride] = __uint_as_float(ln.frame);
                        ln.sample_end = mid;
                    }
                }
                lane_handoff_release();
                if (idle_here) {
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (r < pairs) {
                        lane_handoff_acquire();
                        const uint32_t from = __float_as_uint(ln.pend[13 * kPendStride + ((int)r - (int)my_lane)]);
                        const float* theirs = strip_of(from);
                        const uint32_t pk = __float_as_uint(theirs[0 * kPendStride]);
                        px = (int)(pk & 0xFFFFu); ky = (int)(pk >> 16);
                        out_index = __float_as_uint(theirs[1 * kPendStride]);
                        sample = (int)__float_as_uint(theirs[2 * kPendStride]);
                        ln.sample_end = (int)__float_as_uint(theirs[3 * kPendStride]);
                        if constexpr (BATCH) ln.frame = __float_as_uint(theirs[4 * kPendStride]);
                        accum = mk(0, 0, 0);                       // bit pattern 0: integer sums start at zero
                        restream();
                        state = ST_GEN;
                    }
                }
            }
        }
    }
    if (state == ST_GEN) {
        {
            float jx = ((float)sample + rand01(rng)) / (float)spp;          // :995-996
            float jy = ((float)sample + rand01(rng)) / (float)spp;
            float u = ((float)px + jx) / (float)(W - 1);                     // :952-953
            float v = ((float)ky + jy) / (float)(H - 1);
            const float* cam = BATCH ? args.batch[ln.frame].cam : P.cam;
            const F3 cam_o = ld3(cam + kCamOrigin), cam_llc = ld3(cam + kCamLlc), cam_h = ld3(cam + kCamHorizontal), cam_v = ld3(cam + kCamVertical);
            ro = cam_o;
            rd = ((cam_llc + (cam_h * u)) + (cam_v * v)) - cam_o;           // :957-961
            depth = 0;
            L = mk(0, 0, 0);
            thr = mk(1, 1, 1);
            if (COUNT) c[C_SAMPLES]++;
            launch = 1;                                 // depth 0: no roulette, max_depth >= 1 (host guarantees)
        }
    }
    // Ray start, shared by camera rays, bounces and shadow rays.  Mirrors the head of bvh_hit_closest :394-410: the root box is
    // tested first; a miss means the BVH contributes nothing and the lane goes straight on to the state that consumes the result.
    if (launch) {
        if (COUNT) c[C_RAYS]++;
        if (PROBE && (ln.aux >> 24) < 255u) ln.aux += 1u << 24;
        rinv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        closest = kTMax;
        hit_slot = -1;
        sp = 0;
        steps = 0;
        state = ST_SHADE - 1 + launch;                 // ST_SHADE for a closest-hit ray, ST_SHADOW_DONE for a shadow ray
        // which tree the ray starts on: the certified second tree when the launch uses it (wave-uniform), the reference tree otherwise
        const bool second = args.accel != 0;
        ln.cull = kTMax;
        ln.relax = second ? kCullRelax : 1.0f;
        ln.aux &= ~(kOnRef | kTie);
        const int root_ref = second ? S.accel_root_ref : S.root_ref;
        if (root_ref != kRefNone) {
            if (COUNT) c[C_BOX_FETCHES]++;
            float t_entry;
            if (slab(ld3(second ? S.accel_root_lo : S.root_lo), ld3(second ? S.accel_root_hi : S.root_hi), ro, rinv, closest, t_entry)) { cur = root_ref; state = ST_TRAV_CLOSEST - 1 + launch; }
        }
        // nothing to walk and no spheres to test: a closest-hit ray has missed the scene (:744-747)
        if (state == ST_SHADE && num_spheres == 0) end_sample();
    }
    if constexpr (RNGMODE == 1) ln.rng = rng.n;
}

}  // namespace dsrt
