// render_wavefront.hip -- the sampling loop as a workgroup-local WAVEFRONT: lanes are workers, not pixel owners.
//
// Same arithmetic per pixel as render_kernel.hip (both run path_machine.h's state machine and the same traversal
// steps, so both produce the reference's bytes); what changes is who executes it and when.
//
// Problem with one-lane-one-pixel (measured, profiles/r01): bounce depths run from 1 to 50 and rays from 1 to several
// hundred node visits, so at any instant only ~1/3 of a wave's lanes are in the same loop -- the node loop ran at 36 %,
// the triangle loop at 25 % and the shading code at 33 % lane occupancy.
//
// Here each 256-lane workgroup owns P pixel SLOTS (P = 4 x its lane count) whose path state lives in a private strip of
// global memory (L2 / Infinity-Cache resident), and cycles through four phases separated by workgroup barriers:
//   A1  every slot whose closest-hit ray finished is SHADED            (all lanes in the shading code)
//   A2  every slot whose shadow ray finished takes its sun term        (all lanes in that short block)
//   A3  every slot that needs a new sample / pixel generates one       (background pixels burn several samples here)
//   T   every ray emitted by A1-A3 sits in one list; lanes pull rays from it until it is empty, each lane walking one
//       ray at a time with the while-while traversal and taking the next ray the moment its own ends.
// Within a phase all lanes run the same code, and in T a lane is idle only when the list has run dry.  The lists are
// 16-bit slot numbers in LDS, appended with LDS atomics; nothing is ever exchanged between workgroups.
//
// A pixel's operations still happen strictly in the reference's order (one ray of a pixel is in flight at a time; its
// LCG stream is advanced only by its own state machine), so the output is bit-identical -- tests/test_gpu_parity.py
// runs every case through this kernel too.
#include "path_machine.h"

namespace dsrt {

constexpr int kWfThreads = 256;
constexpr int kWfWaves = kWfThreads / 64;
constexpr int kWfSlots = 1024;                 // pixel slots per workgroup

enum SlotField : int {
    SF_STATE, SF_PIX, SF_OUT, SF_SAMPLE, SF_DEPTH, SF_RNG,
    SF_ACC, SF_THR = SF_ACC + 3, SF_L = SF_THR + 3, SF_RO = SF_L + 3, SF_RD = SF_RO + 3, SF_RINV = SF_RD + 3,
    SF_HIT_T = SF_RINV + 3, SF_HIT_SLOT, SF_HIT_U, SF_HIT_V,
    SF_PC, SF_PT = SF_PC + 3, SF_PO = SF_PT + 3, SF_PD = SF_PO + 3, SF_PEND_END = SF_PD + 3,
    kSlotFields
};

__host__ __device__ constexpr size_t wavefront_state_words_per_block() { return (size_t)kSlotFields * kWfSlots; }

struct SlotIO {
    uint32_t* base;
    int slot;
    __device__ __forceinline__ uint32_t u(int f) const { return base[(size_t)f * kWfSlots + slot]; }
    __device__ __forceinline__ float f(int fl) const { return __uint_as_float(u(fl)); }
    __device__ __forceinline__ F3 v(int fl) const { return mk(f(fl), f(fl + 1), f(fl + 2)); }
    __device__ __forceinline__ void su(int fl, uint32_t x) const { base[(size_t)fl * kWfSlots + slot] = x; }
    __device__ __forceinline__ void sf(int fl, float x) const { su(fl, __float_as_uint(x)); }
    __device__ __forceinline__ void sv(int fl, F3 x) const { sf(fl, x.x); sf(fl + 1, x.y); sf(fl + 2, x.z); }
};

__device__ __forceinline__ void load_lane(Lane& ln, const SlotIO& io) {
    ln.state = (int)io.u(SF_STATE);
    const uint32_t pix = io.u(SF_PIX);
    ln.px = (int)(pix & 0xFFFFu); ln.ky = (int)(pix >> 16);
    ln.out_index = io.u(SF_OUT); ln.sample = (int)io.u(SF_SAMPLE); ln.depth = (int)io.u(SF_DEPTH); ln.rng = io.u(SF_RNG);
    ln.accum = io.v(SF_ACC); ln.thr = io.v(SF_THR); ln.L = io.v(SF_L);
    ln.ro = io.v(SF_RO); ln.rd = io.v(SF_RD); ln.rinv = io.v(SF_RINV);
    ln.closest = io.f(SF_HIT_T); ln.hit_slot = (int)io.u(SF_HIT_SLOT); ln.hit_u = io.f(SF_HIT_U); ln.hit_v = io.f(SF_HIT_V);
    ln.pend_contrib = io.v(SF_PC); ln.pend_thr = io.v(SF_PT); ln.pend_o = io.v(SF_PO); ln.pend_d = io.v(SF_PD);
    ln.pend_end = io.u(SF_PEND_END) != 0;
}

__device__ __forceinline__ void store_lane(const Lane& ln, const SlotIO& io) {
    io.su(SF_STATE, (uint32_t)ln.state);
    io.su(SF_PIX, (uint32_t)ln.px | ((uint32_t)ln.ky << 16));
    io.su(SF_OUT, ln.out_index); io.su(SF_SAMPLE, (uint32_t)ln.sample); io.su(SF_DEPTH, (uint32_t)ln.depth); io.su(SF_RNG, ln.rng);
    io.sv(SF_ACC, ln.accum); io.sv(SF_THR, ln.thr); io.sv(SF_L, ln.L);
    io.sv(SF_RO, ln.ro); io.sv(SF_RD, ln.rd); io.sv(SF_RINV, ln.rinv);
    io.sf(SF_HIT_T, ln.closest); io.su(SF_HIT_SLOT, (uint32_t)ln.hit_slot); io.sf(SF_HIT_U, ln.hit_u); io.sf(SF_HIT_V, ln.hit_v);
    io.sv(SF_PC, ln.pend_contrib); io.sv(SF_PT, ln.pend_thr); io.sv(SF_PO, ln.pend_o); io.sv(SF_PD, ln.pend_d);
    io.su(SF_PEND_END, ln.pend_end ? 1u : 0u);
}

template <int K, bool COUNT, bool CHECKED, bool ANYHIT>
__global__ void __launch_bounds__(kWfThreads) dsrt_wavefront_kernel(const RenderArgs args) {
    const DeviceScene& S = args.scene;
    __shared__ uint2 lds_stack[kWfWaves][K][64];
    __shared__ uint16_t q_shade[kWfSlots], q_shadow[kWfSlots], q_ray[kWfSlots], q_gen[2][kWfSlots];
    __shared__ uint32_t n_shade, n_shadow, n_ray, n_gen[2], ray_next;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* const strip = args.wf_state + (size_t)blockIdx.x * wavefront_state_words_per_block();

    uint32_t c[kNumCounters];
#pragma unroll
    for (int i = 0; i < kNumCounters; ++i) c[i] = 0;
    uint32_t flags = 0;

    // every slot starts out needing a pixel
    for (int s = tid; s < kWfSlots; s += kWfThreads) {
        Lane fresh;
        store_lane(fresh, SlotIO{strip, s});
        q_gen[0][s] = (uint16_t)s;
    }
    if (tid == 0) { n_shade = 0; n_shadow = 0; n_ray = 0; n_gen[0] = kWfSlots; n_gen[1] = 0; ray_next = 0; }
    int g = 0;                                  // which q_gen is being consumed this round
    __syncthreads();

    // Run one slot's state machine until it has a ray to walk, or (stop_at_gen) until it needs a new sample, or the
    // budget runs out; then file the slot in the list of whoever serves it next.
    auto serve = [&](int slot, int budget, bool stop_at_gen, uint16_t* gen_out, uint32_t* gen_count) {
        Lane ln;
        const SlotIO io{strip, slot};
        load_lane(ln, io);
        for (int b = 0; b < budget && ln.state < ST_TRAV_CLOSEST; ++b) {
            if (stop_at_gen && ln.state <= ST_GEN) break;
            if (COUNT) { c[C_ADV_ACTIVE]++; }
            advance_step<COUNT, CHECKED, ANYHIT>(ln, args, c, flags);
        }
        store_lane(ln, io);
        if (ln.state == ST_TRAV_CLOSEST || ln.state == ST_TRAV_SHADOW) q_ray[atomicAdd(&n_ray, 1u)] = (uint16_t)slot;
        else if (ln.state != ST_DONE) gen_out[atomicAdd(gen_count, 1u)] = (uint16_t)slot;
    };

    for (;;) {
        // ---------------- A1 + A2: shade finished closest-hit rays, settle finished shadow rays; then, after a barrier,
        //                  A3: new samples / new pixels (and whatever A1/A2 left unfinished).  One code instance. ----------------
#pragma nounroll
        for (int ph = 0; ph < 2; ++ph) {
            const uint32_t n0 = ph == 0 ? n_shade : n_gen[g];
            const uint32_t n1 = ph == 0 ? n_shadow : 0u;
            const uint16_t* l0 = ph == 0 ? q_shade : q_gen[g];
            uint16_t* out = ph == 0 ? q_gen[g] : q_gen[g ^ 1];
            uint32_t* out_n = ph == 0 ? &n_gen[g] : &n_gen[g ^ 1];
            const int budget = ph == 0 ? 8 : args.advance_budget;
            for (uint32_t i = tid; i < n0 + n1; i += kWfThreads) serve(i < n0 ? l0[i] : q_shadow[i - n0], budget, ph == 0, out, out_n);
            __syncthreads();
            if (ph == 0 && tid == 0) { n_shade = 0; n_shadow = 0; }
        }
        __syncthreads();
        if (tid == 0) { n_gen[g] = 0; ray_next = 0; }
        g ^= 1;
        const uint32_t rays = n_ray;
        const uint32_t pending = n_gen[g];
        __syncthreads();
        if (rays == 0 && pending == 0) break;

        // ---------------- T: walk the ray list; a lane takes the next ray as soon as its own has ended ----------------
        if (rays) {
            bool have = false, dry = false;
            int slot = 0, rstate = 0;
            F3 ro = mk(0, 0, 0), rd = mk(0, 0, 1), rinv = mk(0, 0, 0);
            int cur = kRefNone, sp = 0, hit_slot = -1;
            float closest = kTMax, hit_u = 0.0f, hit_v = 0.0f;
            uint32_t steps = 0;
            for (;;) {
                if (!have && !dry) {
                    const uint32_t idx = atomicAdd(&ray_next, 1u);
                    if (idx < rays) {
                        slot = q_ray[idx];
                        const SlotIO io{strip, slot};
                        rstate = (int)io.u(SF_STATE);
                        ro = io.v(SF_RO); rd = io.v(SF_RD); rinv = io.v(SF_RINV);
                        cur = S.root_ref; sp = 0; hit_slot = -1; closest = kTMax; steps = 0;
                        have = true;
                    } else dry = true;
                }
                if (!__any(have)) break;

                // ---- phase I: internal nodes ----
                for (;;) {
                    const bool at_node = have && cur >= 0 && cur != kRefNone;
                    const int n_node = __popcll(__ballot(at_node));
                    const int n_leaf = __popcll(__ballot(have && cur < 0));
                    if (n_node == 0 || 4 * n_node < args.leaf_ratio4 * n_leaf) break;
                    if (COUNT) c[C_NODE_SLOTS]++;
                    if (at_node) {
                        bool finished = false;
                        if (++steps > kStepCap) { flags |= kFlagStepCap; finished = true; }
                        else if (CHECKED && (unsigned)cur >= (unsigned)S.num_pairs) { flags |= kFlagBadNodeRef; finished = true; }
                        else {
                            const float4* rec = S.pairs + (size_t)cur * 4;
                            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                            const int ref_l = __float_as_int(q3.x), ref_r = __float_as_int(q3.y);
                            if (COUNT) { c[C_NODES_ENTERED]++; c[C_INTERNAL_ENTERED]++; c[C_BOX_FETCHES] += 2; }
                            const F3 l_lo = mk(q0.x, q0.y, q0.z), l_hi = mk(q0.w, q1.x, q1.y);
                            const F3 r_lo = mk(q1.z, q1.w, q2.x), r_hi = mk(q2.y, q2.z, q2.w);
                            float tl, tr;
                            const bool hl = slab(l_lo, l_hi, ro, rinv, closest, tl);
                            const bool hr = slab(r_lo, r_hi, ro, rinv, closest, tr);
                            const F3 cl = mk(0.5f * (l_lo.x + l_hi.x), 0.5f * (l_lo.y + l_hi.y), 0.5f * (l_lo.z + l_hi.z));
                            const F3 cr = mk(0.5f * (r_lo.x + r_hi.x), 0.5f * (r_lo.y + r_hi.y), 0.5f * (r_lo.z + r_hi.z));
                            const float dl = dot(cl - ro, rd), dr = dot(cr - ro, rd);
                            const bool left_near = dl < dr;
                            if (hl && hr) {
                                const int far_ref = left_near ? ref_r : ref_l;
                                const float far_t = left_near ? tr : tl;
                                cur = left_near ? ref_l : ref_r;
                                const uint2 e = make_uint2((uint32_t)far_ref, __float_as_uint(far_t));
                                if (sp < K) lds_stack[wave][sp][lane] = e;
                                else if (sp - K < args.spill_entries) { args.spill[(size_t)(sp - K) * args.spill_stride + glane] = e; if (COUNT) c[C_STACK_SPILLS]++; }
                                else { flags |= kFlagStackOverflow; finished = true; }
                                sp++;
                                if (COUNT && (uint32_t)sp > c[C_MAX_STACK]) c[C_MAX_STACK] = (uint32_t)sp;
                            } else if (hl) cur = ref_l;
                            else if (hr) cur = ref_r;
                            else {
                                cur = kRefNone;
                                for (;;) {
                                    if (sp == 0) { finished = true; break; }
                                    sp--;
                                    uint2 e = lds_stack[wave][sp < K ? sp : K - 1][lane];
                                    if (sp >= K) e = args.spill[(size_t)(sp - K) * args.spill_stride + glane];
                                    if (closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                                }
                            }
                        }
                        if (finished) cur = kRefNone;
                    }
                }

                // ---- phase L: parked leaves, triangle by triangle ----
                const bool at_leaf = have && cur < 0;
                if (__any(at_leaf)) {
                    int first = 0, count = 0;
                    bool finished = false;
                    if (at_leaf) {
                        first = leaf_payload(cur);
                        count = leaf_code(cur) + 1;
                        if (count == 8) {
                            if (CHECKED && first >= S.num_big_leaves) { flags |= kFlagBadBigLeaf; first = 0; count = 0; }
                            else { const int2 bl = S.big_leaves[first]; first = bl.x; count = bl.y; }
                        }
                        if (CHECKED && (first < 0 || first + count > S.num_tris)) { flags |= kFlagBadTriSlot; count = 0; }
                        if (++steps > kStepCap) { flags |= kFlagStepCap; count = 0; finished = true; }
                        if (COUNT) c[C_NODES_ENTERED]++;
                    }
                    for (int i = 0; __any(i < count); ++i) {
                        if (COUNT) c[C_TRI_SLOTS]++;
                        if (i < count) {
                            const int tslot = first + i;
                            const float4* tp = S.tri_isect + (size_t)tslot * 3;
                            const float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
                            if (COUNT) c[C_TRI_TESTS]++;
                            const F3 v0 = mk(a0.x, a0.y, a0.z), e1 = mk(a0.w, a1.x, a1.y), e2 = mk(a1.z, a1.w, a2.x);
                            const F3 pvec = cross(rd, e2);
                            const float det = dot(e1, pvec);
                            const float inv_det = 1.0f / det;
                            const F3 tvec = ro - v0;
                            const float u = dot(tvec, pvec) * inv_det;
                            const F3 qvec = cross(tvec, e1);
                            const float v = dot(rd, qvec) * inv_det;
                            const float t = dot(e2, qvec) * inv_det;
                            const bool accept = !(fabsf(det) < 1e-8f) && !(u < 0.0f) && !(u > 1.0f) && !(v < 0.0f) && !(u + v > 1.0f) &&
                                                !(t < kTMin) && !(t > closest);
                            if (accept) {
                                closest = t; hit_slot = tslot; hit_u = u; hit_v = v;
                                if (COUNT) c[C_HIT_UPDATES]++;
                                if (ANYHIT && rstate == ST_TRAV_SHADOW) { finished = true; count = 0; }
                            }
                        }
                    }
                    if (at_leaf) {
                        cur = kRefNone;
                        if (!finished) {
                            for (;;) {
                                if (sp == 0) break;
                                sp--;
                                uint2 e = lds_stack[wave][sp < K ? sp : K - 1][lane];
                                if (sp >= K) e = args.spill[(size_t)(sp - K) * args.spill_stride + glane];
                                if (closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                            }
                        }
                    }
                }

                // ---- retire finished rays: result to the slot, slot to the list of its next phase ----
                if (have && cur == kRefNone) {
                    const SlotIO io{strip, slot};
                    io.sf(SF_HIT_T, closest); io.su(SF_HIT_SLOT, (uint32_t)hit_slot); io.sf(SF_HIT_U, hit_u); io.sf(SF_HIT_V, hit_v);
                    if (rstate == ST_TRAV_CLOSEST) { io.su(SF_STATE, (uint32_t)ST_SHADE); q_shade[atomicAdd(&n_shade, 1u)] = (uint16_t)slot; }
                    else { io.su(SF_STATE, (uint32_t)ST_SHADOW_DONE); q_shadow[atomicAdd(&n_shadow, 1u)] = (uint16_t)slot; }
                    have = false;
                }
            }
        }
        __syncthreads();
        if (tid == 0) n_ray = 0;
        __syncthreads();
    }

    flush_counters<COUNT>(args, c);
    if (flags) atomicOr(args.flags, flags);
}

template <int K>
static hipError_t launch_wf_k(const RenderArgs& a, int blocks, bool count, bool checked, bool anyhit, hipStream_t stream) {
    const dim3 grid(blocks), block(kWfThreads);
    if (count) {
        if (anyhit) hipLaunchKernelGGL((dsrt_wavefront_kernel<K, true, true, true>), grid, block, 0, stream, a);
        else        hipLaunchKernelGGL((dsrt_wavefront_kernel<K, true, true, false>), grid, block, 0, stream, a);
    } else if (checked) {
        hipLaunchKernelGGL((dsrt_wavefront_kernel<K, false, true, true>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((dsrt_wavefront_kernel<K, false, false, true>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_wavefront(const RenderArgs& a, int lds_entries, int blocks, bool count, bool checked, bool anyhit, hipStream_t stream) {
    switch (lds_entries) {
    case 8:  return launch_wf_k<8>(a, blocks, count, checked, anyhit, stream);
    case 12: return launch_wf_k<12>(a, blocks, count, checked, anyhit, stream);
    case 16: return launch_wf_k<16>(a, blocks, count, checked, anyhit, stream);
    case 24: return launch_wf_k<24>(a, blocks, count, checked, anyhit, stream);
    default: return hipErrorInvalidValue;
    }
}

int wavefront_slots_per_block() { return kWfSlots; }
size_t wavefront_state_words() { return wavefront_state_words_per_block(); }

}  // namespace dsrt
