// microbench.hip -- calibration kernels for the roofline of the render kernel (include/dsrt.h: dsrt_microbench_gather).
//
// The render kernel's dominant memory operation is a fully divergent GATHER: every live lane of a wave reads its own 64-byte
// node record (four 16-byte loads) from a table of a few tens of MB that lives in L2 / Infinity Cache.  What bounds that is
// not HBM (the counters show 0.08 of the HBM peak) but how many 16-byte requests per cycle a CU's vector L1 (TA/TCP) accepts.
// These kernels have the render kernel's launch shape -- 256-thread workgroups, 4 waves per SIMD, grid = resident set, every
// lane gathering random aligned 64-byte records from a table of the node array's size -- and nothing else, so that their
// record rate is the ceiling of that access shape on the device the bench runs on:
//   mode 0  lane gather        each lane reads its record with 4 x global_load_dwordx4 (what the render kernel does)
//   mode 1  quad gather, DMA   the 4 lanes of a quad read ONE record per load instruction (16 B each, 64 contiguous bytes) with
//                              global_load_lds_dwordx4 into a per-wave LDS tile; each lane reads its record back with 4 x ds_read_b128
//   mode 2  quad gather, regs  same, through registers: global_load_dwordx4 + ds_write_b128 + ds_read_b128
// `dependent` chains the next record index on the loaded data (a traversal step cannot start before the previous record is in);
// `live` of the 64 lanes of every wave take part (the others are masked off, as in a half-empty traversal loop);
// `pad` v_fma per loaded record (rounded up to 16) stand in for the slab arithmetic between two node fetches.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dsrt.h"
#include "../host/host_internal.hpp"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kTileStride = 64 + 1;           // in float4: one 1-KiB DMA piece per tile; 16 bytes of padding rotate the banks between the four tiles

template <int MODE, bool DEP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
gather_kernel(const float4* __restrict__ table, uint32_t n_rec, int iters, int live, int pad, float* __restrict__ sink) {
    __shared__ float4 tiles[4][4 * kTileStride];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = (mix32((glane >> 6) * 64u + ((lane * 37u + 11u) & 63u)) & 63u) < (uint32_t)live || live >= 64;
    uint32_t state = mix32(glane * 2654435761u + 12345u);
    float acc = 0.0f;
    float4* const my_tiles = tiles[wave];
    for (int it = 0; it < iters; ++it) {
        const uint32_t idx = __umulhi(state, n_rec);                      // 0 .. n_rec-1
        float4 q0, q1, q2, q3;
        if (MODE == 0) {
            if (active) {
                const float4* rec = table + (size_t)idx * 4;
                q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3];
            } else { q0 = q1 = q2 = q3 = make_float4(0, 0, 0, 0); }
        } else {
            // quad-cooperative: load instruction j fetches the records of lanes 4q + j; lane l supplies bytes [16 (l & 3), +16)
            float4 part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // quad broadcast of lane 4q + j's index and live bit (DPP quad_perm [j,j,j,j]: a plain VALU move, no LDS)
                const int mine_packed = (int)(idx | (active ? 0x80000000u : 0u));
                const uint32_t packed = (uint32_t)(j == 0 ? __builtin_amdgcn_mov_dpp(mine_packed, 0x00, 0xF, 0xF, true)
                                                 : j == 1 ? __builtin_amdgcn_mov_dpp(mine_packed, 0x55, 0xF, 0xF, true)
                                                 : j == 2 ? __builtin_amdgcn_mov_dpp(mine_packed, 0xAA, 0xF, 0xF, true)
                                                          : __builtin_amdgcn_mov_dpp(mine_packed, 0xFF, 0xF, 0xF, true));
                const uint32_t oidx = packed & 0x7FFFFFFFu;
                const bool oact = (packed >> 31) != 0u;
                const float4* src = table + (size_t)oidx * 4 + (lane & 3u);
                if (MODE == 1) {
                    if (oact) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(my_tiles + j * kTileStride), 16, 0, 0);
                } else {
                    part[j] = oact ? *src : make_float4(0, 0, 0, 0);
                }
            }
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) my_tiles[j * kTileStride + lane] = part[j];
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (active) {
                // four ds_read_b128 and their wait in one statement (hipcc would re-split the reads around the arithmetic below)
                const uint32_t mine = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(my_tiles + (lane & 3u) * kTileStride + (lane >> 2) * 4u);
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v r0, r1, r2, r3;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(mine) : "memory");
                q0 = make_float4(r0.x, r0.y, r0.z, r0.w); q1 = make_float4(r1.x, r1.y, r1.z, r1.w);
                q2 = make_float4(r2.x, r2.y, r2.z, r2.w); q3 = make_float4(r3.x, r3.y, r3.z, r3.w);
            } else { q0 = q1 = q2 = q3 = make_float4(0, 0, 0, 0); }
            __builtin_amdgcn_wave_barrier();
        }
        float v = (((q0.x + q1.y) + (q2.z + q3.w)) + ((q0.w + q1.z) + (q2.y + q3.x))) + (((q0.y + q1.x) + (q2.w + q3.z)) + ((q0.z + q1.w) + (q2.x + q3.y)));
        {   // `pad` v_fma per record in four interleaved chains, 16 per trip of the loop (its scalar overhead is amortised 16 times)
            float a0 = v, a1 = q0.y, a2 = q1.x, a3 = q2.z;
            for (int p = 0; p < pad; p += 16) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0 = __builtin_fmaf(a0, 1.0000001f, a1); a1 = __builtin_fmaf(a1, 0.9999999f, a2);
                    a2 = __builtin_fmaf(a2, 1.0000001f, a3); a3 = __builtin_fmaf(a3, 0.9999999f, a0);
                }
            }
            v = (a0 + a1) + (a2 + a3);
        }
        acc += v;
        state = DEP ? mix32(state ^ __float_as_uint(v)) : mix32(state + 0x9E3779B9u);
    }
    if (acc == 123.456f) sink[glane] = acc;           // keeps the loads alive; never true for the table's contents
}

// ---------------------------------------------------------------------------------------------------------------
// The vector-ALU ISSUE cost of every instruction kind the render kernel is made of (dsrt_microbench_valu).  The render kernel is
// bound by VALU issue, so the ceiling it is priced against has to be measured, not assumed: `waves_per_simd` workgroups per CU (one
// wave per SIMD each; enough dynamic LDS per workgroup that no more fit) each run `iters` x 32 instructions from EIGHT INDEPENDENT
// register streams (no instruction reads the result of any of the seven before it), written as inline assembly so that the instruction
// counted is the instruction issued.  pattern 0: 32 x the instruction; pattern 1: 16 x (the instruction, then a v_add_f32 on another
// stream) -- some kinds cost far more back to back than next to something else; pattern 2: 8 x (the instruction twice, v_add_f32 twice);
// pattern 3: 16 x the instruction, then 16 x v_add_f32 (does the order INSIDE a wave matter when eight waves share the SIMD?); pattern 4:
// alternating with v_pk_mul_f32 (only for kinds on the scalar streams); pattern 5: 16 x the instruction, then 16 x v_pk_mul_f32.
// Every wave stamps its loop with the shader-clock counter (s_memtime) and the 100 MHz wall clock (s_memrealtime); the host reports
// instructions per second and the frequency the shader clock ran at, i.e. cycles per wave-instruction per SIMD.  The PMC route
// (SQ_INSTS_VALU over GRBM_GUI_ACTIVE) is tools/valu_pmc.sh.  Kinds: kValuKindNames below.
// `lane_mask`: the lanes of every wave that execute the loop (the rest branch around it): does a half-empty wave issue faster?
// ---------------------------------------------------------------------------------------------------------------
typedef float v2f_mb __attribute__((ext_vector_type(2)));
constexpr int kValuKinds = 32;
const char* const kValuKindNames[kValuKinds] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cndmask_b32_e32(vcc)", "v_max3_f32", "v_add_f32", "v_mul_f32", "v_cndmask_b32_e64(sgpr pair)", "v_cndmask_b32_e64(vcc)", "v_cmp_lt_f32_e32(vcc)", "v_min_f32", "v_mov_b32", "v_pk_add_f32", "v_rcp_f32", "v_cmp_lt_f32_e64(sgpr pair)", "v_and_b32", "v_bfi_b32", "v_max_f32", "v_med3_f32", "v_sub_f32", "v_add_u32", "v_lshlrev_b32", "v_mul_lo_u32", "v_fmac_f32", "v_xor_b32", "v_cvt_f32_u32", "v_sqrt_f32", "v_min3_f32", "v_pk_mov_b32", "v_mov_b32_dpp(quad_perm)", "v_lshl_add_u32", "v_or_b32"};

#define DSRT_A0 "%0"
#define DSRT_A1 "%1"
#define DSRT_A2 "%2"
#define DSRT_A3 "%3"
#define DSRT_A4 "%4"
#define DSRT_A5 "%5"
#define DSRT_A6 "%6"
#define DSRT_A7 "%7"
#define DSRT_P0 "%8"
#define DSRT_P1 "%9"
#define DSRT_P2 "%10"
#define DSRT_P3 "%11"
#define DSRT_P4 "%12"
#define DSRT_P5 "%13"
#define DSRT_P6 "%14"
#define DSRT_P7 "%15"
#define A(i) DSRT_A##i
#define P(i) DSRT_P##i
#define DSRT_ADD(i) "v_add_f32 " A(i) ", " A(i) ", %17\n\t"
#define DSRT_R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define DSRT_PAT0(OP) DSRT_R8(OP) DSRT_R8(OP) DSRT_R8(OP) DSRT_R8(OP)
#define DSRT_PAT1(OP) OP(0) DSRT_ADD(4) OP(1) DSRT_ADD(5) OP(2) DSRT_ADD(6) OP(3) DSRT_ADD(7) OP(0) DSRT_ADD(4) OP(1) DSRT_ADD(5) OP(2) DSRT_ADD(6) OP(3) DSRT_ADD(7) \
                      OP(0) DSRT_ADD(4) OP(1) DSRT_ADD(5) OP(2) DSRT_ADD(6) OP(3) DSRT_ADD(7) OP(0) DSRT_ADD(4) OP(1) DSRT_ADD(5) OP(2) DSRT_ADD(6) OP(3) DSRT_ADD(7)
#define DSRT_PAT2(OP) OP(0) OP(1) DSRT_ADD(4) DSRT_ADD(5) OP(2) OP(3) DSRT_ADD(6) DSRT_ADD(7) OP(0) OP(1) DSRT_ADD(4) DSRT_ADD(5) OP(2) OP(3) DSRT_ADD(6) DSRT_ADD(7) \
                      OP(0) OP(1) DSRT_ADD(4) DSRT_ADD(5) OP(2) OP(3) DSRT_ADD(6) DSRT_ADD(7) OP(0) OP(1) DSRT_ADD(4) DSRT_ADD(5) OP(2) OP(3) DSRT_ADD(6) DSRT_ADD(7)
#define DSRT_VALU_OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) \
                           : "v"(m), "v"(c), "v"(pm), "v"(pc), "s"(smask) : "vcc", "s20", "s21"
#define DSRT_PAT3(OP) DSRT_R8(OP) DSRT_R8(OP) DSRT_R8(DSRT_ADD) DSRT_R8(DSRT_ADD)
#define DSRT_PKM(i) "v_pk_mul_f32 " P(i) ", " P(i) ", %18\n\t"
#define DSRT_PAT4(OP) OP(0) DSRT_PKM(4) OP(1) DSRT_PKM(5) OP(2) DSRT_PKM(6) OP(3) DSRT_PKM(7) OP(0) DSRT_PKM(4) OP(1) DSRT_PKM(5) OP(2) DSRT_PKM(6) OP(3) DSRT_PKM(7) \
                      OP(0) DSRT_PKM(4) OP(1) DSRT_PKM(5) OP(2) DSRT_PKM(6) OP(3) DSRT_PKM(7) OP(0) DSRT_PKM(4) OP(1) DSRT_PKM(5) OP(2) DSRT_PKM(6) OP(3) DSRT_PKM(7)
#define DSRT_PAT5(OP) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3) DSRT_R8(DSRT_PKM) DSRT_R8(DSRT_PKM)
#define DSRT_VALU_ASM(OP) do { if (PATTERN == 0) asm volatile(DSRT_PAT0(OP) DSRT_VALU_OPERANDS); else if (PATTERN == 1) asm volatile(DSRT_PAT1(OP) DSRT_VALU_OPERANDS); \
                               else if (PATTERN == 2) asm volatile(DSRT_PAT2(OP) DSRT_VALU_OPERANDS); else if (PATTERN == 3) asm volatile(DSRT_PAT3(OP) DSRT_VALU_OPERANDS); \
                               else if (PATTERN == 4) asm volatile(DSRT_PAT4(OP) DSRT_VALU_OPERANDS); else asm volatile(DSRT_PAT5(OP) DSRT_VALU_OPERANDS); } while (0)

template <int KIND, int PATTERN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8)))
valu_kernel(int iters, unsigned long long lane_mask, unsigned long long smask, unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const float seed = (float)(threadIdx.x & 7u) * 1e-3f;
    float a0 = seed, a1 = seed + 1.0f, a2 = seed + 2.0f, a3 = seed + 3.0f, a4 = seed + 4.0f, a5 = seed + 5.0f, a6 = seed + 6.0f, a7 = seed + 7.0f;
    v2f_mb p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    const float m = 0.99999994f, c = 1e-7f;
    const v2f_mb pm = {m, m}, pc = {c, c};
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if ((lane_mask >> lane) & 1ull) {
        r0 = __builtin_amdgcn_s_memrealtime();
        t0 = __builtin_readcyclecounter();
        asm volatile("s_mov_b64 vcc, %0" :: "s"(smask) : "vcc");       // the selects on vcc read this (nothing else in the loop writes vcc, except the compare kind)
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {
#define DSRT_OP(i) "v_fma_f32 " A(i) ", " A(i) ", %16, %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 1) {
#define DSRT_OP(i) "v_pk_fma_f32 " P(i) ", " P(i) ", %18, %19\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 2) {
#define DSRT_OP(i) "v_pk_mul_f32 " P(i) ", " P(i) ", %18\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 3) {
#define DSRT_OP(i) "v_cndmask_b32_e32 " A(i) ", " A(i) ", %16, vcc\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 4) {
#define DSRT_OP(i) "v_max3_f32 " A(i) ", " A(i) ", %16, %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 5) {
#define DSRT_OP(i) "v_add_f32 " A(i) ", " A(i) ", %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 6) {
#define DSRT_OP(i) "v_mul_f32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 7) {
#define DSRT_OP(i) "v_cndmask_b32_e64 " A(i) ", " A(i) ", %16, %20\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 8) {
#define DSRT_OP(i) "v_cndmask_b32_e64 " A(i) ", " A(i) ", %16, vcc\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 9) {
#define DSRT_OP(i) "v_cmp_lt_f32_e32 vcc, " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 10) {
#define DSRT_OP(i) "v_min_f32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 11) {
#define DSRT_OP(i) "v_mov_b32 " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 12) {
#define DSRT_OP(i) "v_pk_add_f32 " P(i) ", " P(i) ", %19\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 13) {
#define DSRT_OP(i) "v_rcp_f32 " A(i) ", " A(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 14) {
#define DSRT_OP(i) "v_cmp_lt_f32_e64 s[20:21], " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 15) {
#define DSRT_OP(i) "v_and_b32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 16) {
#define DSRT_OP(i) "v_bfi_b32 " A(i) ", %16, %17, " A(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 17) {
#define DSRT_OP(i) "v_max_f32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 18) {
#define DSRT_OP(i) "v_med3_f32 " A(i) ", " A(i) ", %16, %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 19) {
#define DSRT_OP(i) "v_sub_f32 " A(i) ", " A(i) ", %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 20) {
#define DSRT_OP(i) "v_add_u32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 21) {
#define DSRT_OP(i) "v_lshlrev_b32 " A(i) ", 1, " A(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 22) {
#define DSRT_OP(i) "v_mul_lo_u32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 23) {
#define DSRT_OP(i) "v_fmac_f32 " A(i) ", %16, %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 24) {
#define DSRT_OP(i) "v_xor_b32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 25) {
#define DSRT_OP(i) "v_cvt_f32_u32 " A(i) ", " A(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 26) {
#define DSRT_OP(i) "v_sqrt_f32 " A(i) ", " A(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 27) {
#define DSRT_OP(i) "v_min3_f32 " A(i) ", " A(i) ", %16, %17\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 28) {
#define DSRT_OP(i) "v_pk_mov_b32 " P(i) ", %18, " P(i) "\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 29) {
#define DSRT_OP(i) "v_mov_b32_dpp " A(i) ", " A(i) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 30) {
#define DSRT_OP(i) "v_lshl_add_u32 " A(i) ", " A(i) ", 1, %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
            else if (KIND == 31) {
#define DSRT_OP(i) "v_or_b32 " A(i) ", " A(i) ", %16\n\t"
                DSRT_VALU_ASM(DSRT_OP);
#undef DSRT_OP
            }
        }
        t1 = __builtin_readcyclecounter();
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const uint32_t first = (uint32_t)__builtin_ctzll(lane_mask);
    if (lane == first) { stamps[2 * (size_t)gwave] = t1 - t0; stamps[2 * (size_t)gwave + 1] = r1 - r0; }
    const float r = (((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7))) + (((p0.x + p1.y) + (p2.x + p3.y)) + ((p4.x + p5.y) + (p6.x + p7.y)));
    if (r == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
#undef A
#undef P

template <int KIND, int PATTERN>
hipError_t launch_valu_one(int blocks, size_t lds, int iters, unsigned long long lane_mask, unsigned long long* stamps, float* sink) {
    hipError_t e = hipFuncSetAttribute((const void*)valu_kernel<KIND, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((valu_kernel<KIND, PATTERN>), dim3(blocks), dim3(256), lds, nullptr, iters, lane_mask, 0x5A5A5A5AA5A5A5A5ull, stamps, sink);
    return hipGetLastError();
}

template <int KIND>
hipError_t launch_valu(int kind, int pattern, int blocks, size_t lds, int iters, unsigned long long lane_mask, unsigned long long* stamps, float* sink) {
    if (kind == KIND) {
        if (pattern == 0) return launch_valu_one<KIND, 0>(blocks, lds, iters, lane_mask, stamps, sink);
        if (pattern == 1) return launch_valu_one<KIND, 1>(blocks, lds, iters, lane_mask, stamps, sink);
        if (pattern == 2) return launch_valu_one<KIND, 2>(blocks, lds, iters, lane_mask, stamps, sink);
        if (pattern == 3) return launch_valu_one<KIND, 3>(blocks, lds, iters, lane_mask, stamps, sink);
        if (pattern == 4) return launch_valu_one<KIND, 4>(blocks, lds, iters, lane_mask, stamps, sink);
        return launch_valu_one<KIND, 5>(blocks, lds, iters, lane_mask, stamps, sink);
    }
    if constexpr (KIND + 1 < kValuKinds) return launch_valu<KIND + 1>(kind, pattern, blocks, lds, iters, lane_mask, stamps, sink);
    return hipErrorInvalidValue;
}

template <int MODE>
hipError_t launch_gather(bool dep, const float4* table, uint32_t n_rec, int iters, int live, int pad, float* sink, int blocks, hipStream_t s) {
    if (dep) hipLaunchKernelGGL((gather_kernel<MODE, true>), dim3(blocks), dim3(256), 0, s, table, n_rec, iters, live, pad, sink);
    else     hipLaunchKernelGGL((gather_kernel<MODE, false>), dim3(blocks), dim3(256), 0, s, table, n_rec, iters, live, pad, sink);
    return hipGetLastError();
}

bool ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    dsrt::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define MB_TRY(expr) do { if (!ok((expr), #expr)) { cleanup(); return DSRT_ERR_HIP; } } while (0)

}  // namespace

extern "C" int dsrt_microbench_gather(int device, int mode, int dependent, int live_lanes, int pad_valu, size_t table_bytes, int iters,
                                      float* out_ms, double* out_records) {
    if (mode < 0 || mode > 2 || live_lanes < 1 || live_lanes > 64 || pad_valu < 0 || pad_valu > 4096 || iters < 1 || iters > (1 << 20) ||
        table_bytes < 4096 || table_bytes > ((size_t)1 << 34) || !out_ms || !out_records) {
        dsrt::set_error("dsrt_microbench_gather: bad argument");
        return DSRT_ERR_INVALID;
    }
    float4* table = nullptr;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (table) (void)hipFree(table);
        if (sink) (void)hipFree(sink);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    MB_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    MB_TRY(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount * 4;              // the render kernel's resident set
    const uint32_t n_rec = (uint32_t)(table_bytes / 64);
    MB_TRY(hipMalloc((void**)&table, (size_t)n_rec * 64));
    MB_TRY(hipMalloc((void**)&sink, (size_t)blocks * 256 * sizeof(float)));
    {
        std::vector<float> host((size_t)n_rec * 16);
        uint32_t s = 1u;
        for (float& f : host) { s = s * 1664525u + 1013904223u; f = (float)(s >> 8) * (1.0f / 16777216.0f); }
        MB_TRY(hipMemcpy(table, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    auto run = [&](int n) -> hipError_t {
        if (mode == 0) return launch_gather<0>(dependent != 0, table, n_rec, n, live_lanes, pad_valu, sink, blocks, nullptr);
        if (mode == 1) return launch_gather<1>(dependent != 0, table, n_rec, n, live_lanes, pad_valu, sink, blocks, nullptr);
        return launch_gather<2>(dependent != 0, table, n_rec, n, live_lanes, pad_valu, sink, blocks, nullptr);
    };
    MB_TRY(run(iters < 64 ? iters : 64));                         // warm the caches and the clocks
    MB_TRY(hipDeviceSynchronize());
    MB_TRY(hipEventRecord(e0, nullptr));
    MB_TRY(run(iters));
    MB_TRY(hipEventRecord(e1, nullptr));
    MB_TRY(hipEventSynchronize(e1));
    MB_TRY(hipEventElapsedTime(out_ms, e0, e1));
    // live lanes are chosen per lane by a hash: count them the same way on the host
    double live_total = 0;
    for (uint32_t g = 0; g < (uint32_t)blocks * 256u; ++g) {
        const uint32_t lane = g & 63u;
        uint32_t x = (g >> 6) * 64u + ((lane * 37u + 11u) & 63u);
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        if (live_lanes >= 64 || (x & 63u) < (uint32_t)live_lanes) live_total += 1.0;
    }
    *out_records = live_total * (double)iters;
    cleanup();
    return DSRT_OK;
}

// Streaming-copy calibration: what this board's HBM delivers to a plain float4 grid-stride copy (16 B per lane per access, read + write counted), the
// denominator next to the spec's 8 TB/s.  Buffers far beyond the 256 MB Infinity Cache, so that neither side is served on the die.
typedef float nt_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) { const nt_v4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store(float4 a, float4* p) { nt_v4 v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<nt_v4*>(p)); }

// MODE 0: copy; 1: read only (every lane sums what it loads, one store per lane at the end); 2: write only; 3: copy with non-temporal loads and stores
template <int MODE>
__global__ void __launch_bounds__(256) dsrt_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const size_t first = i;
    for (; i + 3 * stride < n; i += 4 * stride) {               // four independent 16-byte accesses in flight per lane
        float4 a = acc, b = acc, c = acc, d = acc;
        if (MODE == 3) { a = nt_load(src + i); b = nt_load(src + i + stride); c = nt_load(src + i + 2 * stride); d = nt_load(src + i + 3 * stride); }
        else if (MODE != 2) { a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride]; }
        if (MODE == 1) { acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w; }
        else if (MODE == 3) { nt_store(a, dst + i); nt_store(b, dst + i + stride); nt_store(c, dst + i + 2 * stride); nt_store(d, dst + i + 3 * stride); }
        else { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
    }
    for (; i < n; i += stride) { if (MODE == 1) acc.x += src[i].x; else if (MODE == 2) dst[i] = acc; else dst[i] = src[i]; }
    if (MODE == 1) dst[first] = acc;
}

extern "C" int dsrt_microbench_copy(int device, int mode, size_t bytes, int blocks_per_cu, int reps, float* out_ms, double* out_bytes_moved) {
    if (mode < 0 || mode > 3 || bytes < ((size_t)1 << 20) || bytes > ((size_t)1 << 36) || blocks_per_cu < 1 || blocks_per_cu > 64 || reps < 1 || reps > 1000 || !out_ms || !out_bytes_moved) {
        dsrt::set_error("dsrt_microbench_copy: bad argument");
        return DSRT_ERR_INVALID;
    }
    float4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (src) (void)hipFree(src);
        if (dst) (void)hipFree(dst);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    MB_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    MB_TRY(hipGetDeviceProperties(&prop, device));
    const size_t n = bytes / sizeof(float4);
    MB_TRY(hipMalloc((void**)&src, n * sizeof(float4)));
    MB_TRY(hipMalloc((void**)&dst, n * sizeof(float4)));
    MB_TRY(hipMemset(src, 0x3c, n * sizeof(float4)));
    MB_TRY(hipMemset(dst, 0, n * sizeof(float4)));
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    const int blocks = prop.multiProcessorCount * blocks_per_cu;
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(dsrt_copy_kernel<0>, dim3(blocks), dim3(256), 0, nullptr, (const float4*)src, dst, n);
        else if (mode == 1) hipLaunchKernelGGL(dsrt_copy_kernel<1>, dim3(blocks), dim3(256), 0, nullptr, (const float4*)src, dst, n);
        else if (mode == 2) hipLaunchKernelGGL(dsrt_copy_kernel<2>, dim3(blocks), dim3(256), 0, nullptr, (const float4*)src, dst, n);
        else hipLaunchKernelGGL(dsrt_copy_kernel<3>, dim3(blocks), dim3(256), 0, nullptr, (const float4*)src, dst, n);
    };
    launch();                                                                                                         // warm-up
    MB_TRY(hipGetLastError());
    MB_TRY(hipDeviceSynchronize());
    MB_TRY(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) launch();
    MB_TRY(hipGetLastError());
    MB_TRY(hipEventRecord(e1, nullptr));
    MB_TRY(hipEventSynchronize(e1));
    MB_TRY(hipEventElapsedTime(out_ms, e0, e1));
    *out_bytes_moved = (mode == 1 || mode == 2 ? 1.0 : 2.0) * (double)(n * sizeof(float4)) * (double)reps;
    cleanup();
    return DSRT_OK;
}

extern "C" int dsrt_microbench_valu_kinds(void) { return kValuKinds; }
extern "C" const char* dsrt_microbench_valu_kind_name(int kind) { return kind >= 0 && kind < kValuKinds ? kValuKindNames[kind] : ""; }

extern "C" int dsrt_microbench_valu(int device, int kind, int pattern, int waves_per_simd, int iters, uint64_t lane_mask, float* out_ms, double* out_wave_instructions,
                                    double* out_shader_clock_GHz) {
    if (kind < 0 || kind >= kValuKinds || pattern < 0 || pattern > 5 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || iters > (1 << 24) || lane_mask == 0 || !out_ms ||
        !out_wave_instructions || !out_shader_clock_GHz) {
        dsrt::set_error("dsrt_microbench_valu: bad argument");
        return DSRT_ERR_INVALID;
    }
    unsigned long long* stamps = nullptr;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (stamps) (void)hipFree(stamps);
        if (sink) (void)hipFree(sink);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    MB_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    MB_TRY(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount * waves_per_simd;          // 256 threads = one wave per SIMD of a CU per workgroup
    // dynamic LDS per workgroup such that waves_per_simd workgroups fit a CU's 160 KB and one more does not
    const size_t lds = ((size_t)160 * 1024 / (size_t)waves_per_simd) / 1024 * 1024 - (waves_per_simd == 8 ? 0 : 1024);
    const size_t waves = (size_t)blocks * 4;
    MB_TRY(hipMalloc((void**)&stamps, 2 * waves * sizeof(unsigned long long)));
    MB_TRY(hipMalloc((void**)&sink, waves * 64 * sizeof(float)));
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    MB_TRY(launch_valu<0>(kind, pattern, blocks, lds, iters, (unsigned long long)lane_mask, stamps, sink));      // warms the clocks as well: same length as the timed run
    MB_TRY(hipDeviceSynchronize());
    MB_TRY(hipEventRecord(e0, nullptr));
    MB_TRY(launch_valu<0>(kind, pattern, blocks, lds, iters, (unsigned long long)lane_mask, stamps, sink));
    MB_TRY(hipEventRecord(e1, nullptr));
    MB_TRY(hipEventSynchronize(e1));
    MB_TRY(hipEventElapsedTime(out_ms, e0, e1));
    std::vector<unsigned long long> host(2 * waves);
    MB_TRY(hipMemcpy(host.data(), stamps, 2 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (size_t w = 0; w < waves; ++w) { cyc += (double)host[2 * w]; real += (double)host[2 * w + 1]; }
    *out_wave_instructions = (double)iters * 32.0 * (double)waves;
    *out_shader_clock_GHz = real > 0 ? cyc / real * 0.1 : 0.0;             // s_memrealtime ticks at 100 MHz
    cleanup();
    return DSRT_OK;
}
