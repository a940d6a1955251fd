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
//     serial stream, :990-999) and pulls the next pixel from a global queue when it finishes, so a wave never
//     idles behind its slowest pixel.  Work items are 8x8-pixel blocks in queue order: a wave starts coherent.
//   * A lane is a small state machine (fetch pixel / start sample / start bounce / traverse / shade / shadow).
//     The wave alternates between an ADVANCE phase (all non-traversing lanes step their state machine) and a
//     TRAVERSE phase (all lanes with a live ray walk the BVH together); the traverse phase is left as soon as
//     fewer lanes are walking than waiting (ballot + popcount), which keeps both phases mostly full although
//     bounce depths diverge from 1 to 50.
//   * Node visit = ONE 64-byte record with both child boxes (device_layout.h); the chosen child is not
//     re-tested on entry and a postponed child carries its slab entry distance on the stack, so a pop is a
//     single compare.  Both are bit-identical to the reference's re-tests: see `slab()` and the pop below.
//   * The traversal stack is a short stack in LDS, [entry][lane] so the 64 lanes of a wave hit 64 different
//     banks; entries beyond the LDS part spill to a per-lane strip in global memory (rare).
//   * The hit record is assembled once per ray from (slot, t, u, v) instead of on every accepted candidate,
//     and shadow rays stop at the first accepted triangle (same boolean as the reference's closest-hit search).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math: IEEE div/sqrt are part of the contract).
#include "device_layout.h"
#include "../../include/dsrt_detmath.h"

namespace dsrt {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 mk(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ F3 normalize(F3 a) {                 // f3_norm :51-56
    float L = sqrtf(dot(a, a));
    if (L <= 0.0f) return mk(0.0f, 0.0f, 0.0f);
    float inv = 1.0f / L;
    return mk(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ F3 clamp01(F3 a) {
    return mk(fminf(1.0f, fmaxf(0.0f, a.x)), fminf(1.0f, fmaxf(0.0f, a.y)), fminf(1.0f, fmaxf(0.0f, a.z)));
}

constexpr float kPi = 3.14159265358979323846f;                  // PI_F :96
constexpr float kTMin = 0.001f, kTMax = 1e9f;                   // scene_hit(ray, 0.001f, 1e9f) :744, :816

__device__ __forceinline__ float rand01(uint32_t& s) {          // :77-80
    s = s * 1664525u + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / 16777216.0f;
}

__device__ __forceinline__ F3 random_in_unit_sphere(uint32_t& rng) {   // :82-91
    for (;;) {
        float x = rand01(rng) * 2.0f - 1.0f;
        float y = rand01(rng) * 2.0f - 1.0f;
        float z = rand01(rng) * 2.0f - 1.0f;
        F3 p = mk(x, y, z);
        if (dot(p, p) >= 1.0f) continue;
        return p;
    }
}

// sample_cosine_hemisphere :121-141 with build_onb :112-118 and random_cosine_direction :99-109
__device__ __forceinline__ F3 sample_cosine_hemisphere(F3 normal, uint32_t& rng, float& pdf) {
    F3 w = normalize(normal);
    F3 a = (fabsf(w.x) > 0.9f) ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
    F3 v = normalize(cross(w, a));
    F3 u = cross(v, w);
    float r1 = rand01(rng);
    float r2 = rand01(rng);
    float lz = sqrtf(1.0f - r2);
    float phi = 2.0f * kPi * r1;
    float lx = dsrt_cosf(phi) * sqrtf(r2);
    float ly = dsrt_sinf(phi) * sqrtf(r2);
    F3 d = normalize(((u * lx) + (v * ly)) + (w * lz));
    float c = fmaxf(0.0f, dot(d, normal));
    pdf = (c > 0.0f) ? (c / kPi) : 0.0f;
    return d;
}

// sample_sphere_light_direction :145-189
__device__ __forceinline__ void sample_sphere_light(const GPUSphere& sph, F3 origin, uint32_t& rng, F3& dir, float& pdf) {
    float z = 2.0f * rand01(rng) - 1.0f;
    float phi = 2.0f * kPi * rand01(rng);
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float x = r * dsrt_cosf(phi);
    float y = r * dsrt_sinf(phi);
    F3 center = mk(sph.center.x, sph.center.y, sph.center.z);
    F3 p_light = center + (mk(x, y, z) * sph.radius);
    F3 to_light = p_light - origin;
    float dist2 = dot(to_light, to_light);
    float dist = sqrtf(dist2);
    if (dist <= 0.0f) { pdf = 0.0f; dir = mk(0.0f, 0.0f, 1.0f); return; }
    F3 wi = to_light * (1.0f / dist);
    F3 n_light = normalize(p_light - center);
    float cos_l = fmaxf(0.0f, dot(n_light, wi * -1.0f));
    if (cos_l <= 0.0f) { pdf = 0.0f; dir = wi; return; }
    float area = 4.0f * kPi * sph.radius * sph.radius;
    pdf = dist2 / (cos_l * area);
    dir = wi;
}

__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return v - (n * (2.0f * dot(v, n))); }      // :195
__device__ __forceinline__ F3 refract(F3 v, F3 n, float eta) {                                    // :199-206
    F3 uv = normalize(v);
    float c = fminf(dot(uv * -1.0f, n), 1.0f);
    F3 perp = (uv + (n * c)) * eta;
    F3 par = n * (-sqrtf(fabsf(1.0f - dot(perp, perp))));
    return perp + par;
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx) {                          // :208-212
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * dsrt_powf(1.0f - cosine, 5.0f);
}

// One box of bbox_hit :285-315 against a ray whose 1/dir is hoisted (same division, done once per ray).
// Returns hit and the entry distance tmin = max(t_min, t0x, t0y, t0z).
//   The reference walks the axes with `t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
//   if (t_max <= t_min) return false;`.  t_min only grows and t_max only shrinks, so failing after any axis implies
//   failing after the last, and the result is `!(tmax_final <= tmin_final)`.  A NaN t0/t1 (0 * inf, ray origin on a
//   slab plane with a zero direction component) loses both of the reference's comparisons and leaves the bound
//   unchanged -- which is what fmaxf/fminf do with one NaN operand; the running bounds themselves are never NaN.
//   +0/-0 differences cannot matter: the bounds are only ever compared.
__device__ __forceinline__ bool slab(F3 lo, F3 hi, F3 o, F3 inv, float t_max, float& t_entry) {
    float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    float t0x = inv.x < 0.0f ? bx : ax, t1x = inv.x < 0.0f ? ax : bx;
    float t0y = inv.y < 0.0f ? by : ay, t1y = inv.y < 0.0f ? ay : by;
    float t0z = inv.z < 0.0f ? bz : az, t1z = inv.z < 0.0f ? az : bz;
    float tmin = fmaxf(fmaxf(kTMin, t0x), fmaxf(t0y, t0z));
    float tmax = fminf(fminf(t_max, t1x), fminf(t1y, t1z));
    t_entry = tmin;
    return !(tmax <= tmin);
}

__device__ __forceinline__ bool hit_sphere(const GPUSphere& sph, F3 o, F3 d, float t_max, float& t_out, F3& n_out) {   // :478-504
    F3 center = mk(sph.center.x, sph.center.y, sph.center.z);
    F3 oc = o - center;
    float a = dot(d, d);
    float half_b = dot(oc, d);
    float c = dot(oc, oc) - sph.radius * sph.radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float root = (-half_b - sq) / a;
    if (root < kTMin || root > t_max) {
        root = (-half_b + sq) / a;
        if (root < kTMin || root > t_max) return false;
    }
    t_out = root;
    F3 p = mk(o.x + root * d.x, o.y + root * d.y, o.z + root * d.z);
    n_out = (p - center) * (1.0f / sph.radius);
    return true;
}

__device__ __forceinline__ F3 tex2d(const DeviceScene& s, int tex_id, float u, float v, uint32_t& n_fetch) {           // :232-259
    if (tex_id < 0 || tex_id >= s.num_textures || !s.tex_headers || !s.tex_pool) return mk(1.0f, 1.0f, 1.0f);
    GPUTextureHeader th = s.tex_headers[tex_id];
    u = u - floorf(u);
    v = v - floorf(v);
    int i = (int)(u * (float)(th.width - 1));
    int j = (int)((1.0f - v) * (float)(th.height - 1));
    int idx = th.offset + (j * th.width + i) * 3;
    if (idx < 0 || idx + 2 >= s.tex_pool_floats) return mk(1.0f, 1.0f, 1.0f);
    n_fetch++;
    return mk(s.tex_pool[idx + 0], s.tex_pool[idx + 1], s.tex_pool[idx + 2]);
}

enum : int {
    ST_FETCH = 0, ST_GEN = 1, ST_BOUNCE = 2, ST_SHADE = 3, ST_SHADOW_DONE = 4,                      // advance-phase states
    ST_TRAV_CLOSEST = 8, ST_TRAV_SHADOW = 9,                                                         // traverse-phase states
    ST_DONE = 16
};

constexpr int kWavesPerBlock = 4;
constexpr uint32_t kStepCap = 1u << 22;     // no ray walks more node/leaf steps than this (guards against corrupt input)

template <int K, bool COUNT, bool CHECKED, bool ANYHIT>
__global__ void __launch_bounds__(64 * kWavesPerBlock) dsrt_render_kernel(const RenderArgs args) {
    const DeviceScene& S = args.scene;
    const FrameParams& P = args.frame;
    __shared__ uint2 lds_stack[kWavesPerBlock][K][64];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;

    // ---- lane state ----
    int state = ST_FETCH;
    int px = 0, ky = 0, sample = 0, depth = 0;
    uint32_t out_index = 0, rng = 0;
    F3 accum = mk(0, 0, 0), thr = mk(1, 1, 1), L = mk(0, 0, 0);
    F3 ro = mk(0, 0, 0), rd = mk(0, 0, 1), rinv = mk(0, 0, 0);
    int cur = kRefNone, sp = 0, hit_slot = -1;
    float closest = kTMax, hit_u = 0.0f, hit_v = 0.0f;
    uint32_t steps = 0;
    // postponed continuation while the shadow ray is in flight
    F3 pend_contrib = mk(0, 0, 0), pend_thr = mk(0, 0, 0), pend_o = mk(0, 0, 0), pend_d = mk(0, 0, 0);
    bool pend_end = false;
    // counters (counting build)
    uint32_t c[kNumCounters];
#pragma unroll
    for (int i = 0; i < kNumCounters; ++i) c[i] = 0;
    uint32_t flags = 0;

    const int spp = P.spp;
    const int W = P.width, H = P.height;
    const F3 cam_o = ld3(P.cam_origin), cam_llc = ld3(P.cam_llc), cam_h = ld3(P.cam_horizontal), cam_v = ld3(P.cam_vertical);
    const F3 root_lo = ld3(S.root_lo), root_hi = ld3(S.root_hi);

    auto flush_counters = [&]() {
        if (COUNT) {
#pragma unroll
            for (int i = 0; i < kNumCounters; ++i) {
                if (i == C_MAX_STACK) atomicMax((unsigned long long*)&args.counters[i], (unsigned long long)c[i]);
                else if (c[i]) atomicAdd((unsigned long long*)&args.counters[i], (unsigned long long)c[i]);
                if (i != C_MAX_STACK) c[i] = 0;
            }
        }
    };

    // ray_color's return and the accumulate in render_kernel: clamp the SAMPLE to [0,1] (:935), add (:999), next sample.
    auto end_sample = [&]() {
        accum = accum + clamp01(L);
        sample++;
        state = ST_GEN;
    };

    // Set up the traversal of the ray in (ro, rd).  Mirrors the head of bvh_hit_closest :394-410: the root box is
    // tested first; a miss means the BVH contributes nothing and the lane goes straight to `after`.
    auto start_ray = [&](int trav_state, int after) {
        if (COUNT) c[C_RAYS]++;
        rinv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        closest = kTMax;
        hit_slot = -1;
        sp = 0;
        steps = 0;
        state = after;
        if (S.root_ref != kRefNone) {
            if (COUNT) c[C_BOX_FETCHES]++;
            float t_entry;
            if (slab(root_lo, root_hi, ro, rinv, closest, t_entry)) { cur = S.root_ref; state = trav_state; }
        }
        // nothing to walk and no spheres to test: a closest-hit ray has missed the scene (:744-747)
        if (state == ST_SHADE && S.num_spheres == 0) end_sample();
    };

    for (;;) {
        // =====================================================================================
        // ADVANCE phase
        // =====================================================================================
        for (int budget = 0; budget < args.advance_budget; ++budget) {
            if (!__any(state < ST_TRAV_CLOSEST)) break;
            if (COUNT) { c[C_ADV_SLOTS]++; if (state < ST_TRAV_CLOSEST) c[C_ADV_ACTIVE]++; }
            if (state == ST_FETCH) {
                uint32_t item = atomicAdd(args.queue, 1u);
                if (item >= P.total_items) {
                    state = ST_DONE;
                } else {
                    const uint32_t tt = (uint32_t)(P.tile * P.tile);
                    const uint32_t k = item / tt, within = item % tt;
                    const uint32_t g = k * (uint32_t)P.shard_count + (uint32_t)P.shard_rank;
                    const uint32_t tx = g % (uint32_t)P.tiles_x, ty = g / (uint32_t)P.tiles_x;
                    const uint32_t sub = within >> 6, l = within & 63u, per_row = (uint32_t)P.tile >> 3;
                    const uint32_t in_x = (sub % per_row) * 8u + (l & 7u), in_y = (sub / per_row) * 8u + (l >> 3);
                    const int x = (int)(tx * (uint32_t)P.tile + in_x), row = (int)(ty * (uint32_t)P.tile + in_y);
                    if (x < W && row < H) {
                        px = x;
                        ky = H - 1 - row;                               // the kernel's y: 0 at the bottom (:984, :1027)
                        out_index = P.compact_output ? (k * tt + in_y * (uint32_t)P.tile + in_x) : ((uint32_t)row * (uint32_t)W + (uint32_t)x);
                        rng = (uint32_t)(px + ky * W) ^ P.seed32;       // :990
                        accum = mk(0, 0, 0);
                        sample = 0;
                        state = ST_GEN;
                    }
                }
            } else if (state == ST_GEN) {
                if (sample >= spp) {
                    // tone map + store :1003-1030
                    float inv_spp = 1.0f / (float)spp;
                    F3 col = accum * inv_spp;
                    col = mk(fmaxf(col.x, 0.0f), fmaxf(col.y, 0.0f), fmaxf(col.z, 0.0f));
                    col = mk(fminf(col.x, 10.0f), fminf(col.y, 10.0f), fminf(col.z, 10.0f));
                    col = mk(dsrt_powf(col.x, P.inv_gamma), dsrt_powf(col.y, P.inv_gamma), dsrt_powf(col.z, P.inv_gamma));
                    col = clamp01(col);
                    const size_t o = (size_t)out_index * 3;
                    args.out_rgb8[o + 0] = (unsigned char)(255.99f * col.x);
                    args.out_rgb8[o + 1] = (unsigned char)(255.99f * col.y);
                    args.out_rgb8[o + 2] = (unsigned char)(255.99f * col.z);
                    if (args.out_f32) { args.out_f32[o + 0] = col.x; args.out_f32[o + 1] = col.y; args.out_f32[o + 2] = col.z; }
                    flush_counters();
                    state = ST_FETCH;
                } else {
                    float jx = ((float)sample + rand01(rng)) / (float)spp;          // :995-996
                    float jy = ((float)sample + rand01(rng)) / (float)spp;
                    float u = ((float)px + jx) / (float)(W - 1);                     // :952-953
                    float v = ((float)ky + jy) / (float)(H - 1);
                    ro = cam_o;
                    rd = ((cam_llc + (cam_h * u)) + (cam_v * v)) - cam_o;           // :957-961
                    depth = 0;
                    L = mk(0, 0, 0);
                    thr = mk(1, 1, 1);
                    if (COUNT) c[C_SAMPLES]++;
                    start_ray(ST_TRAV_CLOSEST, ST_SHADE);      // depth 0: no roulette, max_depth >= 1 (host guarantees)
                }
            } else if (state == ST_BOUNCE) {
                // top of the depth loop :727-744
                bool go = depth < P.max_depth;
                if (go && depth >= 5) {
                    float p = fmaxf(thr.x, fmaxf(thr.y, thr.z));
                    p = fminf(p, 0.95f);
                    if (rand01(rng) > p) go = false;
                    else thr = thr * (1.0f / p);
                }
                if (!go) end_sample();
                else start_ray(ST_TRAV_CLOSEST, ST_SHADE);
            } else if (state == ST_SHADE) {
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
                for (int i = 0; i < S.num_spheres; ++i) {
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
                    const int mtype = __float_as_int(m0.x);
                    if (mtype == MAT_DIFFUSE_LIGHT) {                                                 // :754-758
                        L = L + (thr * mk(m1.w, m2.x, m2.y));
                        end_sample();
                    } else {
                        F3 albedo = mk(m1.x, m1.y, m1.z);                                             // :763-774
                        if (tex_id >= 0 && S.tri_uv) {
                            const float4* uvp = S.tri_uv + (size_t)hit_slot * 2;
                            const float4 u0 = uvp[0], u1 = uvp[1];
                            const float wgt = 1.0f - hit_u - hit_v;
                            const float u_tex = wgt * u0.x + hit_u * u0.z + hit_v * u1.x;
                            const float v_tex = wgt * u0.y + hit_u * u0.w + hit_v * u1.y;
                            albedo = albedo * tex2d(S, tex_id, u_tex, v_tex, c[C_TEX_FETCHES]);
                        }
                        if (mtype == MAT_DIELECTRIC) {                                               // scatter_dielectric :621-661
                            float eta = m2.w;
                            if (eta <= 0.0f || !isfinite(eta)) eta = 1.5f;
                            const float ratio = front ? (1.0f / eta) : eta;
                            const F3 unit = normalize(rd);
                            const float cos_t = fminf(dot(unit * -1.0f, hn), 1.0f);
                            const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
                            const bool cannot = ratio * sin_t > 1.0f;
                            const float rprob = schlick(cos_t, ratio);
                            F3 dir;
                            if (cannot || rprob > rand01(rng)) dir = reflect(unit, hn);
                            else dir = refract(unit, hn, ratio);
                            ro = hp; rd = dir;                      // attenuation is (1,1,1): throughput unchanged
                            depth++;
                            state = ST_BOUNCE;
                        } else if (mtype == MAT_METAL) {                                             // scatter_metal :603-619
                            const F3 refl = reflect(normalize(rd), hn);
                            const float fuzz = fmaxf(0.0f, fminf(1.0f, m2.z));
                            const F3 dir = refl + (random_in_unit_sphere(rng) * fuzz);
                            if (dot(dir, hn) > 0.0f) {
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
                                const F3 Ldir = normalize(mk(-P.sun_dir[0], -P.sun_dir[1], -P.sun_dir[2]));
                                const float cos_t = fmaxf(0.0f, dot(hn, Ldir));
                                if (cos_t > 0.0f) {
                                    sh_o = hp + (hn * 1e-3f);
                                    sh_d = Ldir;
                                    const float pdf_brdf = cos_t / kPi;
                                    const float pdf_mix = 0.5f * 1.0f + 0.5f * pdf_brdf;
                                    const float weight = (cos_t / kPi) / pdf_mix;
                                    pend_contrib = thr * (albedo * (ld3(P.sun_radiance) * weight));
                                    need_shadow = true;
                                }
                            }
                            // ---- next direction.  The shadow ray draws no random numbers, so sampling the bounce
                            //      before tracing it leaves the LCG stream exactly as the reference's order does. ----
                            bool end_after = false;
                            F3 ndir = mk(0, 0, 1), nthr = thr;
                            if (S.num_lights == 0) {                                                 // :852-866
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
                            if (need_shadow) {
                                pend_end = end_after; pend_thr = nthr; pend_o = hp; pend_d = ndir;
                                ro = sh_o; rd = sh_d;
                                start_ray(ST_TRAV_SHADOW, ST_SHADOW_DONE);
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
            } else if (state == ST_SHADOW_DONE) {
                // blocked = scene_hit(shadow_ray) :816: BVH result, then the spheres
                bool blocked = hit_slot >= 0;
                if (!blocked || !ANYHIT) {
                    for (int i = 0; i < S.num_spheres; ++i) {
                        if (COUNT) c[C_SPHERE_TESTS]++;
                        float t_hit; F3 n_hit;
                        if (hit_sphere(S.spheres[i], ro, rd, closest, t_hit, n_hit)) { blocked = true; closest = t_hit; }
                    }
                } else if (COUNT) {
                    c[C_SPHERE_TESTS] += (uint32_t)S.num_spheres;
                }
                if (!blocked) L = L + pend_contrib;
                if (pend_end) end_sample();
                else { thr = pend_thr; ro = pend_o; rd = pend_d; depth++; state = ST_BOUNCE; }
            }
        }

        if (__all(state == ST_DONE)) break;

        // =====================================================================================
        // TRAVERSE phase ("while-while"): lanes at internal nodes step together; lanes that reached a leaf WAIT there
        // until at least as many lanes are parked at leaves as are still descending, then all parked leaves are
        // intersected together, one triangle index at a time, fully predicated.  Each lane still performs exactly the
        // reference's sequence of box tests, triangle tests and pops -- only WHEN a lane runs changes, never what it does.
        // =====================================================================================
        for (int iter = 0;;) {
            const bool walking = (state == ST_TRAV_CLOSEST) || (state == ST_TRAV_SHADOW);
            const int n_walk = __popcll(__ballot(walking));
            if (n_walk == 0) break;
            const int n_wait = __popcll(__ballot(state < ST_TRAV_CLOSEST));
            if (iter >= args.min_walk_iters && n_walk < n_wait) break;

            // ---------------- phase I: internal nodes ----------------
            for (;;) {
                const bool at_node = walking && cur >= 0 && cur != kRefNone;
                const int n_node = __popcll(__ballot(at_node));
                const int n_leaf = __popcll(__ballot(walking && cur < 0));
                if (n_node == 0 || 4 * n_node < args.leaf_ratio4 * n_leaf) break;
                ++iter;
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
                        // nearer child by box centre along the ray :433-453 (only decides anything when both are hit)
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
                            // pop: a postponed child is entered iff its entry distance is still in front of `closest`,
                            // which is bbox_hit(node, ray, t_min, closest) for a box already known to be hit
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
                    if (finished) { cur = kRefNone; state = (state == ST_TRAV_CLOSEST) ? ST_SHADE : ST_SHADOW_DONE; }
                }
            }

            // ---------------- phase L: every lane parked at a leaf intersects it, triangle by triangle :413-420 ----------------
            const bool at_leaf = ((state == ST_TRAV_CLOSEST) || (state == ST_TRAV_SHADOW)) && cur < 0;
            if (__any(at_leaf)) {
                ++iter;
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
                        const int slot = first + i;
                        const float4* tp = S.tri_isect + (size_t)slot * 3;
                        const float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
                        if (COUNT) c[C_TRI_TESTS]++;
                        // Moller-Trumbore :336-353, evaluated in full; the reference's early returns become one predicate.
                        // Each `if (x) return false` is kept as `!(x)` so that NaNs fall the same way.
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
                            closest = t; hit_slot = slot; hit_u = u; hit_v = v;
                            if (COUNT) c[C_HIT_UPDATES]++;
                            if (ANYHIT && state == ST_TRAV_SHADOW) { finished = true; count = 0; }
                        }
                    }
                }
                if (at_leaf) {
                    cur = kRefNone;
                    if (!finished) {
                        for (;;) {
                            if (sp == 0) { finished = true; break; }
                            sp--;
                            uint2 e = lds_stack[wave][sp < K ? sp : K - 1][lane];
                                if (sp >= K) e = args.spill[(size_t)(sp - K) * args.spill_stride + glane];
                            if (closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                        }
                    }
                    if (finished) { cur = kRefNone; state = (state == ST_TRAV_CLOSEST) ? ST_SHADE : ST_SHADOW_DONE; }
                }
            }
        }
    }

    flush_counters();
    if (flags) atomicOr(args.flags, flags);
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

// ---- launchers (called from device_api.hip) -----------------------------------------------------------
template <int K>
static hipError_t launch_k(const RenderArgs& a, int blocks, bool count, bool checked, bool anyhit, hipStream_t stream) {
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    if (count) {
        if (anyhit) hipLaunchKernelGGL((dsrt_render_kernel<K, true, true, true>), grid, block, 0, stream, a);
        else        hipLaunchKernelGGL((dsrt_render_kernel<K, true, true, false>), grid, block, 0, stream, a);
    } else if (checked) {
        hipLaunchKernelGGL((dsrt_render_kernel<K, false, true, true>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((dsrt_render_kernel<K, false, false, true>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs& a, int lds_entries, int blocks, bool count, bool checked, bool anyhit, hipStream_t stream) {
    switch (lds_entries) {
    case 8:  return launch_k<8>(a, blocks, count, checked, anyhit, stream);
    case 12: return launch_k<12>(a, blocks, count, checked, anyhit, stream);
    case 16: return launch_k<16>(a, blocks, count, checked, anyhit, stream);
    case 24: return launch_k<24>(a, blocks, count, checked, anyhit, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_deinterleave(const uint8_t* gathered, uint8_t* image, int W, int H, int tile, int tiles_x, int shard_count,
                               size_t shard_stride_bytes, hipStream_t stream) {
    const size_t n = (size_t)W * H;
    hipLaunchKernelGGL(dsrt_deinterleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, image, W, H, tile, tiles_x,
                       shard_count, shard_stride_bytes);
    return hipGetLastError();
}

hipError_t launch_math(int fn, const float* x, float y, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(dsrt_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, x, y, out, n);
    return hipGetLastError();
}

int kernel_waves_per_block() { return kWavesPerBlock; }

}  // namespace dsrt
