// device_math.h -- fp32 vector helpers, sampling routines and primitive tests shared by the render kernels.
// Every function restates one of src/gpu_render.cu (lines cited inline) with the same operations in the same order.
#pragma once

#include "device_layout.h"
#include "../../include/dsrt_detmath.h"

#ifdef DSRT_DEVICE_LIBM
// Second compilation of render_kernel.hip (Makefile: hip_render_kernel_devlibm.o; DsrtRenderDesc.math_mode 1): the device math library's own sinf / cosf /
// powf in place of include/dsrt_detmath.h.  These three functions are the one place where the default mode deliberately differs from what the reference's
// source computes when the same compiler builds it (numerics contract); in this compilation nothing differs, and its images equal, byte
// for byte, those of the reference's own kernel translated by hipify-perl and run on the same GPU (oracle/_ref/ref_gpu, tests/test_gpu_reference_kernel.py).
#define dsrt_sinf(x) sinf(x)
#define dsrt_cosf(x) cosf(x)
#define dsrt_powf(x, y) powf(x, y)
#endif

namespace dsrt {

// Wave votes straight from the compare mask (HIP's __ballot/__any/__all take an int and cost a v_cndmask + v_cmp each).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }


typedef float v2f __attribute__((ext_vector_type(2)));     // a (left, right) pair: maps onto the packed fp32 VALU ops

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

// rng_mode 0 -- the reference's generator: ONE 32-bit LCG stream per pixel, shared by all its samples (:77-80, :990).
__device__ __forceinline__ float rand01(uint32_t& s) {          // :77-80
    s = s * 1664525u + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / 16777216.0f;
}

// rng_mode 1 -- rocRAND's Philox4x32-10, in stateless (counter) form.  Draw n of sample stream `sub` is word n & 3 of
// ten_rounds(counter = (n >> 2, 0, sub.lo, sub.hi), key = seed), which is exactly what rocrand_init(seed, sub, 0, &st)
// followed by n + 1 calls of rocrand(&st) returns (rocrand_philox4x32_10.h: seed() / restart() / next());
// dsrt_selftest_philox checks that on the device against rocRAND's own engine.  Every (pixel, sample) has its own
// sub-sequence, so samples are independent work items.  The 32-bit word is mapped to [0,1) like the LCG's: low 24 bits / 2^24.
struct PhiloxStream {
    uint32_t key0, key1;      // seed
    uint32_t sub0, sub1;      // sub-sequence = pixel * spp + sample
    uint32_t n;               // draws taken so far in this sub-sequence
    // the 4-word block the last draw came from (one 10-round evaluation serves up to four consecutive draws); a scratch of
    // the current advance step, not part of the stream's identity
    uint32_t blk = 0xFFFFFFFFu, blk_sub0 = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
};
__device__ __forceinline__ void philox4x32_10_block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t& o0, uint32_t& o1,
                                                    uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
__device__ __forceinline__ uint32_t philox4x32_10_word(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t which) {
    uint32_t o0, o1, o2, o3;
    philox4x32_10_block(k0, k1, c0, c1, c2, c3, o0, o1, o2, o3);
    return which == 0 ? o0 : (which == 1 ? o1 : (which == 2 ? o2 : o3));
}
__device__ __forceinline__ float rand01(PhiloxStream& g) {
    const uint32_t b = g.n >> 2;
    if (b != g.blk || g.sub0 != g.blk_sub0) {                // (a sample's sub-sequences differ in their low word)
        philox4x32_10_block(g.key0, g.key1, b, 0u, g.sub0, g.sub1, g.w0, g.w1, g.w2, g.w3);
        g.blk = b; g.blk_sub0 = g.sub0;
    }
    const uint32_t which = g.n & 3u;
    const uint32_t w = which == 0 ? g.w0 : (which == 1 ? g.w1 : (which == 2 ? g.w2 : g.w3));
    g.n++;
    return (float)(w & 0x00FFFFFFu) / 16777216.0f;
}

template <class Rng>
__device__ __forceinline__ F3 random_in_unit_sphere(Rng& rng) {   // :82-91
    for (;;) {
        float x = rand01(rng) * 2.0f - 1.0f;
        float y = rand01(rng) * 2.0f - 1.0f;
        float z = rand01(rng) * 2.0f - 1.0f;
        F3 p = mk(x, y, z);
        if (dot(p, p) >= 1.0f) continue;
        return p;
    }
}

// build_onb :112-118
__device__ __forceinline__ void build_onb(F3 n, F3& u, F3& v, F3& w) {
    w = normalize(n);
    F3 a = (fabsf(w.x) > 0.9f) ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
    v = normalize(cross(w, a));
    u = cross(v, w);
}

// sample_cosine_hemisphere :121-141 with random_cosine_direction :99-109
template <class Rng>
__device__ __forceinline__ F3 sample_cosine_hemisphere(F3 normal, Rng& rng, float& pdf) {
    F3 u, v, w;
    build_onb(normal, u, v, w);
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
template <class Rng>
__device__ __forceinline__ void sample_sphere_light(const GPUSphere& sph, F3 origin, Rng& rng, F3& dir, float& pdf) {
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

// scatter_metal :603-619: the scattered direction; returns whether the ray goes on (false: absorbed, the path ends).
template <class Rng>
__device__ __forceinline__ bool scatter_metal(F3 rd, F3 hn, float mat_fuzz, Rng& rng, F3& dir) {
    const F3 refl = reflect(normalize(rd), hn);
    const float fuzz = fmaxf(0.0f, fminf(1.0f, mat_fuzz));
    dir = refl + (random_in_unit_sphere(rng) * fuzz);
    return dot(dir, hn) > 0.0f;
}

// scatter_dielectric :621-661: the scattered direction (no attenuation; always goes on).  The random number is drawn only when the ray
// CAN refract (`||` short-circuits, as in the reference).
template <class Rng>
__device__ __forceinline__ F3 scatter_dielectric(F3 rd, F3 hn, bool front, float ref_idx, Rng& rng) {
    float eta = ref_idx;
    if (eta <= 0.0f || !isfinite(eta)) eta = 1.5f;
    const float ratio = front ? (1.0f / eta) : eta;
    const F3 unit = normalize(rd);
    const float cos_t = fminf(dot(unit * -1.0f, hn), 1.0f);
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    const bool cannot = ratio * sin_t > 1.0f;
    const float rprob = schlick(cos_t, ratio);
    if (cannot || rprob > rand01(rng)) return reflect(unit, hn);
    return refract(unit, hn, ratio);
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

}  // namespace dsrt
