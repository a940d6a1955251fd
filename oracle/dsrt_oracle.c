/*
 * dsrt_oracle.c -- CPU restatement of the reference renderer's sampling loop.
 * TEST INFRASTRUCTURE ONLY; see dsrt_oracle.h for who may use it and for the pinning status: PINNED -- its images equal, byte for byte,
 * those the reference's own kernel rendered (tests/golden/ref_gpu_detmath_images.json, checked by tests/test_oracle_reference_fixtures.py in
 * the CPU suite), and everything that feeds the loop is pinned through oracle/_ref/ref_host (tests/golden/ref_*).
 *
 * Every function names the lines of /root/reference/src/gpu_render.cu it follows.  The data walk
 * is the reference's: 40-byte AoS BVH nodes, 116-byte AoS triangles, tri_indices indirection,
 * int stack, bbox re-tests and all -- deliberately NOT the layout or traversal the HIP kernel uses,
 * so the two implementations only meet at the results.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "dsrt_oracle.h"
#include "../include/dsrt_detmath.h"

#include <math.h>
#include <string.h>

typedef struct { float x, y, z; } V3;

#ifdef DSRT_ORACLE_LIBM
#define O_SINF sinf
#define O_COSF cosf
#define O_POWF powf
#else
#define O_SINF dsrt_sinf
#define O_COSF dsrt_cosf
#define O_POWF dsrt_powf
#endif

static const float PI_F = 3.14159265358979323846f;   /* :96 */

/* ---- float3 helpers, :11-64 ---- */
static inline V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
static inline V3 from_f3(DsrtF3 a) { return v3(a.x, a.y, a.z); }
static inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float len2(V3 a) { return dot(a, a); }
static inline V3 norm(V3 a) {                          /* :51-56 */
    float L = sqrtf(len2(a));
    if (L <= 0.0f) return v3(0.0f, 0.0f, 0.0f);
    float invL = 1.0f / L;
    return v3(a.x * invL, a.y * invL, a.z * invL);
}
static inline V3 clamp01(V3 a) {                       /* :58-64 */
    return v3(fminf(1.0f, fmaxf(0.0f, a.x)), fminf(1.0f, fmaxf(0.0f, a.y)), fminf(1.0f, fmaxf(0.0f, a.z)));
}

typedef struct { V3 orig, dir; } Ray;                  /* RayD :217-227 */
static inline V3 ray_at(const Ray* r, float t) {
    return v3(r->orig.x + t * r->dir.x, r->orig.y + t * r->dir.y, r->orig.z + t * r->dir.z);
}

typedef struct {                                       /* HitRecord :264-279 */
    float t; V3 p; V3 normal; int mat_id; int tri_tex_id; int tri_index; float u, v; int front_face;
} Hit;
#ifdef DSRT_ORACLE_RAY_HOOK
/* experiment builds that #include this file (sah_certificate_probe.c) see every BVH answer of the reference walk: scene_hit below calls this */
static void DSRT_ORACLE_RAY_HOOK(const GPUScene* s, const Ray* ray, float t_min, float t_max, int bvh_hit, const Hit* rec);
#endif
static inline void set_face_normal(Hit* h, const Ray* r, V3 outward) {
    h->front_face = (dot(r->dir, outward) < 0.0f);
    h->normal = h->front_face ? outward : scale(outward, -1.0f);
}

typedef struct { uint32_t state; DsrtOracleCounters* c; } Rng;

/* rand01 :77-80 */
static inline float rand01(Rng* g) {
    g->state = g->state * 1664525u + 1013904223u;
    g->c->rng_draws++;
    return (float)(g->state & 0x00FFFFFFu) / 16777216.0f;
}
float dsrt_oracle_rand01(uint32_t* state) {
    *state = *state * 1664525u + 1013904223u;
    return (float)(*state & 0x00FFFFFFu) / 16777216.0f;
}

/* random_in_unit_sphere :82-91 */
static V3 random_in_unit_sphere(Rng* g) {
    for (;;) {
        float x = rand01(g) * 2.0f - 1.0f;
        float y = rand01(g) * 2.0f - 1.0f;
        float z = rand01(g) * 2.0f - 1.0f;
        V3 p = v3(x, y, z);
        if (len2(p) >= 1.0f) continue;
        return p;
    }
}

/* random_cosine_direction :99-109 */
static V3 random_cosine_direction(Rng* g) {
    float r1 = rand01(g);
    float r2 = rand01(g);
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * PI_F * r1;
    float x = O_COSF(phi) * sqrtf(r2);
    float y = O_SINF(phi) * sqrtf(r2);
    return v3(x, y, z);
}

/* build_onb :112-118 */
static void build_onb(V3 n, V3* u, V3* v, V3* w) {
    *w = norm(n);
    V3 a = (fabsf(w->x) > 0.9f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    *v = norm(cross(*w, a));
    *u = cross(*v, *w);
}

/* sample_cosine_hemisphere :121-141 */
static V3 sample_cosine_hemisphere(V3 normal, Rng* g, float* pdf_out) {
    V3 u, v, w;
    build_onb(normal, &u, &v, &w);
    V3 local = random_cosine_direction(g);
    V3 world = add(add(scale(u, local.x), scale(v, local.y)), scale(w, local.z));
    world = norm(world);
    float cos_theta = fmaxf(0.0f, dot(world, normal));
    *pdf_out = (cos_theta > 0.0f) ? (cos_theta / PI_F) : 0.0f;
    return world;
}

/* sample_sphere_light_direction :145-189 */
static void sample_sphere_light_direction(const GPUSphere* sph, V3 origin, Rng* g, V3* dir_out, float* pdf_out) {
    float z = 2.0f * rand01(g) - 1.0f;
    float phi = 2.0f * PI_F * rand01(g);
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float x = r * O_COSF(phi);
    float y = r * O_SINF(phi);
    V3 local = v3(x, y, z);
    V3 center = from_f3(sph->center);
    V3 p_light = add(center, scale(local, sph->radius));
    V3 to_light = sub(p_light, origin);
    float dist2 = len2(to_light);
    float dist = sqrtf(dist2);
    if (dist <= 0.0f) { *pdf_out = 0.0f; *dir_out = v3(0, 0, 1); return; }
    V3 wi = scale(to_light, 1.0f / dist);
    V3 n_light = norm(sub(p_light, center));
    float cos_l = fmaxf(0.0f, dot(n_light, scale(wi, -1.0f)));
    if (cos_l <= 0.0f) { *pdf_out = 0.0f; *dir_out = wi; return; }
    float area = 4.0f * PI_F * sph->radius * sph->radius;
    *pdf_out = dist2 / (cos_l * area);
    *dir_out = wi;
}

/* reflect :195, refract :199-206, schlick :208-212 */
static inline V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.0f * dot(v, n))); }
static V3 refract(V3 v, V3 n, float eta) {
    V3 uv = norm(v);
    float cos_theta = fminf(dot(scale(uv, -1.0f), n), 1.0f);
    V3 perp = scale(add(uv, scale(n, cos_theta)), eta);
    V3 par = scale(n, -sqrtf(fabsf(1.0f - len2(perp))));
    return add(perp, par);
}
static float schlick(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * O_POWF(1.0f - cosine, 5.0f);
}

/* tex2D :232-259 */
static V3 tex2d(const GPUScene* s, int tex_id, float u, float v, DsrtOracleCounters* c) {
    if (tex_id < 0 || tex_id >= s->num_textures || !s->textures || !s->texture_pool) return v3(1.0f, 1.0f, 1.0f);
    const GPUTextureHeader* th = &s->textures[tex_id];
    int w = th->width, h = th->height;
    u = u - floorf(u);
    v = v - floorf(v);
    int i = (int)(u * (float)(w - 1));
    int j = (int)((1.0f - v) * (float)(h - 1));
    int idx = th->offset + (j * w + i) * 3;
    if (idx < 0 || idx + 2 >= s->texture_pool_floats) return v3(1.0f, 1.0f, 1.0f);
    c->tex_fetches++;
    return v3(s->texture_pool[idx + 0], s->texture_pool[idx + 1], s->texture_pool[idx + 2]);
}

/* bbox_hit :285-315 */
static int bbox_hit(const GPUBVHNode* node, const Ray* r, float t_min, float t_max, DsrtOracleCounters* c) {
    c->box_tests++;
    const float o[3] = { r->orig.x, r->orig.y, r->orig.z };
    const float d[3] = { r->dir.x, r->dir.y, r->dir.z };
    const float mn[3] = { node->bbox_min.x, node->bbox_min.y, node->bbox_min.z };
    const float mx[3] = { node->bbox_max.x, node->bbox_max.y, node->bbox_max.z };
    for (int a = 0; a < 3; ++a) {
        float invD = 1.0f / d[a];
        float t0 = (mn[a] - o[a]) * invD;
        float t1 = (mx[a] - o[a]) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return 0;
    }
    return 1;
}

/* hit_triangle_index :322-380 */
static int hit_triangle_index(const GPUScene* s, int tri_index, const Ray* ray, float t_min, float t_max, Hit* rec,
                              DsrtOracleCounters* c) {
    c->tri_tests++;
    const GPUTriangle* tri = &s->triangles[tri_index];
    V3 v0 = from_f3(tri->v0), v1 = from_f3(tri->v1), v2 = from_f3(tri->v2);
    V3 edge1 = sub(v1, v0), edge2 = sub(v2, v0);
    V3 pvec = cross(ray->dir, edge2);
    float det = dot(edge1, pvec);
    if (fabsf(det) < 1e-8f) return 0;
    float invDet = 1.0f / det;
    V3 tvec = sub(ray->orig, v0);
    float u = dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return 0;
    V3 qvec = cross(tvec, edge1);
    float v = dot(ray->dir, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = dot(edge2, qvec) * invDet;
    if (t < t_min || t > t_max) return 0;

    rec->t = t;
    rec->p = ray_at(ray, t);
    float w = 1.0f - u - v;
    V3 n = add(add(scale(from_f3(tri->n0), w), scale(from_f3(tri->n1), u)), scale(from_f3(tri->n2), v));
    n = norm(n);
    set_face_normal(rec, ray, n);
    rec->u = u;
    rec->v = v;
    rec->mat_id = tri->material_id;
    rec->tri_tex_id = tri->albedo_tex;
    rec->tri_index = tri_index;
    return 1;
}

/* bvh_hit_closest :387-473 */
static int bvh_hit_closest(const GPUScene* s, const Ray* ray, float t_min, float t_max, Hit* out, DsrtOracleCounters* c) {
    if (!s->bvh_nodes || s->num_bvh_nodes <= 0 || !s->tri_indices) return 0;
    int stack[64];
    int sp = 0;
    int node_index = 0;
    int hit_anything = 0;
    float closest = t_max;
    Hit tmp;
    c->box_fetches++;                                  /* the root box */
    for (;;) {
        const GPUBVHNode* node = &s->bvh_nodes[node_index];
        if (bbox_hit(node, ray, t_min, closest, c)) {
            c->nodes_entered++;
            if (node->tri_count > 0) {
                for (int i = 0; i < node->tri_count; ++i) {
                    int tri_idx = s->tri_indices[node->tri_offset + i];
                    if (hit_triangle_index(s, tri_idx, ray, t_min, closest, &tmp, c)) {
                        hit_anything = 1;
                        closest = tmp.t;
                        *out = tmp;
                        c->hit_updates++;
                    }
                }
                if (sp == 0) break;
                node_index = stack[--sp];
            } else {
                c->internal_entered++;
                c->box_fetches += 2;
                const GPUBVHNode* left = &s->bvh_nodes[node->left];
                const GPUBVHNode* right = &s->bvh_nodes[node->right];
                int hit_left = bbox_hit(left, ray, t_min, closest, c);
                int hit_right = bbox_hit(right, ray, t_min, closest, c);
                if (hit_left && hit_right) {
                    V3 cL = v3(0.5f * (left->bbox_min.x + left->bbox_max.x), 0.5f * (left->bbox_min.y + left->bbox_max.y),
                               0.5f * (left->bbox_min.z + left->bbox_max.z));
                    V3 cR = v3(0.5f * (right->bbox_min.x + right->bbox_max.x), 0.5f * (right->bbox_min.y + right->bbox_max.y),
                               0.5f * (right->bbox_min.z + right->bbox_max.z));
                    float dL = dot(sub(cL, ray->orig), ray->dir);
                    float dR = dot(sub(cR, ray->orig), ray->dir);
                    int near_idx = (dL < dR) ? node->left : node->right;
                    int far_idx = (dL < dR) ? node->right : node->left;
                    if (sp >= 64) return -1;           /* the reference would overrun its stack here */
                    stack[sp++] = far_idx;
                    if ((uint64_t)sp > c->max_stack) c->max_stack = (uint64_t)sp;
                    node_index = near_idx;
                } else if (hit_left) {
                    node_index = node->left;
                } else if (hit_right) {
                    node_index = node->right;
                } else {
                    if (sp == 0) break;
                    node_index = stack[--sp];
                }
            }
        } else {
            if (sp == 0) break;
            node_index = stack[--sp];
        }
    }
    return hit_anything;
}

/* hit_sphere :478-504 */
static int hit_sphere(const GPUSphere* sph, const Ray* ray, float t_min, float t_max, float* t_out, V3* n_out,
                      DsrtOracleCounters* c) {
    c->sphere_tests++;
    V3 center = from_f3(sph->center);
    V3 oc = sub(ray->orig, center);
    float a = dot(ray->dir, ray->dir);
    float half_b = dot(oc, ray->dir);
    float cc = dot(oc, oc) - sph->radius * sph->radius;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return 0;
    float sqrtd = sqrtf(disc);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return 0;
    }
    *t_out = root;
    V3 p = ray_at(ray, root);
    *n_out = scale(sub(p, center), 1.0f / sph->radius);
    return 1;
}

/* scene_hit :509-551 */
static int scene_hit(const GPUScene* s, const Ray* ray, float t_min, float t_max, Hit* rec, DsrtOracleCounters* c) {
    c->rays++;
    Hit best;
    memset(&best, 0, sizeof best);
    int hit_any = 0;
    float closest = t_max;
    Hit tri_rec;
    const int bvh_hit = bvh_hit_closest(s, ray, t_min, closest, &tri_rec, c) > 0;
#ifdef DSRT_ORACLE_RAY_HOOK
    /* (experiment builds only, e.g. sah_certificate_probe.c, which #includes this file: every BVH answer of the reference walk is shown to the experiment) */
    DSRT_ORACLE_RAY_HOOK(s, ray, t_min, t_max, bvh_hit, &tri_rec);
#endif
    if (bvh_hit) {
        hit_any = 1;
        closest = tri_rec.t;
        best = tri_rec;
    }
    for (int i = 0; i < s->num_spheres; ++i) {
        float t_hit; V3 n_hit;
        if (hit_sphere(&s->spheres[i], ray, t_min, closest, &t_hit, &n_hit, c)) {
            hit_any = 1;
            closest = t_hit;
            best.t = t_hit;
            best.p = ray_at(ray, t_hit);
            set_face_normal(&best, ray, n_hit);
            best.mat_id = s->spheres[i].material_id;
            best.tri_tex_id = -1;
            best.tri_index = -1;
            best.u = 0.0f;
            best.v = 0.0f;
        }
    }
    if (hit_any) *rec = best;
    return hit_any;
}

int dsrt_oracle_scene_hit(const GPUScene* scene, const float orig[3], const float dir[3], float t_min, float t_max,
                          float out[9], int ids[4]) {
    DsrtOracleCounters c;
    memset(&c, 0, sizeof c);
    Ray r = { v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]) };
    Hit h;
    memset(&h, 0, sizeof h);
    int hit = scene_hit(scene, &r, t_min, t_max, &h, &c);
    if (hit) {
        out[0] = h.t; out[1] = h.p.x; out[2] = h.p.y; out[3] = h.p.z;
        out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = h.u; out[8] = h.v;
        ids[0] = h.mat_id; ids[1] = h.tri_tex_id; ids[2] = h.tri_index; ids[3] = h.front_face;
    }
    return hit;
}

/* scatter_metal :603-619 */
static int scatter_metal(const GPUMaterial* mat, const Ray* in, const Hit* rec, Rng* g, Ray* scattered, V3* atten, V3 albedo) {
    V3 reflected = reflect(norm(in->dir), rec->normal);
    float fuzz = fmaxf(0.0f, fminf(1.0f, mat->fuzz));
    V3 dir = add(reflected, scale(random_in_unit_sphere(g), fuzz));
    scattered->orig = rec->p;
    scattered->dir = dir;
    *atten = albedo;
    return dot(scattered->dir, rec->normal) > 0.0f;
}

/* scatter_dielectric :621-661 */
static int scatter_dielectric(const GPUMaterial* mat, const Ray* in, const Hit* rec, Rng* g, Ray* scattered, V3* atten) {
    *atten = v3(1.0f, 1.0f, 1.0f);
    float eta = mat->ref_idx;
    if (eta <= 0.0f || !isfinite(eta)) eta = 1.5f;
    float ratio = rec->front_face ? (1.0f / eta) : eta;
    V3 unit_dir = norm(in->dir);
    float cos_theta = fminf(dot(scale(unit_dir, -1.0f), rec->normal), 1.0f);
    float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    int cannot_refract = ratio * sin_theta > 1.0f;
    float reflect_prob = schlick(cos_theta, ratio);
    V3 direction;
    if (cannot_refract || reflect_prob > rand01(g)) direction = reflect(unit_dir, rec->normal);
    else direction = refract(unit_dir, rec->normal, ratio);
    scattered->orig = rec->p;
    scattered->dir = direction;
    return 1;
}

static int is_emissive_sphere(const GPUScene* s, int i) {           /* :843-846 */
    const GPUMaterial* lm = &s->materials[s->spheres[i].material_id];
    return lm->type == MAT_DIFFUSE_LIGHT && (lm->emissive.x > 0 || lm->emissive.y > 0 || lm->emissive.z > 0);
}

/* ray_color :715-936 */
static V3 ray_color(const GPUScene* s, Ray ray, Rng* g, DsrtOracleCounters* c) {
    V3 L = v3(0, 0, 0);
    V3 throughput = v3(1, 1, 1);
    int max_depth = (s->params.max_depth > 0) ? s->params.max_depth : 12;
    c->samples++;

    for (int depth = 0; depth < max_depth; ++depth) {
        if (depth >= 5) {                                            /* Russian roulette :732-738 */
            float p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
            p = fminf(p, 0.95f);
            if (rand01(g) > p) break;
            throughput = scale(throughput, 1.0f / p);
        }

        Hit rec;
        if (!scene_hit(s, &ray, 0.001f, 1e9f, &rec, c)) break;       /* :744-747 */
        if (depth == 0) c->primary_hits++;

        const GPUMaterial* mat = &s->materials[rec.mat_id];
        c->shaded_hits++;

        if (mat->type == MAT_DIFFUSE_LIGHT) {                        /* :754-758 */
            L = add(L, mul(throughput, from_f3(mat->emissive)));
            break;
        }

        V3 albedo = from_f3(mat->albedo);                            /* :763-774 */
        if (rec.tri_tex_id >= 0) {
            const GPUTriangle* tri = &s->triangles[rec.tri_index];
            float w = 1.0f - rec.u - rec.v;
            float u_tex = w * tri->uv0.x + rec.u * tri->uv1.x + rec.v * tri->uv2.x;
            float v_tex = w * tri->uv0.y + rec.u * tri->uv1.y + rec.v * tri->uv2.y;
            albedo = mul(albedo, tex2d(s, rec.tri_tex_id, u_tex, v_tex, c));
        }

        if (mat->type == MAT_DIELECTRIC || mat->type == MAT_METAL) { /* :779-795 */
            Ray scattered; V3 atten; int ok;
            if (mat->type == MAT_DIELECTRIC) ok = scatter_dielectric(mat, &ray, &rec, g, &scattered, &atten);
            else ok = scatter_metal(mat, &ray, &rec, g, &scattered, &atten, albedo);
            if (!ok) break;
            throughput = mul(throughput, atten);
            ray = scattered;
            continue;
        }

        if (s->sun_enabled) {                                        /* :800-836 */
            V3 Ldir = norm(v3(-s->sun_dir.x, -s->sun_dir.y, -s->sun_dir.z));
            float cos_theta = fmaxf(0.0f, dot(rec.normal, Ldir));
            if (cos_theta > 0.0f) {
                Ray shadow = { add(rec.p, scale(rec.normal, 1e-3f)), Ldir };
                Hit shadow_rec;
                int blocked = scene_hit(s, &shadow, 0.001f, 1e9f, &shadow_rec, c);
                if (!blocked) {
                    float pdf_light = 1.0f;
                    float pdf_brdf = cos_theta / PI_F;
                    float pdf_mix = 0.5f * pdf_light + 0.5f * pdf_brdf;
                    float scattering_pdf = cos_theta / PI_F;
                    float weight = scattering_pdf / pdf_mix;
                    V3 sun = mul(throughput, mul(albedo, scale(from_f3(s->sun_radiance), weight)));
                    L = add(L, sun);
                }
            }
        }

        int num_lights = 0;                                          /* :841-847 */
        for (int i = 0; i < s->num_spheres; i++) if (is_emissive_sphere(s, i)) num_lights++;

        if (num_lights == 0) {                                       /* :852-866 */
            float pdf_brdf;
            V3 dir = sample_cosine_hemisphere(rec.normal, g, &pdf_brdf);
            if (pdf_brdf <= 0) break;
            float cos_theta = fmaxf(0.0f, dot(dir, rec.normal));
            float scattering_pdf = cos_theta / PI_F;
            throughput = mul(throughput, scale(albedo, scattering_pdf / pdf_brdf));
            ray.orig = rec.p;
            ray.dir = dir;
            continue;
        }

        V3 dir;                                                      /* :871-932 */
        float pdf_val = 0.0f;
        float choose = rand01(g);
        if (choose < 0.5f) {
            int k = (int)(rand01(g) * (float)num_lights);
            if (k >= num_lights) k = num_lights - 1;
            int found = 0, light_idx = -1;
            for (int i = 0; i < s->num_spheres; i++) {
                if (is_emissive_sphere(s, i)) {
                    if (found == k) { light_idx = i; break; }
                    found++;
                }
            }
            float pdf_light_cond = 0.0f;
            sample_sphere_light_direction(&s->spheres[light_idx], rec.p, g, &dir, &pdf_light_cond);
            if (pdf_light_cond <= 0) break;
            float cos_theta = fmaxf(0.0f, dot(dir, rec.normal));
            if (cos_theta <= 0) break;
            float pdf_light = pdf_light_cond / (float)num_lights;
            float pdf_brdf = cos_theta / PI_F;
            pdf_val = 0.5f * pdf_light + 0.5f * pdf_brdf;
        } else {
            float pdf_brdf = 0.0f;
            dir = sample_cosine_hemisphere(rec.normal, g, &pdf_brdf);
            if (pdf_brdf <= 0) break;
            pdf_val = 0.5f * pdf_brdf;
        }
        float cos_theta = fmaxf(0.0f, dot(dir, rec.normal));
        float scattering_pdf = cos_theta / PI_F;
        float weight = scattering_pdf / pdf_val;
        throughput = mul(throughput, scale(albedo, weight));
        ray.orig = rec.p;
        ray.dir = dir;
    }
    return clamp01(L);                                               /* :935 */
}

/* make_camera_ray_jittered :941-968 (vec3 there is float, inc/vec3.h:14-22) */
static Ray camera_ray(const GPUCamera* cam, int px, int py, int W, int H, float jx, float jy) {
    float u = ((float)px + jx) / (float)(W - 1);
    float v = ((float)py + jy) / (float)(H - 1);
    V3 o = from_f3(cam->origin);
    V3 d = sub(add(add(from_f3(cam->lower_left_corner), scale(from_f3(cam->horizontal), u)),
                   scale(from_f3(cam->vertical), v)), o);
    Ray r = { o, d };
    return r;
}

/* render_kernel :973-1031 with the launcher's gamma default :1043-1045 */
int dsrt_oracle_render_rows(const GPUScene* s, int W, int H, int y0, int y1, uint8_t* rgb8, float* rgb_f32,
                            DsrtOracleCounters* counters) {
    DsrtOracleCounters local;
    memset(&local, 0, sizeof local);
    if (!s || W < 2 || H < 2 || y0 < 0 || y1 > H || y0 > y1) return -1;
    const float gamma = (s->params.gamma > 0.0f) ? s->params.gamma : 1.0f;
    const float inv_gamma = 1.0f / gamma;
    int spp = s->params.samples_per_pixel;
    if (spp < 1) spp = 1;

    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < W; ++x) {
            Rng g = { (uint32_t)(x + y * W) ^ (uint32_t)(s->seed & 0xFFFFFFFFu), &local };
            V3 accum = v3(0, 0, 0);
            for (int k = 0; k < spp; ++k) {
                float jx = ((float)k + rand01(&g)) / (float)spp;
                float jy = ((float)k + rand01(&g)) / (float)spp;
                Ray ray = camera_ray(&s->camera, x, y, W, H, jx, jy);
                accum = add(accum, ray_color(s, ray, &g, &local));
            }
            float inv_spp = 1.0f / (float)spp;
            V3 color = scale(accum, inv_spp);
            color.x = fmaxf(color.x, 0.0f); color.y = fmaxf(color.y, 0.0f); color.z = fmaxf(color.z, 0.0f);
            color.x = fminf(color.x, 10.0f); color.y = fminf(color.y, 10.0f); color.z = fminf(color.z, 10.0f);
            color.x = O_POWF(color.x, inv_gamma);
            color.y = O_POWF(color.y, inv_gamma);
            color.z = O_POWF(color.z, inv_gamma);
            color = clamp01(color);
            size_t idx = ((size_t)(H - 1 - y) * (size_t)W + (size_t)x) * 3;
            if (rgb8) {
                rgb8[idx + 0] = (unsigned char)(255.99f * color.x);
                rgb8[idx + 1] = (unsigned char)(255.99f * color.y);
                rgb8[idx + 2] = (unsigned char)(255.99f * color.z);
            }
            if (rgb_f32) { rgb_f32[idx + 0] = color.x; rgb_f32[idx + 1] = color.y; rgb_f32[idx + 2] = color.z; }
        }
    }
    if (counters) {
        uint64_t* dst = (uint64_t*)counters;
        const uint64_t* src = (const uint64_t*)&local;
        size_t n = sizeof(DsrtOracleCounters) / sizeof(uint64_t);
        for (size_t i = 0; i < n; ++i) {
            if (&dst[i] == &counters->max_stack) { if (src[i] > dst[i]) dst[i] = src[i]; }
            else dst[i] += src[i];
        }
    }
    return 0;
}

int dsrt_oracle_bbox_hit(const float lo[3], const float hi[3], const float orig[3], const float dir[3], float t_min, float t_max) {
    DsrtOracleCounters c;
    memset(&c, 0, sizeof c);
    GPUBVHNode n;
    memset(&n, 0, sizeof n);
    n.bbox_min.x = lo[0]; n.bbox_min.y = lo[1]; n.bbox_min.z = lo[2];
    n.bbox_max.x = hi[0]; n.bbox_max.y = hi[1]; n.bbox_max.z = hi[2];
    Ray r = { v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]) };
    return bbox_hit(&n, &r, t_min, t_max, &c);
}

/* Test hooks for the known-answer vectors of the reference's device helpers (tests/golden/ref_devkat.json). */
void dsrt_oracle_random_in_unit_sphere(uint32_t* state, float out[3]) {
    DsrtOracleCounters c; memset(&c, 0, sizeof c);
    Rng g = { *state, &c };
    V3 p = random_in_unit_sphere(&g);
    *state = g.state; out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void dsrt_oracle_random_cosine_direction(uint32_t* state, float out[3]) {
    DsrtOracleCounters c; memset(&c, 0, sizeof c);
    Rng g = { *state, &c };
    V3 p = random_cosine_direction(&g);
    *state = g.state; out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void dsrt_oracle_camera_ray(const GPUCamera* cam, int px, int py, int W, int H, float jx, float jy, float orig[3], float dir[3]) {
    Ray r = camera_ray(cam, px, py, W, H, jx, jy);
    orig[0] = r.orig.x; orig[1] = r.orig.y; orig[2] = r.orig.z; dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

/* Test hooks for the known answers of the reference's material / frame helpers (tests/golden/ref_matkat.json): the functions ray_color's
 * specular branches and the cosine sampler are made of, called exactly as ray_color calls them. */
void dsrt_oracle_reflect(const float v[3], const float n[3], float out[3]) {
    V3 r = reflect(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void dsrt_oracle_refract(const float v[3], const float n[3], float eta, float out[3]) {
    V3 r = refract(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void dsrt_oracle_normalize(const float v[3], float out[3]) {
    V3 r = norm(v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int dsrt_oracle_scatter_metal(const float dir[3], const float n[3], float fuzz, uint32_t* state, float out_dir[3]) {
    DsrtOracleCounters c; memset(&c, 0, sizeof c);
    Rng g = { *state, &c };
    GPUMaterial m; memset(&m, 0, sizeof m); m.fuzz = fuzz;
    Ray in = { v3(0, 0, 0), v3(dir[0], dir[1], dir[2]) }, sc;
    Hit rec; memset(&rec, 0, sizeof rec); rec.normal = v3(n[0], n[1], n[2]);
    V3 att;
    const int ok = scatter_metal(&m, &in, &rec, &g, &sc, &att, v3(1, 1, 1));
    *state = g.state; out_dir[0] = sc.dir.x; out_dir[1] = sc.dir.y; out_dir[2] = sc.dir.z;
    return ok;
}
void dsrt_oracle_scatter_dielectric(const float dir[3], const float n[3], int front_face, float ref_idx, uint32_t* state, float out_dir[3]) {
    DsrtOracleCounters c; memset(&c, 0, sizeof c);
    Rng g = { *state, &c };
    GPUMaterial m; memset(&m, 0, sizeof m); m.ref_idx = ref_idx;
    Ray in = { v3(0, 0, 0), v3(dir[0], dir[1], dir[2]) }, sc;
    Hit rec; memset(&rec, 0, sizeof rec); rec.normal = v3(n[0], n[1], n[2]); rec.front_face = front_face;
    V3 att;
    (void)scatter_dielectric(&m, &in, &rec, &g, &sc, &att);
    *state = g.state; out_dir[0] = sc.dir.x; out_dir[1] = sc.dir.y; out_dir[2] = sc.dir.z;
}
void dsrt_oracle_build_onb(const float n[3], float u[3], float v[3], float w[3]) {
    V3 uu, vv, ww;
    build_onb(v3(n[0], n[1], n[2]), &uu, &vv, &ww);
    u[0] = uu.x; u[1] = uu.y; u[2] = uu.z; v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; w[0] = ww.x; w[1] = ww.y; w[2] = ww.z;
}
float dsrt_oracle_schlick(float cosine, float ref_idx) { return schlick(cosine, ref_idx); }

float dsrt_oracle_sinf(float x) { return dsrt_sinf(x); }
float dsrt_oracle_cosf(float x) { return dsrt_cosf(x); }
float dsrt_oracle_powf(float x, float y) { return dsrt_powf(x, y); }
