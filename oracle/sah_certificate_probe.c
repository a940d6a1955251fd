/*
 * sah_certificate_probe.c -- EXPERIMENT, CPU only (TEST INFRASTRUCTURE like everything under oracle/; nothing in the product uses it).
 *
 * Question (round-3 review, item 5): can a ray be answered on a BETTER tree (binned SAH, 1.37x faster frames) and still be, provably, the reference's
 * answer on the reference's own median-split tree?  The reference's closest-hit walk (src/gpu_render.cu:387-473) returns, for a ray, the accepted
 * triangle of smallest t -- EXCEPT where its answer depends on its own boxes or its own order:
 *   (a) a triangle whose leaf (or an ancestor) has a zero-thickness box is never reached: bbox_hit's `t_max <= t_min` holds with equality (:312);
 *   (b) two accepted triangles at EXACTLY the same t: the one tested later wins (`t > t_max` rejects, :353);
 *   (c) the smallest-t triangle T is reached only if its leaf's box test passes for this ray, and with every `closest` the walk can hold at that moment
 *       only if t_T > the leaf box's computed entry distance (a hit computed an ulp in front of its own box's entry).
 * Claim: if the accepted triangle of smallest t, T, (1) is not excluded by (a), (2) has no exact tie, and (3) passes its REFERENCE-leaf box's slab test -- evaluated
 * with the reference's own float operations -- with t_entry < t_T, then the reference's walk returns T, whatever else the tree holds: every ancestor's box
 * contains the leaf's, the slab arithmetic is monotone in the box bounds (float subtraction and multiplication by a fixed 1/d round monotonically), so every
 * ancestor's entry distance is <= the leaf's and its exit >= the leaf's; `closest` never drops below t_T (T is the minimum); hence every box on the
 * root-to-leaf path passes whenever it is tested, T is tested with closest >= t_T and accepted, and nothing accepted afterwards has t <= t_T.
 * Moller-Trumbore itself (:322-353) depends on the ray and the triangle only -- not on any tree.
 *
 * This file checks the claim on real paths: it renders rows with the oracle (the reference's walk on the median tree) and, for every BVH query the walk
 * answers, also finds T by a conservative walk of a SECOND tree over the same triangles (box tests relaxed by a relative margin, so that no accepted
 * triangle is culled), evaluates (1)-(3), and counts: rays, rays it would have to flag (re-trace on the reference tree), and -- the number that must be
 * zero -- rays NOT flagged whose certified answer differs from the reference's (hit / triangle / t, u, v bit patterns).
 *
 * Build: oracle/Makefile, target libdsrt_sahprobe.so (gcc, same flags as the oracle).
 */
#define DSRT_ORACLE_RAY_HOOK sah_probe_hook
#include "dsrt_oracle.c"

typedef struct {
    const GPUBVHNode* nodes;        /* the second tree (reference node format) over the SAME triangle array */
    int num_nodes;
    const int* tri_indices;
    const uint8_t* never_hit;       /* per triangle: 1 = unreachable on the reference tree (zero-thickness box on its path) */
    const float* leaf_box;          /* per triangle: lo[3], hi[3] of its leaf on the reference tree */
    float mu;
    uint64_t rays, ref_hits, flagged_tie, flagged_leaf_box, flagged_zero_dir, mismatched, second_tree_tri_tests, second_tree_nodes, shadow_like;
    int first_bad_tri_ref, first_bad_tri_mine;
} SahProbe;

static SahProbe g_probe;

/* Moller-Trumbore exactly as hit_triangle_index :336-353 computes it, without the record: returns 1 and (t, u, v) when every test but the one against
 * `closest` passes. */
static int mt_accepts(const GPUScene* s, int tri_index, const Ray* ray, float t_min, float* t_out, float* u_out, float* v_out) {
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
    if (t < t_min) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* the reference's slab arithmetic (:285-315) on one box, returning the interval instead of the verdict */
static void slab_interval(const float lo[3], const float hi[3], const Ray* r, float t_min, float* te, float* tx) {
    const float o[3] = { r->orig.x, r->orig.y, r->orig.z };
    const float d[3] = { r->dir.x, r->dir.y, r->dir.z };
    float a = t_min, b = INFINITY;
    for (int k = 0; k < 3; ++k) {
        float invD = 1.0f / d[k];
        float t0 = (lo[k] - o[k]) * invD;
        float t1 = (hi[k] - o[k]) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        a = t0 > a ? t0 : a;
        b = t1 < b ? t1 : b;
    }
    *te = a; *tx = b;
}

static void sah_probe_hook(const GPUScene* s, const Ray* ray, float t_min, float t_max, int ref_hit, const Hit* ref_rec) {
    SahProbe* P = &g_probe;
    if (!P->nodes) return;
    P->rays++;
    if (ref_hit) P->ref_hits++;
    /* conservative closest-candidate search on the second tree: boxes relaxed by mu, so that no Moller-Trumbore-accepted triangle in range is missed */
    float best_t = INFINITY, best_u = 0.0f, best_v = 0.0f;
    int best_tri = -1, tie = 0;
    int stack[128];
    int sp = 0, node_index = 0;
    const float mu = P->mu;
    for (;;) {
        const GPUBVHNode* n = &P->nodes[node_index];
        float te, tx;
        const float lo[3] = { n->bbox_min.x, n->bbox_min.y, n->bbox_min.z }, hi[3] = { n->bbox_max.x, n->bbox_max.y, n->bbox_max.z };
        slab_interval(lo, hi, ray, t_min, &te, &tx);
        P->second_tree_nodes++;
        const float pad = mu * (fabsf(te) + fabsf(tx)) + 1e-30f;
        int enter = !(tx + pad < te - pad) && !(te - pad > (best_t < t_max ? best_t : t_max) + pad);        /* NaNs fall towards entering */
        if (enter) {
            if (n->tri_count > 0) {
                for (int i = 0; i < n->tri_count; ++i) {
                    const int tri = P->tri_indices[n->tri_offset + i];
                    if (P->never_hit[tri]) continue;
                    float t, u, v;
                    P->second_tree_tri_tests++;
                    if (!mt_accepts(s, tri, ray, t_min, &t, &u, &v)) continue;
                    if (t > t_max) continue;
                    if (t < best_t) { best_t = t; best_u = u; best_v = v; best_tri = tri; tie = 0; }
                    else if (t == best_t && tri != best_tri) tie = 1;
                }
            } else if (sp + 2 <= 128) {
                stack[sp++] = n->right;
                stack[sp++] = n->left;
            }
        }
        if (sp == 0) break;
        node_index = stack[--sp];
    }
    int flagged = 0;
    if (ray->dir.x == 0.0f || ray->dir.y == 0.0f || ray->dir.z == 0.0f) { flagged = 1; P->flagged_zero_dir++; }     /* 0 * inf in the slab arithmetic: not covered by the argument */
    if (best_tri >= 0) {
        if (tie) { flagged = 1; P->flagged_tie++; }
        float te, tx;
        slab_interval(P->leaf_box + 6 * (size_t)best_tri, P->leaf_box + 6 * (size_t)best_tri + 3, ray, t_min, &te, &tx);
        if (!(tx > te) || !(best_t > te)) { flagged = 1; P->flagged_leaf_box++; }
    }
    if (flagged) return;
    const int mine_hit = best_tri >= 0;
    int same = mine_hit == (ref_hit != 0);
    if (same && mine_hit) {
        uint32_t a[3], b[3];
        memcpy(&a[0], &best_t, 4); memcpy(&a[1], &best_u, 4); memcpy(&a[2], &best_v, 4);
        memcpy(&b[0], &ref_rec->t, 4); memcpy(&b[1], &ref_rec->u, 4); memcpy(&b[2], &ref_rec->v, 4);
        same = best_tri == ref_rec->tri_index && a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
    }
    if (!same) {
        if (!P->mismatched) { P->first_bad_tri_ref = ref_hit ? ref_rec->tri_index : -1; P->first_bad_tri_mine = best_tri; }
        P->mismatched++;
    }
}

/* Renders rows [y0, y1) with the reference walk on `scene`'s own (median) tree and checks every BVH query against the certified answer on the second tree.
 * out[0..9] = rays, reference hits, flagged: exact tie, flagged: leaf box, flagged: zero direction component, NOT FLAGGED AND DIFFERENT (must be 0),
 *             triangle tests on the second tree, nodes visited on the second tree, first differing triangle (reference, certified). */
int dsrt_sahprobe_render_rows(const GPUScene* scene, const GPUBVHNode* second_nodes, int num_second_nodes, const int* second_tri_indices, const uint8_t* never_hit,
                              const float* leaf_box, float mu, int W, int H, int y0, int y1, uint8_t* rgb8, uint64_t out[10], DsrtOracleCounters* counters) {
    memset(&g_probe, 0, sizeof g_probe);
    g_probe.nodes = second_nodes; g_probe.num_nodes = num_second_nodes; g_probe.tri_indices = second_tri_indices;
    g_probe.never_hit = never_hit; g_probe.leaf_box = leaf_box; g_probe.mu = mu;
    g_probe.first_bad_tri_ref = g_probe.first_bad_tri_mine = -2;
    const int rc = dsrt_oracle_render_rows(scene, W, H, y0, y1, rgb8, NULL, counters);
    out[0] = g_probe.rays; out[1] = g_probe.ref_hits; out[2] = g_probe.flagged_tie; out[3] = g_probe.flagged_leaf_box; out[4] = g_probe.flagged_zero_dir;
    out[5] = g_probe.mismatched; out[6] = g_probe.second_tree_tri_tests; out[7] = g_probe.second_tree_nodes;
    out[8] = (uint64_t)(int64_t)g_probe.first_bad_tri_ref; out[9] = (uint64_t)(int64_t)g_probe.first_bad_tri_mine;
    g_probe.nodes = NULL;
    return rc;
}
