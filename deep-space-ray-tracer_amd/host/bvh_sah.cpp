// bvh_sah.cpp -- a surface-area-heuristic BVH over the same triangles, in the same node format as bvh_median.cpp.
//
// NOT the reference's tree.  The reference splits every range at the centroid median (src/gpu_scene_builder.cpp:343-459),
// which is what bvh_median.cpp reproduces and what every parity statement about the reference is made on.  This builder is
// the "non-parity fast mode" of SURVEY.md section 8(f) n4: a binned-SAH tree makes the same rays visit far fewer nodes on
// a mesh of long thin members and large panels, but it changes which of two equal-distance hits is found last, and which
// rays graze past a box in float arithmetic -- so a frame rendered on it is a statistically equivalent image, not the same
// bytes.  (The kernel and the oracle both walk whatever tree the scene carries, so kernel-vs-oracle parity on this tree is
// still exact and tests/test_gpu_parity.py checks it.)
//
// Top-down, 32 centroid bins per axis, all three axes tried, leaf at <= 4 triangles (the reference's leaf size);
// nodes numbered in pre-order like the reference's, so the depth-first re-layout in device_api.hip sees the same shape.
#include <algorithm>
#include <cmath>
#include <exception>
#include <limits>
#include <thread>

#include "host_internal.hpp"

namespace {

struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b.lo[a]); hi[a] = fmaxf(hi[a], b.hi[a]); } }
    double half_area() const {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kMaxBins = 128;
static int g_bins = 32;                     // development override: DSRT_SAH_BINS (4..128)
static int g_sweep_below = 0;               // development override: DSRT_SAH_SWEEP -- ranges of at most this many triangles are split by an exact sweep over the sorted centroids

constexpr int kParallelMin = 1 << 15;       // ranges smaller than this are not worth a thread

struct SahBuilder {
    const std::vector<Box>& tri_box;
    const std::vector<float>& centroid;
    std::vector<int>& order;
    int kLeafMax = 4;                       // the reference's leaf size

    // as in bvh_median.cpp: the two halves of a large range are built on two threads into node vectors of their own and spliced in pre-order
    int build(int start, int end, int level, std::vector<GPUBVHNode>& nodes, int& height, int fork) {
        const int self = (int)nodes.size();
        nodes.emplace_back();
        if (level > height) height = level;
        Box box; box.clear();
        Box cbox; cbox.clear();
        for (int i = start; i < end; ++i) {
            box.grow(tri_box[order[i]]);
            for (int a = 0; a < 3; ++a) {
                const float c = centroid[3 * (size_t)order[i] + a];
                cbox.lo[a] = fminf(cbox.lo[a], c); cbox.hi[a] = fmaxf(cbox.hi[a], c);
            }
        }
        {
            GPUBVHNode& n = nodes[self];
            n.bbox_min = DsrtF3{box.lo[0], box.lo[1], box.lo[2]};
            n.bbox_max = DsrtF3{box.hi[0], box.hi[1], box.hi[2]};
            n.left = n.right = -1;
            n.tri_offset = start;
            n.tri_count = end - start;
        }
        const int count = end - start;
        if (count <= kLeafMax) return self;

        int sweep_mid = -1;
        if (g_sweep_below > 0 && count <= g_sweep_below) {
            // (development knob) small ranges: every one of the n - 1 splits of the centroids sorted along each axis, not just the bin boundaries
            std::vector<int> idx(order.begin() + start, order.begin() + end), best_idx;
            std::vector<double> right((size_t)count);
            double best = std::numeric_limits<double>::infinity();
            for (int a = 0; a < 3; ++a) {
                std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return centroid[3 * (size_t)x + a] < centroid[3 * (size_t)y + a]; });
                Box acc; acc.clear();
                for (int i = count - 1; i >= 1; --i) { acc.grow(tri_box[idx[(size_t)i]]); right[(size_t)i] = acc.half_area(); }
                acc.clear();
                for (int i = 0; i + 1 < count; ++i) {
                    acc.grow(tri_box[idx[(size_t)i]]);
                    const double cost = acc.half_area() * (i + 1) + right[(size_t)i + 1] * (count - i - 1);
                    if (cost < best) { best = cost; sweep_mid = start + i + 1; best_idx = idx; }
                }
            }
            if (sweep_mid > 0) std::copy(best_idx.begin(), best_idx.end(), order.begin() + start);
        }
        // best binned split over the three axes
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3 && sweep_mid < 0; ++a) {
            const float extent = cbox.hi[a] - cbox.lo[a];
            if (!(extent > 0.0f)) continue;
            const int kBins = g_bins;
            Box bin_box[kMaxBins]; int bin_n[kMaxBins];
            for (int b = 0; b < kBins; ++b) { bin_box[b].clear(); bin_n[b] = 0; }
            const float scale = (float)kBins / extent;
            for (int i = start; i < end; ++i) {
                int b = (int)((centroid[3 * (size_t)order[i] + a] - cbox.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bin_box[b].grow(tri_box[order[i]]);
                bin_n[b]++;
            }
            double right_area[kMaxBins]; int right_n[kMaxBins];
            Box acc; acc.clear(); int n = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bin_box[b]); n += bin_n[b]; right_area[b] = acc.half_area(); right_n[b] = n; }
            acc.clear(); n = 0;
            for (int b = 0; b + 1 < kBins; ++b) {
                acc.grow(bin_box[b]); n += bin_n[b];
                if (n == 0 || right_n[b + 1] == 0) continue;
                const double cost = acc.half_area() * n + right_area[b + 1] * right_n[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        int mid;
        if (sweep_mid > 0) {
            mid = sweep_mid;
        } else if (best_axis < 0) {
            // every centroid coincides: nothing to sort by; split the range in half by position (keeps leaves small)
            mid = (start + end) / 2;
        } else {
            const int kBins = g_bins;
            const float lo = cbox.lo[best_axis], scale = (float)kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
            const float* c = centroid.data();
            const int axis = best_axis, bin = best_bin;
            auto it = std::partition(order.begin() + start, order.begin() + end, [=](int t) {
                int b = (int)((c[3 * (size_t)t + axis] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= bin;
            });
            mid = (int)(it - order.begin());
            if (mid == start || mid == end) mid = (start + end) / 2;      // cannot happen with n > 0 on both sides; belt and braces
        }
        nodes[self].tri_offset = 0;
        nodes[self].tri_count = 0;
        if (fork > 0 && end - start >= kParallelMin) {
            std::vector<GPUBVHNode> left, right;
            left.reserve((size_t)(mid - start) * 2 / 3 + 64); right.reserve((size_t)(end - mid) * 2 / 3 + 64);
            int hl = 0, hr = 0;
            std::exception_ptr failed;
            std::thread worker([&]() { try { build(start, mid, level + 1, left, hl, fork - 1); } catch (...) { failed = std::current_exception(); } });
            try { build(mid, end, level + 1, right, hr, fork - 1); } catch (...) { worker.join(); throw; }
            worker.join();
            if (failed) std::rethrow_exception(failed);
            auto splice = [&](const std::vector<GPUBVHNode>& sub) {
                const int base = (int)nodes.size();
                for (GPUBVHNode n : sub) {
                    if (n.tri_count == 0 && n.left >= 0) { n.left += base; n.right += base; }
                    nodes.push_back(n);
                }
                return base;
            };
            const int l = splice(left);
            nodes[self].left = l;
            const int r = splice(right);
            nodes[self].right = r;
            height = std::max(height, std::max(hl, hr));
            return self;
        }
        const int l = build(start, mid, level + 1, nodes, height, 0);
        nodes[self].left = l;
        const int r = build(mid, end, level + 1, nodes, height, 0);
        nodes[self].right = r;
        return self;
    }
};

}  // namespace

namespace dsrt {

// The builder on plain arrays (host_internal.hpp).  `pad_all` > 0 widens EVERY triangle box by that much on every side (the certified second tree of
// device_api.hip needs boxes that no rounding of the slab arithmetic can make a hit triangle miss); `skip` (may be null) marks triangles to leave out.
int build_sah_tree(const GPUTriangle* tris, size_t n, float pad_all, const uint8_t* skip, std::vector<GPUBVHNode>& nodes, std::vector<int>& order, int& height_out, int leaf_max) {
    nodes.clear();
    order.clear();
    height_out = 0;
    if (n == 0) return DSRT_OK;
    if (n > (size_t)1 << 28) { dsrt::set_error("more than 2^28 triangles"); return DSRT_ERR_INVALID; }
    if (const char* e = std::getenv("DSRT_SAH_BINS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 4 && v <= kMaxBins) g_bins = (int)v; }
    if (const char* e = std::getenv("DSRT_SAH_SWEEP")) { const long v = std::strtol(e, nullptr, 10); if (v >= 0 && v <= 4096) g_sweep_below = (int)v; }
    std::vector<Box> tri_box(n);
    std::vector<float> centroid(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const GPUTriangle& t = tris[i];
        const float v[3][3] = {{t.v0.x, t.v1.x, t.v2.x}, {t.v0.y, t.v1.y, t.v2.y}, {t.v0.z, t.v1.z, t.v2.z}};
        for (int a = 0; a < 3; ++a) {
            tri_box[i].lo[a] = fminf(fminf(v[a][0], v[a][1]), v[a][2]);
            tri_box[i].hi[a] = fmaxf(fmaxf(v[a][0], v[a][1]), v[a][2]);
            centroid[3 * i + a] = 0.5f * (tri_box[i].lo[a] + tri_box[i].hi[a]);
        }
    }
    // A triangle that lies in an axis plane has a box of zero thickness, and a leaf of such triangles a box the reference's slab test can
    // never hit (t_max <= t_min holds with equality, src/gpu_render.cu:312) -- in the reference's own tree such faces survive or vanish by
    // the accident of who shares their leaf.  A builder that is free to choose must not make geometry vanish: flat triangle boxes are
    // widened by 2^-12 of the scene's extent on that axis, enough to survive the rounding of (box - origin) for any camera within 2,000
    // scene extents.  The median builder keeps the reference's boxes, accidents included.
    {
        float lo[3] = {tri_box[0].lo[0], tri_box[0].lo[1], tri_box[0].lo[2]}, hi[3] = {tri_box[0].hi[0], tri_box[0].hi[1], tri_box[0].hi[2]};
        for (size_t i = 1; i < n; ++i)
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tri_box[i].lo[a]); hi[a] = fmaxf(hi[a], tri_box[i].hi[a]); }
        const float pad = dsrt::flat_box_pad(fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]));
        for (size_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                if (pad_all > 0.0f) { tri_box[i].lo[a] -= pad_all; tri_box[i].hi[a] += pad_all; }
                else if (tri_box[i].lo[a] == tri_box[i].hi[a]) { tri_box[i].lo[a] -= pad; tri_box[i].hi[a] += pad; }
            }
    }
    order.reserve(n);
    for (size_t i = 0; i < n; ++i) if (!skip || !skip[i]) order.push_back((int)i);
    if (order.empty()) return DSRT_OK;
    nodes.reserve(order.size());
    SahBuilder b{tri_box, centroid, order, leaf_max >= 1 && leaf_max <= 7 ? leaf_max : 4};
    int height = 0, fork = 0;
    for (unsigned t = dsrt::builder_threads(); t > 1 && fork < 4; t >>= 1) ++fork;
    b.build(0, (int)order.size(), 1, nodes, height, fork);
    height_out = height;
    if (height - 1 > 64) {
        dsrt::set_error("SAH BVH needs a traversal stack deeper than 64 entries");
        return DSRT_ERR_BVH_DEPTH;
    }
    return DSRT_OK;
}

int prepare_second_tree(const GPUScene& h, int leaf_max, SecondTree& out) {
    const int N = h.num_triangles, M = h.num_bvh_nodes;
    out = SecondTree{};
    if (N <= 0 || M <= 0 || !h.bvh_nodes || !h.tri_indices || !h.triangles) return DSRT_OK;
    out.unreachable.assign((size_t)N, 0);
    out.leaf_box.assign((size_t)N * 6, 0.0f);
    {
        struct Walk { int node; bool dead; };
        std::vector<Walk> todo{{0, false}};
        std::vector<char> seen((size_t)M, 0);
        while (!todo.empty()) {
            Walk w = todo.back(); todo.pop_back();
            if (w.node < 0 || w.node >= M || seen[(size_t)w.node]) { set_error("BVH is not a tree"); return DSRT_ERR_INVALID; }
            seen[(size_t)w.node] = 1;
            const GPUBVHNode& n = h.bvh_nodes[w.node];
            // (a box with a NaN bound passes bbox_hit's comparisons on that axis whatever the ray: no statement about reachability is safe -- such a tree gets no second tree)
            if (std::isnan(n.bbox_min.x + n.bbox_min.y + n.bbox_min.z + n.bbox_max.x + n.bbox_max.y + n.bbox_max.z)) { out = SecondTree{}; out.origins_near = false; return DSRT_OK; }
            // zero thickness -- or an inverted box, which fails the same comparison -- on any axis: `t_max <= t_min` (:312) holds for every ray
            const bool dead = w.dead || !(n.bbox_min.x < n.bbox_max.x) || !(n.bbox_min.y < n.bbox_max.y) || !(n.bbox_min.z < n.bbox_max.z);
            if (n.tri_count > 0) {
                if (n.tri_offset < 0 || (long long)n.tri_offset + n.tri_count > N) { set_error("BVH leaf range out of bounds"); return DSRT_ERR_INVALID; }
                for (int i = 0; i < n.tri_count; ++i) {
                    const int t = h.tri_indices[n.tri_offset + i];
                    if (t < 0 || t >= N) { set_error("tri_indices entry out of range"); return DSRT_ERR_INVALID; }
                    float* b = &out.leaf_box[6 * (size_t)t];
                    b[0] = n.bbox_min.x; b[1] = n.bbox_min.y; b[2] = n.bbox_min.z; b[3] = n.bbox_max.x; b[4] = n.bbox_max.y; b[5] = n.bbox_max.z;
                    if (dead) out.unreachable[(size_t)t] = 1;
                }
            } else { todo.push_back({n.left, dead}); todo.push_back({n.right, dead}); }
        }
    }
    const GPUBVHNode& root = h.bvh_nodes[0];
    out.extent = fmaxf(fmaxf(root.bbox_max.x - root.bbox_min.x, root.bbox_max.y - root.bbox_min.y), root.bbox_max.z - root.bbox_min.z);
    out.pad = out.extent > 0.0f ? out.extent * (1.0f / 65536.0f) : 1.0e-6f;
    const int rc = build_sah_tree(h.triangles, (size_t)N, out.pad, out.unreachable.data(), out.nodes, out.order, out.height, leaf_max);
    if (rc != DSRT_OK) return rc;
    // Rays also start on SPHERES (bounces, shadow rays): one whose surface reaches beyond 30 extents of the mesh puts origins where the widening no longer covers the
    // rounding of (box - origin) -- such a scene keeps the reference tree only.
    for (int i = 0; i < h.num_spheres && out.origins_near && h.spheres; ++i) {
        const GPUSphere& sp = h.spheres[i];
        const double dx = (double)sp.center.x - 0.5 * ((double)root.bbox_min.x + root.bbox_max.x), dy = (double)sp.center.y - 0.5 * ((double)root.bbox_min.y + root.bbox_max.y),
                     dz = (double)sp.center.z - 0.5 * ((double)root.bbox_min.z + root.bbox_max.z);
        out.origins_near = std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)sp.radius) <= 30.0 * (double)out.extent;
    }
    return DSRT_OK;
}

}  // namespace dsrt

// Test hook (no GPU): what dsrt_scene_upload would prepare for the certified second tree of this host scene.  counts = {triangles, unreachable on the reference tree,
// triangles in the second tree, its nodes, its height, 1 if no sphere is too far}; arrays (any may be null): unreachable[triangles], leaf_box[6 * triangles], nodes, order.
extern "C" int dsrt_host_scene_second_tree_probe(const DsrtHostScene* hs, int counts[6], float* pad, uint8_t* unreachable, float* leaf_box, GPUBVHNode* nodes, int max_nodes, int* order,
                                                 int max_order) {
    return dsrt::guarded("dsrt_host_scene_second_tree_probe", [&]() -> int {
    if (!hs || !counts) { dsrt::set_error("dsrt_host_scene_second_tree_probe: null argument"); return DSRT_ERR_INVALID; }
    GPUScene view;
    int rc = dsrt_host_scene_view(hs, &view);
    if (rc) return rc;
    dsrt::SecondTree st;
    if ((rc = dsrt::prepare_second_tree(view, 4, st))) return rc;
    int dead = 0;
    for (uint8_t u : st.unreachable) dead += u;
    counts[0] = view.num_triangles; counts[1] = dead; counts[2] = (int)st.order.size(); counts[3] = (int)st.nodes.size(); counts[4] = st.height; counts[5] = st.origins_near ? 1 : 0;
    if (pad) *pad = st.pad;
    if (unreachable) std::copy(st.unreachable.begin(), st.unreachable.end(), unreachable);
    if (leaf_box) std::copy(st.leaf_box.begin(), st.leaf_box.end(), leaf_box);
    if (nodes) { if ((int)st.nodes.size() > max_nodes) { dsrt::set_error("node buffer too small"); return DSRT_ERR_INVALID; } std::copy(st.nodes.begin(), st.nodes.end(), nodes); }
    if (order) { if ((int)st.order.size() > max_order) { dsrt::set_error("order buffer too small"); return DSRT_ERR_INVALID; } std::copy(st.order.begin(), st.order.end(), order); }
    return DSRT_OK;
    });
}

extern "C" int dsrt_host_scene_build_bvh_sah(DsrtHostScene* hs) {
    return dsrt::guarded("dsrt_host_scene_build_bvh_sah", [&]() -> int {
    if (!hs) { dsrt::set_error("dsrt_host_scene_build_bvh_sah: null scene"); return DSRT_ERR_INVALID; }
    hs->tri_indices.clear();
    hs->nodes.clear();
    hs->bvh_height = 0;
    const int rc = dsrt::build_sah_tree(hs->tris.data(), hs->tris.size(), 0.0f, nullptr, hs->nodes, hs->tri_indices, hs->bvh_height, 4);
    if (rc != DSRT_OK) return rc;
    hs->bvh_valid = true;
    return DSRT_OK;
    });
}
