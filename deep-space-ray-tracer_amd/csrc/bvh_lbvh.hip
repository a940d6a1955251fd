// bvh_lbvh.hip -- BVH construction ON THE GPU (include/dsrt.h: dsrt_host_scene_build_bvh_gpu), the "GPU-side BVH build" of SURVEY.md 8(f) n4.
//
// NOT the reference's tree.  The reference builds a median-split BVH on the host, per frame (src/gpu_scene_builder.cpp:343-459; 1.2 s at
// 1 M triangles on one core here).  This is a linear BVH (Morton order + the radix-tree construction of Karras 2012) in the SAME node
// format (GPUBVHNode, leaf <= 4 triangles, tri_indices indirection), so everything downstream -- validation, re-layout, the render
// kernel, the oracle -- takes it unchanged.  Like the SAH tree it is a non-parity fast mode: the image is statistically the reference's,
// not its bytes (equal-distance ties and float grazing cases depend on the boxes: DESIGN.md section 8).
//
// Steps, all on the device (hipCUB for the sort and the scans, everything else plain kernels, HBM-bound and tiny next to a render):
//   1. per triangle: bounds + centroid of the bounds; scene bounds by atomic min / max on order-preserving integer images of the floats
//   2. 63-bit Morton code of the centroid inside the scene bounds (21 bits per axis); radix sort of (code, triangle index)
//   3. radix tree over the N sorted triangles (equal codes are told apart by their position, Karras section 4): internal node i covers the
//      range found from the longest-common-prefix function, split where the prefix of the range's ends first differs (one thread per node)
//   4. boxes bottom-up: every triangle walks to the root, the second thread to arrive at a node (atomic counter) merges its children
//   5. collapse: a subtree of at most 4 triangles becomes one leaf (the reference's leaf size); what remains is numbered by two prefix
//      sums -- kept internal nodes first, in radix-tree order, so the root is node 0, then the leaves -- and written as GPUBVHNode.
//      (Round 2's first version cut the Morton order into runs of 4 whatever lay between them and used 30-bit codes: a run that straddles
//      a gap of the hierarchy makes a leaf box as big as the gap, and the near frame took 1850 ms on that tree.)
// The result is copied into the host scene's vectors (a 1 M-triangle tree is 20 MB), because the scene's home is the host
// (DsrtHostScene) and dsrt_scene_upload re-lays it out for traversal anyway.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dsrt.h"
#include "../host/host_internal.hpp"

using dsrt::set_error;

namespace {

bool ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

struct Bounds { float lo[3], hi[3]; };

// float <-> unsigned with the same ordering (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t ordered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float unordered(uint32_t u) { const uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &v, 4); return f; }

__global__ void tri_bounds_kernel(const GPUTriangle* __restrict__ tris, int n, Bounds* __restrict__ tb, uint32_t* __restrict__ scene6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < n) {
        const GPUTriangle& t = tris[i];
        const float v[3][3] = {{t.v0.x, t.v1.x, t.v2.x}, {t.v0.y, t.v1.y, t.v2.y}, {t.v0.z, t.v1.z, t.v2.z}};
        Bounds b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = lo[a] = fminf(fminf(v[a][0], v[a][1]), v[a][2]);
            b.hi[a] = hi[a] = fmaxf(fmaxf(v[a][0], v[a][1]), v[a][2]);
        }
        tb[i] = b;
    }
    for (int a = 0; a < 3; ++a) {                                     // one atomic per wave and axis
        float l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) { l = fminf(l, __shfl_down(l, off, 64)); h = fmaxf(h, __shfl_down(h, off, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&scene6[a], ordered(l)); atomicMax(&scene6[3 + a], ordered(h)); }
    }
}

__device__ __forceinline__ unsigned long long spread3(unsigned long long v) {      // 21 bits -> every third bit of 63
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x001F00000000FFFFull;
    v = (v | v << 16) & 0x001F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ void morton_kernel(Bounds* __restrict__ tb, int n, const uint32_t* __restrict__ scene6, unsigned long long* __restrict__ code, uint32_t* __restrict__ index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long q[3];
    float extent = 0.0f;
    for (int a = 0; a < 3; ++a) extent = fmaxf(extent, unordered(scene6[3 + a]) - unordered(scene6[a]));
    const float pad = extent > 0.0f ? extent * (1.0f / 4096.0f) : 1.0e-6f;      // host_internal.hpp flat_box_pad: see bvh_sah.cpp for why
    for (int a = 0; a < 3; ++a) {
        const float lo = unordered(scene6[a]), hi = unordered(scene6[3 + a]);
        const float c = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
        q[a] = (unsigned long long)u;
    }
    code[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    index[i] = (uint32_t)i;
    Bounds b = tb[i];
    bool flat = false;
    for (int a = 0; a < 3; ++a)
        if (b.lo[a] == b.hi[a]) { b.lo[a] -= pad; b.hi[a] += pad; flat = true; }     // no leaf of zero thickness (the slab test never hits one)
    if (flat) tb[i] = b;
}

// length of the common prefix of keys i and j; equal codes continue into their positions, which makes every key unique
__device__ __forceinline__ int lcp(const unsigned long long* key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = key[i], b = key[j];
    return a != b ? __clzll((long long)(a ^ b)) : 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", section 4: internal node i of a binary
// radix tree over n sorted keys.  child refs: >= 0 internal node, < 0 triangle ~ref (position in the sorted order).
__global__ void radix_tree_kernel(const unsigned long long* __restrict__ key, int n, int2* __restrict__ children, int2* __restrict__ range,
                                  int* __restrict__ parent_internal, int* __restrict__ parent_tri) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = lcp(key, n, i, i + 1) - lcp(key, n, i, i - 1) > 0 ? 1 : -1;
    const int floor_lcp = lcp(key, n, i, i - d);
    int reach = 2;
    while (lcp(key, n, i, i + reach * d) > floor_lcp) reach <<= 1;
    int len = 0;
    for (int t = reach >> 1; t > 0; t >>= 1)
        if (lcp(key, n, i, i + (len + t) * d) > floor_lcp) len += t;
    const int j = i + len * d;
    const int node_lcp = lcp(key, n, i, j);
    int s = 0;
    for (int div = 2, t = (len + 1) / 2; ; div <<= 1, t = (len + div - 1) / div) {
        if (lcp(key, n, i, i + (s + t) * d) > node_lcp) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? ~gamma : gamma;
    const int right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(left, right);
    range[i] = make_int2(lo, hi);
    if (left < 0) parent_tri[~left] = i; else parent_internal[left] = i;
    if (right < 0) parent_tri[~right] = i; else parent_internal[right] = i;
}

__device__ __forceinline__ Bounds merged(const Bounds& l, const Bounds& r) {
    Bounds m;
    for (int a = 0; a < 3; ++a) { m.lo[a] = fminf(l.lo[a], r.lo[a]); m.hi[a] = fmaxf(l.hi[a], r.hi[a]); }
    return m;
}

__global__ void fit_kernel(const Bounds* __restrict__ tb, const uint32_t* __restrict__ sorted_index, int n, const int2* __restrict__ children,
                           const int* __restrict__ parent_internal, const int* __restrict__ parent_tri, int* __restrict__ arrived, Bounds* ibox) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || n == 1) return;
    int node = parent_tri[j];
    while (node >= 0) {
        // Hand-off between workgroups on different CUs / XCDs: the first child to arrive stops; the second finds both boxes visible --
        // agent-scope release after a node's store (with the explicit wait the compiler may otherwise drop, MI355X_MICROARCH.md
        // "Compiler hazard"), the counter, agent-scope acquire before the loads.
        if (atomicAdd(&arrived[node], 1) == 0) return;
        __threadfence();
        const int2 ch = children[node];
        const Bounds l = ch.x < 0 ? tb[sorted_index[~ch.x]] : ibox[ch.x], r = ch.y < 0 ? tb[sorted_index[~ch.y]] : ibox[ch.y];
        ibox[node] = merged(l, r);
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        node = node == 0 ? -1 : parent_internal[node];
    }
}

constexpr int kLeafMax = 4;                                           // the reference's leaf size (src/gpu_scene_builder.cpp:399)

// item k < n - 1: internal node k; item k >= n - 1: triangle k - (n - 1).  keep: internal nodes that stay internal; leafroot: roots of
// the subtrees that become leaves (an internal node of at most kLeafMax triangles, or a single triangle, under a kept parent)
__global__ void collapse_flags_kernel(int n, const int2* __restrict__ range, const int* __restrict__ parent_internal, const int* __restrict__ parent_tri,
                                      int* __restrict__ keep, int* __restrict__ leafroot) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * n - 1) return;
    auto big = [&](int node) { return range[node].y - range[node].x + 1 > kLeafMax; };
    if (k < n - 1) {
        const bool mine = big(k);
        keep[k] = mine ? 1 : 0;
        leafroot[k] = (!mine && k != 0 && big(parent_internal[k])) ? 1 : 0;
    } else {
        leafroot[k] = big(parent_tri[k - (n - 1)]) ? 1 : 0;
    }
}

__global__ void emit_kernel(int n, int kept, const int2* __restrict__ children, const int2* __restrict__ range, const int* __restrict__ keep,
                            const int* __restrict__ leafroot, const int* __restrict__ kidx, const int* __restrict__ lidx, const Bounds* __restrict__ ibox,
                            const Bounds* __restrict__ tb, const uint32_t* __restrict__ sorted_index, GPUBVHNode* __restrict__ nodes) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * n - 1) return;
    GPUBVHNode out;
    int at;
    Bounds b;
    if (k < n - 1 && keep[k]) {
        const int2 ch = children[k];
        auto ref = [&](int c) { const int item = c < 0 ? (n - 1) + ~c : c; return (c >= 0 && keep[c]) ? kidx[c] : kept + lidx[item]; };
        b = ibox[k];
        out.left = ref(ch.x); out.right = ref(ch.y); out.tri_offset = 0; out.tri_count = 0;
        at = kidx[k];
    } else if (leafroot[k]) {
        if (k < n - 1) { b = ibox[k]; out.tri_offset = range[k].x; out.tri_count = range[k].y - range[k].x + 1; }
        else { b = tb[sorted_index[k - (n - 1)]]; out.tri_offset = k - (n - 1); out.tri_count = 1; }
        out.left = out.right = -1;
        at = kept + lidx[k];
    } else return;
    out.bbox_min = DsrtF3{b.lo[0], b.lo[1], b.lo[2]}; out.bbox_max = DsrtF3{b.hi[0], b.hi[1], b.hi[2]};
    nodes[at] = out;
}

template <typename T>
struct Dev {
    T* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return ok(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)), "hipMalloc"); }
};

}  // namespace

extern "C" int dsrt_host_scene_build_bvh_gpu(DsrtHostScene* hs, int device, float* build_ms, float* total_ms) {
    if (!hs) { set_error("dsrt_host_scene_build_bvh_gpu: null scene"); return DSRT_ERR_INVALID; }
    return dsrt::guarded("dsrt_host_scene_build_bvh_gpu", [&]() -> int {
        const auto t_begin = std::chrono::steady_clock::now();
        hs->tri_indices.clear();
        hs->nodes.clear();
        hs->bvh_height = 0;
        hs->bvh_valid = false;
        const size_t n = hs->tris.size();
        if (n == 0) { hs->bvh_valid = true; if (build_ms) *build_ms = 0; if (total_ms) *total_ms = 0; return DSRT_OK; }
        if (n > (size_t)1 << 28) { set_error("more than 2^28 triangles"); return DSRT_ERR_INVALID; }
        if (!ok(hipSetDevice(device), "hipSetDevice")) return DSRT_ERR_HIP;
        const int N = (int)n, items = 2 * N - 1;

        Dev<GPUTriangle> d_tris; Dev<Bounds> d_tb, d_ibox; Dev<uint32_t> d_scene, d_idx, d_idx2; Dev<unsigned long long> d_code, d_code2;
        Dev<int2> d_children, d_range; Dev<int> d_pi, d_pt, d_arrived, d_keep, d_leafroot, d_kidx, d_lidx; Dev<GPUBVHNode> d_nodes; Dev<unsigned char> d_tmp;
        if (!d_tris.alloc(n) || !d_tb.alloc(n) || !d_ibox.alloc(n) || !d_scene.alloc(6) || !d_code.alloc(n) || !d_code2.alloc(n) || !d_idx.alloc(n) || !d_idx2.alloc(n) ||
            !d_children.alloc(n) || !d_range.alloc(n) || !d_pi.alloc(n) || !d_pt.alloc(n) || !d_arrived.alloc(n) || !d_keep.alloc(n) || !d_kidx.alloc(n) ||
            !d_leafroot.alloc((size_t)items) || !d_lidx.alloc((size_t)items) || !d_nodes.alloc((size_t)items)) return DSRT_ERR_HIP;
        size_t sort_bytes = 0, scan_bytes = 0;
        if (!ok(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_code.p, d_code2.p, d_idx.p, d_idx2.p, N, 0, 63), "hipcub size query") ||
            !ok(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_leafroot.p, d_lidx.p, items), "hipcub size query")) return DSRT_ERR_HIP;
        const size_t tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
        if (!d_tmp.alloc(tmp_bytes)) return DSRT_ERR_HIP;
        if (!ok(hipMemcpy(d_tris.p, hs->tris.data(), n * sizeof(GPUTriangle), hipMemcpyHostToDevice), "hipMemcpy triangles")) return DSRT_ERR_HIP;

        hipEvent_t e0, e1;
        if (!ok(hipEventCreate(&e0), "hipEventCreate") || !ok(hipEventCreate(&e1), "hipEventCreate")) return DSRT_ERR_HIP;
        const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        bool good = ok(hipMemcpy(d_scene.p, init, sizeof init, hipMemcpyHostToDevice), "hipMemcpy");
        good = good && ok(hipEventRecord(e0, nullptr), "hipEventRecord");
        const unsigned bt = (unsigned)((n + 255) / 256), bi = (unsigned)((items + 255) / 256);
        int kept = 0, leaf_count = 0;
        if (good) {
            hipLaunchKernelGGL(tri_bounds_kernel, dim3(bt), dim3(256), 0, nullptr, d_tris.p, N, d_tb.p, d_scene.p);
            hipLaunchKernelGGL(morton_kernel, dim3(bt), dim3(256), 0, nullptr, d_tb.p, N, d_scene.p, d_code.p, d_idx.p);
            size_t bytes = tmp_bytes;
            good = ok(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, bytes, d_code.p, d_code2.p, d_idx.p, d_idx2.p, N, 0, 63), "hipcub sort");
        }
        if (good && N > kLeafMax) {
            good = ok(hipMemsetAsync(d_arrived.p, 0, n * sizeof(int), nullptr), "hipMemsetAsync");
            if (good) {
                hipLaunchKernelGGL(radix_tree_kernel, dim3(bt), dim3(256), 0, nullptr, d_code2.p, N, d_children.p, d_range.p, d_pi.p, d_pt.p);
                hipLaunchKernelGGL(fit_kernel, dim3(bt), dim3(256), 0, nullptr, d_tb.p, d_idx2.p, N, d_children.p, d_pi.p, d_pt.p, d_arrived.p, d_ibox.p);
                hipLaunchKernelGGL(collapse_flags_kernel, dim3(bi), dim3(256), 0, nullptr, N, d_range.p, d_pi.p, d_pt.p, d_keep.p, d_leafroot.p);
                size_t bytes = tmp_bytes;
                good = ok(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, bytes, d_keep.p, d_kidx.p, N - 1), "hipcub scan");
                bytes = tmp_bytes;
                good = good && ok(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, bytes, d_leafroot.p, d_lidx.p, items), "hipcub scan");
            }
            int last[4] = {0, 0, 0, 0};                               // totals = last exclusive sum + last flag
            good = good && ok(hipMemcpy(&last[0], d_kidx.p + (N - 2), sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy") &&
                   ok(hipMemcpy(&last[1], d_keep.p + (N - 2), sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy") &&
                   ok(hipMemcpy(&last[2], d_lidx.p + (items - 1), sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy") &&
                   ok(hipMemcpy(&last[3], d_leafroot.p + (items - 1), sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy");
            kept = last[0] + last[1]; leaf_count = last[2] + last[3];
            if (good)
                hipLaunchKernelGGL(emit_kernel, dim3(bi), dim3(256), 0, nullptr, N, kept, d_children.p, d_range.p, d_keep.p, d_leafroot.p, d_kidx.p, d_lidx.p,
                                   d_ibox.p, d_tb.p, d_idx2.p, d_nodes.p);
        }
        good = good && ok(hipGetLastError(), "LBVH kernels") && ok(hipEventRecord(e1, nullptr), "hipEventRecord") && ok(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.0f;
        if (good) good = ok(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!good) return DSRT_ERR_HIP;

        std::vector<uint32_t> order(n);
        if (!ok(hipMemcpy(order.data(), d_idx2.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy order")) return DSRT_ERR_HIP;
        hs->tri_indices.assign(order.begin(), order.end());
        int total;
        if (N <= kLeafMax) {                                          // the whole mesh is one leaf (it sits at index 0)
            total = 1;
            GPUBVHNode leaf;
            float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (const GPUTriangle& t : hs->tris) {
                const float v[3][3] = {{t.v0.x, t.v1.x, t.v2.x}, {t.v0.y, t.v1.y, t.v2.y}, {t.v0.z, t.v1.z, t.v2.z}};
                for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) { lo[a] = std::fmin(lo[a], v[a][c]); hi[a] = std::fmax(hi[a], v[a][c]); }
            }
            leaf.bbox_min = DsrtF3{lo[0], lo[1], lo[2]}; leaf.bbox_max = DsrtF3{hi[0], hi[1], hi[2]};
            leaf.left = leaf.right = -1; leaf.tri_offset = 0; leaf.tri_count = N;
            hs->nodes.assign(1, leaf);
        } else {
            total = kept + leaf_count;
            if (kept < 1 || leaf_count < 2 || total > items) { set_error("LBVH collapse produced an impossible node count"); return DSRT_ERR_INVALID; }
            hs->nodes.resize((size_t)total);
            if (!ok(hipMemcpy(hs->nodes.data(), d_nodes.p, (size_t)total * sizeof(GPUBVHNode), hipMemcpyDeviceToHost), "hipMemcpy nodes")) return DSRT_ERR_HIP;
        }
        // height = levels on the longest root-to-leaf path (the traversal stack needs height - 1 entries): one pass over the nodes
        std::vector<int> level((size_t)total, 0);
        int height = 1;
        level[0] = 1;
        std::vector<int> todo{0};
        while (!todo.empty()) {
            const int v = todo.back(); todo.pop_back();
            const GPUBVHNode& nd = hs->nodes[(size_t)v];
            if (level[(size_t)v] > height) height = level[(size_t)v];
            if (nd.tri_count > 0) continue;
            if (nd.left < 0 || nd.left >= total || nd.right < 0 || nd.right >= total) { set_error("LBVH build produced a bad child index"); return DSRT_ERR_INVALID; }
            level[(size_t)nd.left] = level[(size_t)nd.right] = level[(size_t)v] + 1;
            todo.push_back(nd.left); todo.push_back(nd.right);
        }
        hs->bvh_height = height;
        hs->bvh_valid = true;
        if (build_ms) *build_ms = ms;
        if (total_ms) *total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        if (height - 1 > 64) { set_error("LBVH needs a traversal stack deeper than 64 entries"); return DSRT_ERR_BVH_DEPTH; }
        return DSRT_OK;
    });
}
