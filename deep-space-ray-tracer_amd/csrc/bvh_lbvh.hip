// bvh_lbvh.hip -- BVH construction ON THE GPU (include/dsrt.h: dsrt_host_scene_build_bvh_gpu), the "GPU-side BVH build" of SURVEY.md 8(f) n4.
//
// NOT the reference's tree.  The reference builds a median-split BVH on the host, per frame (src/gpu_scene_builder.cpp:343-459; 1.2 s at
// 1 M triangles on one core here).  This is a linear BVH (Morton order + the radix-tree construction of Karras 2012) in the SAME node
// format (GPUBVHNode, leaf <= 4 triangles, tri_indices indirection), so everything downstream -- validation, re-layout, the render
// kernel, the oracle -- takes it unchanged.  Like the SAH tree it is a non-parity fast mode: the image is statistically the reference's,
// not its bytes (equal-distance ties and float grazing cases depend on the boxes: DESIGN.md section 8).
//
// Steps, all on the device (hipCUB for the sort, everything else plain kernels, HBM-bound and tiny next to a render):
//   1. per triangle: bounds + centroid of the bounds; scene bounds by atomic min / max on order-preserving integer images of the floats
//   2. 30-bit Morton code of the centroid inside the scene bounds; radix sort of (code, triangle index)
//   3. leaves = runs of 4 consecutive triangles in Morton order; leaf key = (code of its first triangle) << 32 | leaf number (unique)
//   4. internal nodes: node i covers the leaf range found from the longest-common-prefix function of the keys, split where the
//      prefix of the range's ends first differs (one thread per node, no dependencies)
//   5. boxes bottom-up: every leaf walks to the root, the second thread to arrive at a node (atomic counter) merges its children
//   6. nodes written as GPUBVHNode: internal node i at index i (root = 0), leaf j at index (leaves - 1) + j
// The result is copied into the host scene's vectors (a 1 M-triangle tree is 20 MB), because the scene's home is the host
// (DsrtHostScene) and dsrt_scene_upload re-lays it out for traversal anyway.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
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

__device__ __forceinline__ uint32_t spread3(uint32_t v) {            // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void morton_kernel(const Bounds* __restrict__ tb, int n, const uint32_t* __restrict__ scene6, uint32_t* __restrict__ code, uint32_t* __restrict__ index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = unordered(scene6[a]), hi = unordered(scene6[3 + a]);
        const float c = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)u;
    }
    code[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    index[i] = (uint32_t)i;
}

__global__ void leaf_keys_kernel(const uint32_t* __restrict__ sorted_code, int n, int leaves, unsigned long long* __restrict__ key) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < leaves) key[j] = ((unsigned long long)sorted_code[(size_t)j * 4] << 32) | (unsigned long long)(uint32_t)j;
}

__device__ __forceinline__ int lcp(const unsigned long long* key, int leaves, int i, int j) {
    if (j < 0 || j >= leaves) return -1;
    return __clzll((long long)(key[i] ^ key[j]));                     // keys are unique (the leaf number is in the low word)
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", section 4: internal node i of a binary
// radix tree over sorted unique keys.  child refs: >= 0 internal node, < 0 leaf ~ref.
__global__ void radix_tree_kernel(const unsigned long long* __restrict__ key, int leaves, int2* __restrict__ children, int* __restrict__ parent_internal,
                                  int* __restrict__ parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= leaves - 1) return;
    const int d = lcp(key, leaves, i, i + 1) - lcp(key, leaves, i, i - 1) > 0 ? 1 : -1;
    const int floor_lcp = lcp(key, leaves, i, i - d);
    int reach = 2;
    while (lcp(key, leaves, i, i + reach * d) > floor_lcp) reach <<= 1;
    int len = 0;
    for (int t = reach >> 1; t > 0; t >>= 1)
        if (lcp(key, leaves, i, i + (len + t) * d) > floor_lcp) len += t;
    const int j = i + len * d;
    const int node_lcp = lcp(key, leaves, i, j);
    int s = 0;
    for (int div = 2, t = (len + 1) / 2; ; div <<= 1, t = (len + div - 1) / div) {
        if (lcp(key, leaves, i, i + (s + t) * d) > node_lcp) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? ~gamma : gamma;
    const int right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(left, right);
    if (left < 0) parent_leaf[~left] = i; else parent_internal[left] = i;
    if (right < 0) parent_leaf[~right] = i; else parent_internal[right] = i;
}

__global__ void fit_kernel(const Bounds* __restrict__ tb, const uint32_t* __restrict__ sorted_index, int n, int leaves, const int2* __restrict__ children,
                           const int* __restrict__ parent_internal, const int* __restrict__ parent_leaf, int* __restrict__ arrived,
                           GPUBVHNode* nodes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= leaves) return;
    const int first = j * 4, count = (n - first) < 4 ? (n - first) : 4;
    Bounds b = tb[sorted_index[first]];
    for (int k = 1; k < count; ++k) {
        const Bounds o = tb[sorted_index[first + k]];
        for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], o.lo[a]); b.hi[a] = fmaxf(b.hi[a], o.hi[a]); }
    }
    GPUBVHNode leaf;
    leaf.bbox_min = DsrtF3{b.lo[0], b.lo[1], b.lo[2]}; leaf.bbox_max = DsrtF3{b.hi[0], b.hi[1], b.hi[2]};
    leaf.left = leaf.right = -1; leaf.tri_offset = first; leaf.tri_count = count;
    nodes[(leaves - 1) + j] = leaf;
    if (leaves == 1) return;
    // Hand-off between workgroups on different CUs / XCDs: agent-scope release after the stores (with the explicit wait the compiler may
    // otherwise drop, MI355X_MICROARCH.md "Compiler hazard"), the counter, agent-scope acquire before the loads.
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int node = parent_leaf[j];
    while (node >= 0) {
        if (atomicAdd(&arrived[node], 1) == 0) return;               // the first child to arrive stops; the second has both boxes visible
        __threadfence();
        const int2 ch = children[node];
        const GPUBVHNode l = nodes[ch.x < 0 ? (leaves - 1) + ~ch.x : ch.x], r = nodes[ch.y < 0 ? (leaves - 1) + ~ch.y : ch.y];
        GPUBVHNode m;
        m.bbox_min = DsrtF3{fminf(l.bbox_min.x, r.bbox_min.x), fminf(l.bbox_min.y, r.bbox_min.y), fminf(l.bbox_min.z, r.bbox_min.z)};
        m.bbox_max = DsrtF3{fmaxf(l.bbox_max.x, r.bbox_max.x), fmaxf(l.bbox_max.y, r.bbox_max.y), fmaxf(l.bbox_max.z, r.bbox_max.z)};
        m.left = ch.x < 0 ? (leaves - 1) + ~ch.x : ch.x;
        m.right = ch.y < 0 ? (leaves - 1) + ~ch.y : ch.y;
        m.tri_offset = 0; m.tri_count = 0;
        nodes[node] = m;
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        node = node == 0 ? -1 : parent_internal[node];
    }
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
        const int N = (int)n, leaves = (N + 3) / 4, internal = leaves - 1, total = leaves + internal;

        Dev<GPUTriangle> d_tris; Dev<Bounds> d_tb; Dev<uint32_t> d_scene, d_code, d_code2, d_idx, d_idx2; Dev<unsigned long long> d_key;
        Dev<int2> d_children; Dev<int> d_pi, d_pl, d_arrived; Dev<GPUBVHNode> d_nodes; Dev<unsigned char> d_tmp;
        if (!d_tris.alloc(n) || !d_tb.alloc(n) || !d_scene.alloc(6) || !d_code.alloc(n) || !d_code2.alloc(n) || !d_idx.alloc(n) || !d_idx2.alloc(n) ||
            !d_key.alloc((size_t)leaves) || !d_children.alloc((size_t)internal) || !d_pi.alloc((size_t)internal) || !d_pl.alloc((size_t)leaves) ||
            !d_arrived.alloc((size_t)internal) || !d_nodes.alloc((size_t)total)) return DSRT_ERR_HIP;
        size_t tmp_bytes = 0;
        if (!ok(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_code.p, d_code2.p, d_idx.p, d_idx2.p, N, 0, 30), "hipcub size query")) return DSRT_ERR_HIP;
        if (!d_tmp.alloc(tmp_bytes)) return DSRT_ERR_HIP;
        if (!ok(hipMemcpy(d_tris.p, hs->tris.data(), n * sizeof(GPUTriangle), hipMemcpyHostToDevice), "hipMemcpy triangles")) return DSRT_ERR_HIP;

        hipEvent_t e0, e1;
        if (!ok(hipEventCreate(&e0), "hipEventCreate") || !ok(hipEventCreate(&e1), "hipEventCreate")) return DSRT_ERR_HIP;
        const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        bool good = ok(hipMemcpy(d_scene.p, init, sizeof init, hipMemcpyHostToDevice), "hipMemcpy");
        good = good && ok(hipEventRecord(e0, nullptr), "hipEventRecord");
        const unsigned bt = (unsigned)((n + 255) / 256), bl = (unsigned)((leaves + 255) / 256), bi = (unsigned)((internal + 255) / 256);
        if (good) {
            hipLaunchKernelGGL(tri_bounds_kernel, dim3(bt), dim3(256), 0, nullptr, d_tris.p, N, d_tb.p, d_scene.p);
            hipLaunchKernelGGL(morton_kernel, dim3(bt), dim3(256), 0, nullptr, d_tb.p, N, d_scene.p, d_code.p, d_idx.p);
            good = ok(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_code.p, d_code2.p, d_idx.p, d_idx2.p, N, 0, 30), "hipcub sort");
        }
        if (good) {
            hipLaunchKernelGGL(leaf_keys_kernel, dim3(bl), dim3(256), 0, nullptr, d_code2.p, N, leaves, d_key.p);
            good = ok(hipMemsetAsync(d_arrived.p, 0, (size_t)(internal ? internal : 1) * sizeof(int), nullptr), "hipMemsetAsync");
            if (good && internal > 0)
                hipLaunchKernelGGL(radix_tree_kernel, dim3(bi), dim3(256), 0, nullptr, d_key.p, leaves, d_children.p, d_pi.p, d_pl.p);
            if (good)
                hipLaunchKernelGGL(fit_kernel, dim3(bl), dim3(256), 0, nullptr, d_tb.p, d_idx2.p, N, leaves, d_children.p, d_pi.p, d_pl.p, d_arrived.p, d_nodes.p);
            good = good && ok(hipGetLastError(), "LBVH kernels") && ok(hipEventRecord(e1, nullptr), "hipEventRecord") && ok(hipEventSynchronize(e1), "hipEventSynchronize");
        }
        float ms = 0.0f;
        if (good) good = ok(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!good) return DSRT_ERR_HIP;

        hs->nodes.resize((size_t)total);
        std::vector<uint32_t> order(n);
        if (!ok(hipMemcpy(hs->nodes.data(), d_nodes.p, (size_t)total * sizeof(GPUBVHNode), hipMemcpyDeviceToHost), "hipMemcpy nodes") ||
            !ok(hipMemcpy(order.data(), d_idx2.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy order")) return DSRT_ERR_HIP;
        hs->tri_indices.assign(order.begin(), order.end());
        if (leaves == 1) {                                            // a single leaf is the whole tree: it must sit at index 0
            hs->nodes[0] = hs->nodes[(size_t)(leaves - 1)];
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
