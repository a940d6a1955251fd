// device_api.hip -- the device half of the C ABI (include/dsrt.h): context, scene upload + re-layout,
// render launch, tile de-interleave, and the reference's own entry points on top of them.
//
// Replaces: the cudaMalloc/cudaMemcpy half of build_gpu_scene (src/gpu_scene_builder.cpp:322-331, 475-546),
// free_gpu_scene (:603-626) and the host launcher gpu_render_scene (src/gpu_render.cu:1037-1108).
// Differences by design: the scene is converted once into the traversal layout of device_layout.h and stays
// resident across frames (the reference re-uploads everything per frame, src/main.cpp:405); errors are returned,
// not only printed; the launch is asynchronous on a caller-supplied stream; output goes to caller-owned buffers.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dsrt.h"
#include "../host/host_internal.hpp"
#include "device_layout.h"

namespace dsrt {
hipError_t launch_render(const RenderArgs& a, int lds_entries, int rng_mode, int blocks, bool count, bool checked, bool anyhit, bool lean, hipStream_t stream);
hipError_t launch_probe(const RenderArgs& a, int blocks, bool lean, hipStream_t stream);
hipError_t launch_render_batch(const RenderArgs& a, int rng_mode, int blocks, bool lean, hipStream_t stream);
hipError_t launch_batch_table(BatchFrame* table, const uint32_t* sched, uint32_t sched_stride, uint32_t frames, uint32_t tt, int rng_mode, int spp, int light_chunk_len,
                              uint32_t* total_items, hipStream_t stream);
hipError_t launch_resolve(const unsigned long long* sums, int spp, float inv_gamma, size_t n_pixels, uint8_t* out_rgb8, float* out_f32, hipStream_t stream);
hipError_t launch_philox(unsigned long long seed, unsigned long long sub, int n, uint32_t* ours, uint32_t* theirs, hipStream_t stream);
hipError_t launch_deinterleave(const uint8_t* gathered, uint8_t* image, int W, int H, int tile, int tiles_x, int shard_count,
                               size_t shard_stride_bytes, hipStream_t stream);
hipError_t launch_math(int fn, const float* x, float y, float* out, int n, hipStream_t stream);
hipError_t launch_devkat(int fn, const float* in, float* out, int n, hipStream_t stream);
int kernel_waves_per_block();
hipError_t launch_tile_order(const DeviceScene& S, const FrameParams& P, uint32_t* cost, uint32_t* order, uint32_t* sched, uint32_t items_per_pixel,
                             uint32_t resident_lanes, bool cull, hipStream_t stream, const BatchFrame* batch = nullptr, uint32_t frames = 1, uint32_t stride = 0);
hipError_t launch_tile_reorder(const uint32_t* work, uint32_t* order, uint32_t* tmp, const uint32_t* sched, hipStream_t stream);
hipError_t launch_content_hash(const uint32_t* words, size_t n_words, uint64_t salt, uint64_t* d_hash2, hipStream_t stream);
// DsrtRenderDesc.math_mode 1: the same kernels compiled against the device math library's sinf / cosf / powf (render_kernel.hip, second compilation)
namespace devlibm {
hipError_t launch_render(const RenderArgs& a, int lds_entries, int rng_mode, int blocks, bool count, bool checked, bool anyhit, bool lean, hipStream_t stream);
hipError_t launch_probe(const RenderArgs& a, int blocks, bool lean, hipStream_t stream);
hipError_t launch_render_batch(const RenderArgs& a, int rng_mode, int blocks, bool lean, hipStream_t stream);
hipError_t launch_resolve(const unsigned long long* sums, int spp, float inv_gamma, size_t n_pixels, uint8_t* out_rgb8, float* out_f32, hipStream_t stream);
}  // namespace devlibm
}  // namespace dsrt

using namespace dsrt;

namespace {

// Frames in flight on separate streams (dsrt_ctx_clone, dsrt_multi_render_sequence) only overlap on the device if each stream
// gets a hardware queue of its own; the HIP runtime maps streams onto 4 unless told otherwise BEFORE it initialises.  The
// library asks for 16 in its first dsrt_device_count / dsrt_ctx_create call -- documented there in include/dsrt.h; not at load time, so
// that loading the library changes nothing in the host process -- unless the host has set the variable itself.  (No effect if HIP is
// already up.)
void hw_queues_default() {
    static std::once_flag once;
    std::call_once(once, [] { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); });
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr) do { if (!hip_ok((expr), #expr)) return DSRT_ERR_HIP; } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { reset(); }
    void reset() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    int upload(const std::vector<T>& v) {
        reset();
        if (v.empty()) return DSRT_OK;
        HIP_TRY(hipMalloc((void**)&p, v.size() * sizeof(T)));
        n = v.size();
        HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return DSRT_OK;
    }
    int alloc(size_t count) {
        reset();
        if (!count) return DSRT_OK;
        HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return DSRT_OK;
    }
};

// The scene in traversal layout, owning its device memory.
struct PackedScene {
    DevBuf<float4> pairs, tri_pairs, tri_shade, tri_uv, tri_cert, materials;
    DevBuf<uint8_t> pair_depth;
    DevBuf<int2> big_leaves;
    DevBuf<GPUSphere> spheres;
    DevBuf<GPUTextureHeader> tex_headers;
    DevBuf<float> tex_pool;
    DeviceScene view{};
    GPUCamera camera{};
    DsrtF3 sun_dir{}, sun_radiance{};
    int sun_enabled = 0;
    bool has_second_tree = false;   // the certified second tree is resident (pack_scene)
    float scene_extent = 0.0f, scene_centre[3] = {0, 0, 0};
    bool lean = false;              // no spheres, no textures, Lambertian materials only: the production launches use the kernels' LEAN instantiation (path_machine.h)
    bool valid = false;
};

float4 as_f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
float bits(int i) { float f; std::memcpy(&f, &i, 4); return f; }

// Leaf size of the certified second tree (its structure is free: any tree over the reachable triangles will do).  Development override: DSRT_SECOND_TREE_LEAF=1..7.
int second_tree_leaf_max() {
    if (const char* e = std::getenv("DSRT_SECOND_TREE_LEAF")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1 && v <= 7) return (int)v; }
    return 4;
}

// What packing one tree into the shared arrays yields.
struct PackedTree { int root_ref = kRefNone; int stack_need = 0; float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; };

// The arrays both trees are appended to (device_layout.h).  References are absolute: an internal reference is its record's index in `pairs` + kRefBias, a leaf
// reference carries the index of its first record in `isect`; `shade` / `uv` / `cert` grow in step with `isect` (two slots per pair record).
struct PackArrays {
    std::vector<float4> pairs, isect, shade, uv, cert;
    std::vector<int> depth_of_all;                                       // depth of every record of `pairs` (the counting build's histogram reads it)
    std::vector<int2> big;
};

// Appends the tree (nodes, tri_indices) over the scene's triangles.  Validates every index the kernel will follow, so the fast (unchecked) kernel build never sees
// an out-of-range reference.  `cert_boxes` (null, or 6 floats per TRIANGLE): appended per slot to arr.cert -- the certified second tree's records.
int pack_tree(const GPUScene& h, const GPUBVHNode* nodes, int M, const int* tri_indices, int n_indexed, bool textured, const float* cert_boxes, PackArrays& arr, PackedTree& out) {
    const int N = h.num_triangles;
    for (int i = 0; i < n_indexed; ++i) if (tri_indices[i] < 0 || tri_indices[i] >= N) { set_error("tri_indices entry out of range"); return DSRT_ERR_INVALID; }
    // internal node -> slot in `pairs`, assigned in a depth-first walk from the root (also detects cycles / sharing)
    const int base = (int)(arr.pairs.size() / 4);
    std::vector<int> slot_of(M, -1);
    std::vector<char> seen(M, 0);
    struct Item { int node; int internal_above; };
    std::vector<Item> todo{{0, 0}};
    int stack_need = 0;
    std::vector<int> order;                                          // internal nodes in visiting order
    std::vector<int> depth_of;                                       // ... and their depth (root = 0)
    while (!todo.empty()) {
        Item it = todo.back(); todo.pop_back();
        if (it.node < 0 || it.node >= M) { set_error("BVH child index out of range"); return DSRT_ERR_INVALID; }
        if (seen[it.node]) { set_error("BVH is not a tree (node reached twice)"); return DSRT_ERR_INVALID; }
        seen[it.node] = 1;
        const GPUBVHNode& n = nodes[it.node];
        if (n.tri_count > 0) {
            if (n.tri_offset < 0 || (long long)n.tri_offset + n.tri_count > n_indexed) { set_error("BVH leaf range out of bounds"); return DSRT_ERR_INVALID; }
            if (it.internal_above > stack_need) stack_need = it.internal_above;
        } else {
            slot_of[it.node] = (int)order.size();
            order.push_back(it.node);
            depth_of.push_back(it.internal_above);
            todo.push_back({n.right, it.internal_above + 1});
            todo.push_back({n.left, it.internal_above + 1});
        }
    }
    if (stack_need > 64) { set_error("BVH needs a traversal stack deeper than the reference's 64 entries"); return DSRT_ERR_BVH_DEPTH; }
    // Triangle storage: every leaf gets ceil(count / 2) pair records of its own, filled in tri_indices order.
    auto emit_leaf = [&](const GPUBVHNode& n) -> int {
        const int first_pair = (int)(arr.isect.size() / 5);
        for (int i = 0; i < n.tri_count; i += 2) {
            float q[2][9] = {{0}};
            for (int w = 0; w < 2; ++w) {
                const bool real = i + w < n.tri_count;
                const int src = real ? tri_indices[n.tri_offset + i + w] : -1;
                float4 s0 = as_f4(0, 0, 0, 0), s1 = s0, s2 = as_f4(0, bits(0), bits(-1), bits(-1)), u0 = s0, u1 = s0, c0 = s0, c1 = s0;
                if (real) {
                    const GPUTriangle& t = h.triangles[src];
                    const float v[9] = {t.v0.x, t.v0.y, t.v0.z, t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z, t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
                    std::memcpy(q[w], v, sizeof v);
                    s0 = as_f4(t.n0.x, t.n0.y, t.n0.z, t.n1.x);
                    s1 = as_f4(t.n1.y, t.n1.z, t.n2.x, t.n2.y);
                    s2 = as_f4(t.n2.z, bits(t.material_id), bits(t.albedo_tex), bits(src));
                    u0 = as_f4(t.uv0.x, t.uv0.y, t.uv1.x, t.uv1.y);
                    u1 = as_f4(t.uv2.x, t.uv2.y, 0.0f, 0.0f);
                    if (cert_boxes) { const float* cb = cert_boxes + 6 * (size_t)src; c0 = as_f4(cb[0], cb[1], cb[2], cb[3]); c1 = as_f4(cb[4], cb[5], 0.0f, 0.0f); }
                }
                arr.shade.push_back(s0); arr.shade.push_back(s1); arr.shade.push_back(s2);
                if (textured) { arr.uv.push_back(u0); arr.uv.push_back(u1); }
                if (cert_boxes) { arr.cert.push_back(c0); arr.cert.push_back(c1); }
            }
            arr.isect.push_back(as_f4(q[0][0], q[1][0], q[0][1], q[1][1]));
            arr.isect.push_back(as_f4(q[0][2], q[1][2], q[0][3], q[1][3]));
            arr.isect.push_back(as_f4(q[0][4], q[1][4], q[0][5], q[1][5]));
            arr.isect.push_back(as_f4(q[0][6], q[1][6], q[0][7], q[1][7]));
            arr.isect.push_back(as_f4(q[0][8], q[1][8], 0.0f, 0.0f));
        }
        return first_pair;
    };
    auto ref_of = [&](int node) -> int {
        const GPUBVHNode& n = nodes[node];
        if (n.tri_count <= 0) return base + slot_of[node] + kRefBias;
        const int first_pair = emit_leaf(n);
        if (n.tri_count <= 7) return make_leaf_ref(n.tri_count - 1, first_pair);
        arr.big.push_back(make_int2(first_pair, n.tri_count));
        return make_leaf_ref(7, (int)arr.big.size() - 1);
    };
    arr.pairs.resize(((size_t)base + order.size()) * 4);
    for (size_t s = 0; s < order.size(); ++s) {
        const GPUBVHNode& n = nodes[order[s]];
        const GPUBVHNode& l = nodes[n.left];
        const GPUBVHNode& r = nodes[n.right];
        float4* rec = arr.pairs.data() + 4 * ((size_t)base + s);
        rec[0] = as_f4(l.bbox_min.x, r.bbox_min.x, l.bbox_max.x, r.bbox_max.x);
        rec[1] = as_f4(l.bbox_min.y, r.bbox_min.y, l.bbox_max.y, r.bbox_max.y);
        rec[2] = as_f4(l.bbox_min.z, r.bbox_min.z, l.bbox_max.z, r.bbox_max.z);
        const int ref_left = ref_of(n.left), ref_right = ref_of(n.right);          // in this order: leaf records follow the walk
        rec[3] = as_f4(bits(ref_left), bits(ref_right), l.bbox_min.x + l.bbox_max.x, r.bbox_min.x + r.bbox_max.x);   // the x sums of the ordering test, in float as the kernel would form them
    }
    arr.depth_of_all.insert(arr.depth_of_all.end(), depth_of.begin(), depth_of.end());
    const GPUBVHNode& root = nodes[0];
    out.lo[0] = root.bbox_min.x; out.lo[1] = root.bbox_min.y; out.lo[2] = root.bbox_min.z;
    out.hi[0] = root.bbox_max.x; out.hi[1] = root.bbox_max.y; out.hi[2] = root.bbox_max.z;
    out.root_ref = ref_of(0);
    out.stack_need = stack_need;
    return DSRT_OK;
}

// Host-side conversion of reference-layout arrays into the traversal layout.  `second_tree`: also build and pack the certified second tree (below).
int pack_scene(const GPUScene& h, PackedScene& out, bool second_tree) {
    const int N = h.num_triangles, M = h.num_bvh_nodes;
    if (N < 0 || M < 0 || h.num_spheres < 0 || h.num_materials < 0 || h.num_textures < 0 || h.texture_pool_floats < 0) {
        set_error("scene has a negative count"); return DSRT_ERR_INVALID;
    }
    if ((N && !h.triangles) || (h.num_spheres && !h.spheres) || (h.num_materials && !h.materials)) { set_error("scene array pointer is null"); return DSRT_ERR_INVALID; }
    if (N > (1 << 28)) { set_error("more than 2^28 triangles"); return DSRT_ERR_INVALID; }
    const bool has_bvh = h.bvh_nodes && M > 0 && h.tri_indices;        // bvh_hit_closest's own guard, src/gpu_render.cu:394-397
    for (int i = 0; i < N; ++i) if (h.triangles[i].material_id < 0 || h.triangles[i].material_id >= h.num_materials) { set_error("triangle material id out of range"); return DSRT_ERR_INVALID; }
    for (int i = 0; i < h.num_spheres; ++i) if (h.spheres[i].material_id < 0 || h.spheres[i].material_id >= h.num_materials) { set_error("sphere material id out of range"); return DSRT_ERR_INVALID; }

    PackArrays arr;
    std::vector<float4> mats;
    DeviceScene& v = out.view;
    std::memset(&v, 0, sizeof v);
    v.root_ref = kRefNone;
    v.accel_root_ref = kRefNone;
    out.has_second_tree = false;

    if (has_bvh) {
        const bool textured = h.num_textures > 0 && h.textures && h.texture_pool;
        arr.isect.reserve(((size_t)N / 2 + (size_t)M / 2 + 1) * 5 * (second_tree ? 2 : 1));
        arr.shade.reserve(((size_t)N + (size_t)M / 2 + 2) * 3 * (second_tree ? 2 : 1));
        PackedTree ref_tree, acc_tree;
        if (second_tree && N > 0) {
            // THE CERTIFIED SECOND TREE.  A ray's answer on the reference's tree is the accepted triangle of smallest t, except where it depends on the reference's own
            // boxes or order; path_machine.h checks, per ray, the three conditions under which it provably is (the certificate) and re-walks the reference tree
            // otherwise.  What the check needs from here: (a) the triangles the reference walk can never reach -- a zero-thickness box on their root-to-leaf path:
            // bbox_hit's `t_max <= t_min` holds with equality, src/gpu_render.cu:312 -- are left OUT of the second tree; (b) every triangle's leaf box ON THE
            // REFERENCE TREE rides with its slot (tri_cert); (c) the second tree's boxes are widened by 2^-16 of the scene's extent, so that no rounding of the slab
            // arithmetic makes a ray miss the box of a triangle it hits (ray origins up to 30 extents away: render_impl checks the camera).
            SecondTree st2;
            int rc = prepare_second_tree(h, second_tree_leaf_max(), st2);        // host/bvh_sah.cpp: unreachable triangles, reference-leaf boxes, the widened SAH tree, the checks
            if (rc != DSRT_OK) return rc;
            std::vector<GPUBVHNode>& nodes2 = st2.nodes;
            std::vector<int>& order2 = st2.order;
            std::vector<float>& leaf_box = st2.leaf_box;
            const GPUBVHNode& root = h.bvh_nodes[0];
            const float extent = st2.extent;
            const bool origins_near = st2.origins_near;
            // (both trees share one array of node records addressed by a 32-bit byte offset: a scene too big for two trees in it keeps the reference tree only)
            const bool fits = nodes2.size() / 2 + (size_t)M / 2 + 2 * (size_t)kRefBias < ((size_t)1 << 26) && (order2.size() + (size_t)N) / 2 + nodes2.size() / 2 + (size_t)M / 2 < ((size_t)1 << 28);
            if (!nodes2.empty() && origins_near && fits) {
                if ((rc = pack_tree(h, nodes2.data(), (int)nodes2.size(), order2.data(), (int)order2.size(), textured, leaf_box.data(), arr, acc_tree))) return rc;
                out.has_second_tree = true;
                out.scene_extent = extent;
                for (int a = 0; a < 3; ++a) out.scene_centre[a] = 0.5f * ((&root.bbox_min.x)[a] + (&root.bbox_max.x)[a]);
            }
        }
        int rc = pack_tree(h, h.bvh_nodes, M, h.tri_indices, N, textured, nullptr, arr, ref_tree);
        if (rc) return rc;
        for (int a = 0; a < 3; ++a) { v.root_lo[a] = ref_tree.lo[a]; v.root_hi[a] = ref_tree.hi[a]; v.accel_root_lo[a] = acc_tree.lo[a]; v.accel_root_hi[a] = acc_tree.hi[a]; }
        v.root_ref = ref_tree.root_ref;
        v.accel_root_ref = out.has_second_tree ? acc_tree.root_ref : kRefNone;
        v.stack_need = std::max(ref_tree.stack_need, acc_tree.stack_need);

        if (arr.isect.size() / 5 > (size_t)(1 << 28)) { set_error("too many triangle pair records"); return DSRT_ERR_INVALID; }
        if (arr.pairs.size() / 4 + kRefBias >= ((size_t)1 << 26)) { set_error("more than 2^26 - 64 internal BVH nodes (the kernel addresses node records by a 32-bit byte offset)"); return DSRT_ERR_INVALID; }
    }
    std::vector<float4>& pairs = arr.pairs; std::vector<float4>& isect = arr.isect; std::vector<float4>& shade = arr.shade; std::vector<float4>& uv = arr.uv;
    std::vector<int2>& big = arr.big; std::vector<int>& depth_of_all = arr.depth_of_all;
    mats.resize((size_t)h.num_materials * 3);
    for (int i = 0; i < h.num_materials; ++i) {
        const GPUMaterial& m = h.materials[i];
        mats[3 * (size_t)i + 0] = as_f4(bits(m.type), bits(m.albedo_tex), 0.0f, 0.0f);
        mats[3 * (size_t)i + 1] = as_f4(m.albedo.x, m.albedo.y, m.albedo.z, m.emissive.x);
        mats[3 * (size_t)i + 2] = as_f4(m.emissive.y, m.emissive.z, m.fuzz, m.ref_idx);
    }
    int num_lights = 0;                                                   // src/gpu_render.cu:841-847, a scene constant
    for (int i = 0; i < h.num_spheres; ++i) {
        const GPUMaterial& lm = h.materials[h.spheres[i].material_id];
        if (lm.type == MAT_DIFFUSE_LIGHT && (lm.emissive.x > 0 || lm.emissive.y > 0 || lm.emissive.z > 0)) num_lights++;
    }
    if (h.num_textures > 0 && h.textures && h.texture_pool) {
        for (int i = 0; i < h.num_textures; ++i) {
            const GPUTextureHeader& th = h.textures[i];
            if (th.width < 1 || th.height < 1 || th.offset < 0) { set_error("texture header out of range"); return DSRT_ERR_INVALID; }
        }
    }

    int rc;
    {
        std::vector<uint8_t> depth_bytes(pairs.size() / 4);
        for (size_t i = 0; i < depth_bytes.size(); ++i) depth_bytes[i] = (uint8_t)depth_of_all[i];
        if ((rc = out.pair_depth.upload(depth_bytes))) return rc;
    }
    if ((rc = out.pairs.upload(pairs)) || (rc = out.tri_pairs.upload(isect)) || (rc = out.tri_shade.upload(shade)) ||
        (rc = out.tri_uv.upload(uv)) || (rc = out.tri_cert.upload(arr.cert)) || (rc = out.big_leaves.upload(big)) || (rc = out.materials.upload(mats))) return rc;
    std::vector<GPUSphere> sph(h.spheres, h.spheres + h.num_spheres);
    if ((rc = out.spheres.upload(sph))) return rc;
    if (h.num_textures > 0 && h.textures && h.texture_pool) {
        std::vector<GPUTextureHeader> th(h.textures, h.textures + h.num_textures);
        std::vector<float> pool(h.texture_pool, h.texture_pool + h.texture_pool_floats);
        if ((rc = out.tex_headers.upload(th)) || (rc = out.tex_pool.upload(pool))) return rc;
    } else { out.tex_headers.reset(); out.tex_pool.reset(); }

    v.pair_depth = out.pair_depth.p; v.pairs = out.pairs.p; v.pairs_biased = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(out.pairs.p) - (uintptr_t)kRefBias * 64u); v.tri_pairs = out.tri_pairs.p; v.tri_shade = out.tri_shade.p; v.tri_uv = out.tri_uv.p; v.tri_cert = out.tri_cert.p;
    v.big_leaves = out.big_leaves.p; v.materials = out.materials.p; v.spheres = out.spheres.p;
    v.tex_headers = out.tex_headers.p; v.tex_pool = out.tex_pool.p;
    v.num_pairs = (int)(pairs.size() / 4); v.num_tri_pairs = (int)(isect.size() / 5); v.num_big_leaves = (int)big.size();
    v.num_materials = h.num_materials; v.num_spheres = h.num_spheres; v.num_lights = num_lights;
    v.num_textures = (int)out.tex_headers.n; v.tex_pool_floats = (int)out.tex_pool.n;
    out.camera = h.camera;
    out.sun_dir = h.sun_dir; out.sun_radiance = h.sun_radiance; out.sun_enabled = h.sun_enabled ? 1 : 0;
    out.lean = h.num_spheres == 0 && out.tex_headers.n == 0;
    for (int i = 0; i < h.num_materials && out.lean; ++i) out.lean = h.materials[i].type == MAT_LAMBERTIAN;
    out.valid = true;
    return DSRT_OK;
}

}  // namespace

struct DsrtContext {
    int device = 0;
    int num_cus = 0;
    // The resident scene is shared between a context and its clones (dsrt_ctx_clone): one copy in HBM however many frames are in
    // flight.  Camera and sun are per context (they are what changes per frame); so are the working buffers below.
    std::shared_ptr<PackedScene> scene;
    GPUCamera camera{};
    DsrtF3 sun_dir{}, sun_radiance{};
    int sun_enabled = 0;
    bool want_second_tree = false;  // dsrt_ctx_set_certified_tree / DSRT_CERTIFIED_TREE: the next upload also builds and packs the certified second tree
    DevBuf<uint32_t> ctrl;          // [0] queue, [1] flags, then counters (uint64 x kNumCounters) at byte 16
    DevBuf<uint2> spill;
    DevBuf<uint32_t> tile_cost, tile_order, tile_work, tile_tmp, probe_queue;
    DevBuf<BatchFrame> batch_table;  // dsrt_render_batch: one entry per frame
    std::vector<BatchFrame> batch_host;
    DevBuf<unsigned long long> accum_fixed;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t done = nullptr;      // recorded behind every render: the next render on ANY stream waits for it (queue words, spill strip,
    bool done_valid = false;        // pre-pass arrays and partial sums are per context, so a context has one render in flight)
    ~DsrtContext() { if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); if (done) (void)hipEventDestroy(done); }
};

namespace {

// A fresh PackedScene for this context (clones made earlier keep the one they share until they are re-cloned or destroyed).
int install_scene(DsrtContext* ctx, const GPUScene& host_layout) {
    auto fresh = std::make_shared<PackedScene>();
    ctx->scene.reset();
    const int rc = pack_scene(host_layout, *fresh, ctx->want_second_tree);
    if (rc) return rc;
    ctx->scene = std::move(fresh);
    ctx->camera = ctx->scene->camera;
    ctx->sun_dir = ctx->scene->sun_dir; ctx->sun_radiance = ctx->scene->sun_radiance; ctx->sun_enabled = ctx->scene->sun_enabled;
    return DSRT_OK;
}

constexpr size_t kQueueLightWord = 96;                       // the light queue's counter: its own cache line, past the counters
constexpr size_t kCtrlWords = kQueueLightWord + 16;
static_assert(4 + 2 * (size_t)kNumCounters <= kQueueLightWord, "counters overlap the second queue word");

struct Tiling { int tile, tiles_x, tiles_y, total, mine, padded; };
bool make_tiling(const DsrtRenderDesc& d, Tiling& t) {
    t.tile = d.tile_size > 0 ? d.tile_size : 8;
    if (t.tile % 8 != 0 || t.tile > 1024 || d.width < 2 || d.height < 2) return false;
    const int count = d.shard_count > 1 ? d.shard_count : 1;
    if (d.shard_rank < 0 || d.shard_rank >= count) return false;
    t.tiles_x = (d.width + t.tile - 1) / t.tile;
    t.tiles_y = (d.height + t.tile - 1) / t.tile;
    t.total = t.tiles_x * t.tiles_y;
    t.mine = (t.total - d.shard_rank + count - 1) / count;
    t.padded = (t.total + count - 1) / count;
    return true;
}

// Development switches.  They are NOT part of a render's description (include/dsrt.h refuses unknown bits of DsrtRenderDesc.tune[3]); the A/B tools under
// tools/ set them through dsrt_dev_set_experiment(), and a process started with the environment variable DSRT_EXPERIMENT (an integer in C syntax) takes that as
// the initial word -- read ONCE, in the first dsrt_device_count / dsrt_ctx_create call, never per render.  Undefined bits are refused, and a non-zero word is
// announced on stderr, so that a stray variable cannot silently change how a production process schedules its work.
//   64           8 probe samples per pixel instead of 4
//   128          leaf records are not dealt to idle lanes (render_kernel.hip, phase L)
// (the low six bits are never used here: tools/ab_tune.py splits one number into the DSRT_TUNE_* flags and this word)
//   bits 8-19    rng_mode 1: slices per heavy pixel (0 = chosen by the pre-pass)
//   bits 20-22   grid = resident set >> n (frames that overlap on separate streams)
//   bit 24       the general kernels even for a scene that qualifies for the LEAN instantiation (A/B of the two)
//   bit 27       COUNTING BUILD of rng_mode 0 only: the float image receives per pixel (fetch time, end time, wave) as bit patterns,
//                100 MHz ticks, instead of the colour (tools/chain_timeline.py) -- the one switch that changes output (the float image; never the bytes)
//   bits 28-29   rng_mode 1: least samples per work item of a background pixel, 0 = 128, 1 = 64, 2 = 256, 3 = 512
//   bit 31       rng_mode 1: background pixels one item each
// Apart from bit 27 none of them changes a pixel (tests/test_gpu_parity.py runs the render under several of them against the oracle).
constexpr uint32_t kExperimentDefined = 64u | 128u | (0xFFFu << 8) | (7u << 20) | (1u << 24) | (1u << 27) | (3u << 28) | (1u << 31);
std::atomic<uint32_t> g_experiment{0u};

int set_experiment(uint32_t word, const char* from) {
    if (word & ~kExperimentDefined) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s: development switch word 0x%x has undefined bits (defined: 0x%x)", from, word, kExperimentDefined);
        set_error(buf);
        return DSRT_ERR_INVALID;
    }
    if (word != g_experiment.exchange(word) && word) std::fprintf(stderr, "libdsrt_hip: development switches 0x%x in effect (%s)\n", word, from);
    return DSRT_OK;
}

void experiment_from_environment() {            // once per process
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = std::getenv("DSRT_EXPERIMENT");
        if (e && *e && set_experiment((uint32_t)std::strtoul(e, nullptr, 0), "environment variable DSRT_EXPERIMENT") != DSRT_OK)
            std::fprintf(stderr, "libdsrt_hip: DSRT_EXPERIMENT ignored: %s\n", dsrt_last_error());
    });
}

uint32_t experiment_word() { return g_experiment.load(); }

}  // namespace

extern "C" {

int dsrt_dev_set_experiment(uint32_t word) {
    experiment_from_environment();              // (so that a later first context does not overwrite this call's word with the variable's)
    return set_experiment(word, "dsrt_dev_set_experiment");
}

int dsrt_device_count(void) {
    hw_queues_default();
    experiment_from_environment();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dsrt_ctx_create(int device, DsrtContext** out) {
    if (!out) { set_error("dsrt_ctx_create: null out"); return DSRT_ERR_INVALID; }
    *out = nullptr;
    hw_queues_default();
    experiment_from_environment();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return DSRT_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { set_error("device index out of range"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
        return DSRT_ERR_NO_DEVICE;
    }
    auto* ctx = new DsrtContext();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (const char* e = std::getenv("DSRT_CERTIFIED_TREE")) ctx->want_second_tree = e[0] == '1';
    int rc = ctx->ctrl.alloc(kCtrlWords);
    if (rc) { delete ctx; return rc; }
    if (!hip_ok(hipEventCreate(&ctx->ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&ctx->ev1), "hipEventCreate") ||
        !hip_ok(hipEventCreateWithFlags(&ctx->done, hipEventDisableTiming), "hipEventCreateWithFlags")) { delete ctx; return DSRT_ERR_HIP; }
    *out = ctx;
    return DSRT_OK;
}

int dsrt_ctx_clone(const DsrtContext* src, DsrtContext** out) {
    if (!src || !out) { set_error("dsrt_ctx_clone: null argument"); return DSRT_ERR_INVALID; }
    const int rc = dsrt_ctx_create(src->device, out);
    if (rc) return rc;
    (*out)->scene = src->scene;                    // shared, read-only on the device
    (*out)->want_second_tree = src->want_second_tree;
    (*out)->camera = src->camera;
    (*out)->sun_dir = src->sun_dir; (*out)->sun_radiance = src->sun_radiance; (*out)->sun_enabled = src->sun_enabled;
    return DSRT_OK;
}

int dsrt_ctx_device(const DsrtContext* ctx) { return ctx ? ctx->device : -1; }

int dsrt_ctx_set_certified_tree(DsrtContext* ctx, int on) {
    if (!ctx) { set_error("dsrt_ctx_set_certified_tree: null context"); return DSRT_ERR_INVALID; }
    ctx->want_second_tree = on != 0;
    return DSRT_OK;
}

int dsrt_ctx_has_certified_tree(const DsrtContext* ctx) { return ctx && ctx->scene && ctx->scene->valid && ctx->scene->has_second_tree ? 1 : 0; }

void dsrt_ctx_destroy(DsrtContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    delete ctx;
}

int dsrt_scene_upload(DsrtContext* ctx, const GPUScene* scene) {
    return dsrt::guarded("dsrt_scene_upload", [&]() -> int {
    if (!ctx || !scene) { set_error("dsrt_scene_upload: null argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    return install_scene(ctx, *scene);
    });
}

int dsrt_scene_upload_device(DsrtContext* ctx, const GPUScene* d) {
    return dsrt::guarded("dsrt_scene_upload_device", [&]() -> int {
    if (!ctx || !d) { set_error("dsrt_scene_upload_device: null argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (d->num_triangles < 0 || d->num_spheres < 0 || d->num_materials < 0 || d->num_bvh_nodes < 0 || d->num_textures < 0 || d->texture_pool_floats < 0) {
        set_error("scene has a negative count"); return DSRT_ERR_INVALID;
    }
    // Bring the reference-layout arrays back to the host, then convert as usual.
    std::vector<GPUTriangle> tris((size_t)(d->triangles ? d->num_triangles : 0));
    std::vector<GPUSphere> sph((size_t)(d->spheres ? d->num_spheres : 0));
    std::vector<GPUMaterial> mats((size_t)(d->materials ? d->num_materials : 0));
    std::vector<int> idx((size_t)(d->tri_indices ? d->num_triangles : 0));
    std::vector<GPUBVHNode> nodes((size_t)(d->bvh_nodes ? d->num_bvh_nodes : 0));
    std::vector<GPUTextureHeader> th((size_t)(d->textures ? d->num_textures : 0));
    std::vector<float> pool((size_t)(d->texture_pool ? d->texture_pool_floats : 0));
    auto pull = [](void* dst, const void* src, size_t bytes) { return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess; };
    HIP_TRY(pull(tris.data(), d->triangles, tris.size() * sizeof(GPUTriangle)));
    HIP_TRY(pull(sph.data(), d->spheres, sph.size() * sizeof(GPUSphere)));
    HIP_TRY(pull(mats.data(), d->materials, mats.size() * sizeof(GPUMaterial)));
    HIP_TRY(pull(idx.data(), d->tri_indices, idx.size() * sizeof(int)));
    HIP_TRY(pull(nodes.data(), d->bvh_nodes, nodes.size() * sizeof(GPUBVHNode)));
    HIP_TRY(pull(th.data(), d->textures, th.size() * sizeof(GPUTextureHeader)));
    HIP_TRY(pull(pool.data(), d->texture_pool, pool.size() * sizeof(float)));
    GPUScene h = *d;
    h.triangles = tris.empty() ? nullptr : tris.data();
    h.spheres = sph.empty() ? nullptr : sph.data();
    h.materials = mats.empty() ? nullptr : mats.data();
    h.tri_indices = idx.empty() ? nullptr : idx.data();
    h.bvh_tri_indices = const_cast<int*>(h.tri_indices);
    h.bvh_nodes = nodes.empty() ? nullptr : nodes.data();
    h.textures = th.empty() ? nullptr : th.data();
    h.texture_pool = pool.empty() ? nullptr : pool.data();
    return install_scene(ctx, h);
    });
}

int dsrt_scene_set_camera_sun(DsrtContext* ctx, const GPUCamera* cam, const float sun_dir_model[3]) {
    if (!ctx || !cam) { set_error("dsrt_scene_set_camera_sun: null argument"); return DSRT_ERR_INVALID; }
    if (!ctx->scene || !ctx->scene->valid) { set_error("no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    ctx->camera = *cam;
    if (sun_dir_model) ctx->sun_dir = DsrtF3{sun_dir_model[0], sun_dir_model[1], sun_dir_model[2]};
    return DSRT_OK;
}

int dsrt_shard_layout(const DsrtRenderDesc* desc, int* tiles_total, int* tiles_this_shard, int* tiles_per_shard_padded, size_t* rgb8_bytes_padded) {
    Tiling t;
    if (!desc || !make_tiling(*desc, t)) { set_error("dsrt_shard_layout: bad descriptor"); return DSRT_ERR_INVALID; }
    if (tiles_total) *tiles_total = t.total;
    if (tiles_this_shard) *tiles_this_shard = t.mine;
    if (tiles_per_shard_padded) *tiles_per_shard_padded = t.padded;
    if (rgb8_bytes_padded) *rgb8_bytes_padded = (size_t)t.padded * t.tile * t.tile * 3;
    return DSRT_OK;
}

// What a batch launch adds to a render: the frames' cameras and sun directions (the context's own camera is not used).
struct BatchInput { int frames; const GPUCamera* cameras; const DsrtF3* sun_dirs; };

static int render_impl(DsrtContext* ctx, const DsrtRenderDesc* desc, uint8_t* d_rgb8, float* d_f32, void* stream_v, DsrtStats* stats, const BatchInput* batch) {
    if (!ctx || !desc || !d_rgb8) { set_error("dsrt_render: null argument"); return DSRT_ERR_INVALID; }
    if (!ctx->scene || !ctx->scene->valid) { set_error("dsrt_render: no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    if (desc->rng_mode != 0 && desc->rng_mode != 1) { set_error("dsrt_render: rng_mode must be 0 (reference LCG stream per pixel) or 1 (Philox4x32-10 stream per sample)"); return DSRT_ERR_INVALID; }
    if (desc->math_mode != 0 && desc->math_mode != 1) { set_error("dsrt_render: math_mode must be 0 (deterministic sin / cos / pow shared with the CPU oracle) or 1 (the device math library's)"); return DSRT_ERR_INVALID; }
    const bool libm = desc->math_mode == 1;
    const bool lean = ctx->scene->lean && !(experiment_word() & (1u << 24));
    if (desc->tune[3] & ~DSRT_TUNE_FLAG_MASK) { set_error("dsrt_render: tune[3] has bits set that this ABI version does not define (DSRT_TUNE_* in include/dsrt.h)"); return DSRT_ERR_INVALID; }
    const uint32_t flags = (uint32_t)desc->tune[3];
    const uint32_t xp = experiment_word();
    Tiling t;
    if (!make_tiling(*desc, t)) { set_error("dsrt_render: bad size, tile or shard"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;
    const PackedScene& sc = *ctx->scene;
    if (ctx->done_valid) HIP_TRY(hipStreamWaitEvent(stream, ctx->done, 0));     // a context's working buffers serve one render at a time

    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    a.scene = sc.view;
    FrameParams& f = a.frame;
    const GPUCamera& c = ctx->camera;
    {
        const float cam12[12] = {c.origin.x, c.origin.y, c.origin.z, c.lower_left_corner.x, c.lower_left_corner.y, c.lower_left_corner.z,
                                 c.horizontal.x, c.horizontal.y, c.horizontal.z, c.vertical.x, c.vertical.y, c.vertical.z};
        static_assert(sizeof cam12 == sizeof f.cam, "camera block");
        std::memcpy(f.cam, cam12, sizeof cam12);
    }
    f.sun_dir[0] = ctx->sun_dir.x; f.sun_dir[1] = ctx->sun_dir.y; f.sun_dir[2] = ctx->sun_dir.z;
    f.sun_radiance[0] = ctx->sun_radiance.x; f.sun_radiance[1] = ctx->sun_radiance.y; f.sun_radiance[2] = ctx->sun_radiance.z;
    f.sun_enabled = ctx->sun_enabled;
    f.width = desc->width; f.height = desc->height;
    f.spp = desc->spp < 1 ? 1 : desc->spp;                              // src/gpu_render.cu:987-988
    f.max_depth = desc->max_depth > 0 ? desc->max_depth : 12;           // :723-725
    const float gamma = desc->gamma > 0.0f ? desc->gamma : 1.0f;        // :1043
    f.inv_gamma = 1.0f / gamma;
    f.seed32 = (uint32_t)(desc->seed & 0xFFFFFFFFu);
    f.seed_hi = (uint32_t)(desc->seed >> 32);
    f.tile = t.tile; f.tiles_x = t.tiles_x; f.tiles_y = t.tiles_y;
    f.shard_rank = desc->shard_rank; f.shard_count = desc->shard_count > 1 ? desc->shard_count : 1;
    f.local_tiles = t.mine;
    if ((unsigned long long)t.mine * (unsigned long long)(t.tile * t.tile) >= (1ull << 31)) { set_error("dsrt_render: image too large (2^31 pixels per shard)"); return DSRT_ERR_INVALID; }
    f.total_items = (uint32_t)t.mine * (uint32_t)(t.tile * t.tile);
    f.compact_output = desc->shard_count > 1 ? 1 : 0;
    f.chunks = 1; f.chunk_len = f.spp; f.light_chunk_len = f.spp;
    const int frames = batch ? batch->frames : 1;
    if (batch) {
        if (frames < 1 || !batch->cameras || !batch->sun_dirs) { set_error("dsrt_render_batch: no frames"); return DSRT_ERR_INVALID; }
        if (desc->collect_counters || desc->checked) { set_error("dsrt_render_batch uses the production kernel only (no counters, not checked)"); return DSRT_ERR_INVALID; }
        // 32-bit output indices and work-item numbers: frames x pixels (x 8 sample slices in rng_mode 1) stay below 2^32
        const unsigned long long frame_px = desc->shard_count > 1 ? (unsigned long long)t.padded * t.tile * t.tile : (unsigned long long)desc->width * desc->height;
        if ((unsigned long long)frames * frame_px * (desc->rng_mode == 1 ? 16ull : 1ull) >= (1ull << 32)) {
            set_error("dsrt_render_batch: too many pixels in one batch (split the sequence)"); return DSRT_ERR_INVALID;
        }
    }
    const size_t out_pixels = (desc->shard_count > 1 ? (size_t)t.padded * t.tile * t.tile : (size_t)desc->width * desc->height) * (size_t)frames;
    if (desc->rng_mode == 1) {
        // a pixel's samples are independent streams: pixels of tiles that see geometry are split into 8 work items.  More slices cost
        // more in half-empty advance passes than their shorter tail saves; fewer leave whole waves without work at the end of a small
        // job, which sample stealing (inside a wave, path_machine.h) cannot repair.  One rank's share of an 8-GPU 1080p x 1000 frame,
        // stealing on: 1 / 2 / 4 / 8 / 16 / 32 / 64 slices = 307 / 289 / 188 / 171 / 181 / 191 / 203 ms; whole frame on one GPU
        // 1091 / 1076 / 1063 / 1050 / 1047 (profiles/r02/README.md).  An item's integer sums are 32-bit in units of 2^-20: at most 4095
        // samples per item, so with more samples than that every pixel is sliced, whatever its tile sees.
        f.chunk_len = (f.spp + 7) / 8;
        if ((xp >> 8) & 0xFFFu) f.chunk_len = (f.spp + (int)((xp >> 8) & 0xFFFu) - 1) / (int)((xp >> 8) & 0xFFFu);   // experiment: slices per pixel
        if (f.chunk_len < 1) f.chunk_len = 1;
        if (f.chunk_len > 4095) f.chunk_len = 4095;
        f.chunks = (f.spp + f.chunk_len - 1) / f.chunk_len;
        // Background pixels (tiles that see no geometry) are cut too, but not below kLightLen samples per item: uncut, their 1000-sample
        // items are the longest jobs of a frame and the light queue is served last (one of 8 shares of the near frame 166 -> 149 ms, frame
        // 70 60 -> 43 ms); cut as finely as the heavy pixels, their three 64-bit atomics per item cost the 250-spp sequence, whose frames
        // overlap and have no tail to lose, a fifth of its frame rate (47 -> 38 frames/s).
        {
            const int code = (int)((xp >> 28) & 3u);                                          // experiment
            const int kLightLen = code == 0 ? 128 : (code == 1 ? 64 : (code == 2 ? 256 : 512));
            f.light_chunk_len = (xp & 0x80000000u) ? f.spp : std::max(f.chunk_len, kLightLen);
            if (f.light_chunk_len > 4095) f.light_chunk_len = 4095;
        }
        if (desc->width > 65535 || desc->height > 65535) { set_error("dsrt_render: rng_mode 1 hands samples between lanes with 16-bit pixel coordinates (width, height <= 65535)"); return DSRT_ERR_INVALID; }
        if ((unsigned long long)f.total_items * 64ull >= (1ull << 32)) { set_error("dsrt_render: image too large for rng_mode 1 (more than 2^32 sample slices)"); return DSRT_ERR_INVALID; }
        f.total_items *= 64u;                                   // upper bound (the pre-pass picks 8 to 64 slices per heavy pixel): sizes the grid only
        const size_t words = out_pixels * 3;
        if (ctx->accum_fixed.n < words) { int rc = ctx->accum_fixed.alloc(words); if (rc) return rc; }
        HIP_TRY(hipMemsetAsync(ctx->accum_fixed.p, 0, words * sizeof(unsigned long long), stream));       // 24 bytes per pixel
        a.accum_fixed = ctx->accum_fixed.p;
    }
    a.out_rgb8 = d_rgb8;
    a.out_f32 = d_f32;
    a.queue = ctx->ctrl.p;
    a.queue_light = ctx->ctrl.p + kQueueLightWord;
    a.flags = ctx->ctrl.p + 1;
    a.counters = (uint64_t*)(ctx->ctrl.p + 4);

    // LDS short-stack: 8 entries per lane (what fits beside the tree top and the continuation strip); deeper entries spill.
    const int K = 8;
    if (desc->stack_entries != 0 && desc->stack_entries != 8) { set_error("dsrt_render: stack_entries must be 0 or 8"); return DSRT_ERR_INVALID; }
    const int threads_per_block = 64 * kernel_waves_per_block();
    const int resident_blocks = ctx->num_cus * 4;                         // 4 waves per SIMD = 4 workgroups of 4 waves per CU (render_kernel.hip)
    int blocks = resident_blocks;                                         // persistent: exactly the resident set
    if ((xp >> 20) & 7u) blocks = std::max(1, resident_blocks >> ((xp >> 20) & 7u));     // experiment: a fraction of it (frames that overlap)
    {
        const long long needed = ((long long)f.total_items + threads_per_block - 1) / threads_per_block;
        if (needed < blocks && !batch) blocks = (int)(needed > 0 ? needed : 1);
    }
    const int spill_entries = sc.view.stack_need > K ? sc.view.stack_need - K : 0;
    const size_t lanes = (size_t)blocks * threads_per_block;
    if (spill_entries > 0 && ctx->spill.n < lanes * (size_t)spill_entries) {
        int rc = ctx->spill.alloc(lanes * (size_t)spill_entries);
        if (rc) return rc;
    }
    a.spill = ctx->spill.p;
    a.spill_stride = (uint32_t)lanes;
    a.spill_entries = spill_entries;
    a.min_walk_iters = desc->tune[0] > 0 ? desc->tune[0] : 64;      // (192 when the rays walk the certified second tree: set below, once a.accel is known)
    a.advance_budget = desc->tune[1] > 0 ? desc->tune[1] : 12;
    a.leaf_ratio4 = desc->tune[2] > 0 ? desc->tune[2] : 10;            // (16 until the node loop looked at its votes every other iteration: profiles/r03/ab_loop_knobs_after_unroll.jsonl)
    a.deal_leaves = (xp & 128u) ? 0 : 1;
    // The certified second tree (pack_scene): used when it is resident, the host has not asked for the plain reference walk, and every camera of the launch is within
    // 30 scene extents of the scene's centre (the widening of the second tree's boxes covers the rounding of (box - origin) up to there; bounce and shadow rays start
    // on the geometry).  Otherwise every ray walks the reference tree, as without the option.
    a.accel = 0;
    if (sc.has_second_tree && !(flags & DSRT_TUNE_REFERENCE_WALK)) {
        auto near_enough = [&](const DsrtF3& o) {
            const double dx = (double)o.x - sc.scene_centre[0], dy = (double)o.y - sc.scene_centre[1], dz = (double)o.z - sc.scene_centre[2];
            return std::sqrt(dx * dx + dy * dy + dz * dz) <= 30.0 * (double)sc.scene_extent;
        };
        bool ok = true;
        if (batch) { for (int i = 0; i < batch->frames && ok; ++i) ok = near_enough(batch->cameras[i].origin); }
        else ok = near_enough(ctx->camera.origin);
        a.accel = ok ? 1 : 0;
    }
    a.audit = a.accel && desc->collect_counters == 3 ? 1 : 0;
    // Walks of the second tree are a third shorter, so an advance pass is dearer against a node iteration than on the reference tree: the traverse phase stays three times
    // longer before it yields (interleaved medians, near frame: 760 -> 741 ms in rng_mode 0, 734 -> 701 in rng_mode 1; the reference walk gains nothing from it).
    if (a.accel && desc->tune[0] <= 0) a.min_walk_iters = 192;
    // ... and its leaves come sooner: the node loop yields to the leaf pass at 0.6 parked lane-slots per descending lane instead of 1.0 (rng_mode 0: near frame -0.5 ... -1.1 %,
    // frame 85 -5 %, frame 92 -2 %, one of 8 shares -1.6 %, far frames unchanged; rng_mode 1 loses 1 % and keeps 1.0: profiles/r04/ab_leaf_ratio_*.jsonl).
    if (a.accel && desc->rng_mode == 0 && desc->tune[2] <= 0) a.leaf_ratio4 = 6;
    a.helpers = (flags & DSRT_TUNE_NO_HELPERS) ? 0 : 1;
    a.steal = ((flags & DSRT_TUNE_NO_STEALING) ? 0 : 1) | (((xp & (1u << 27)) && desc->collect_counters) ? 8 : 0);      // (8: timing image, counting build)
    // rng_mode 0: waves that hold a pixel of a heavy tile get issue priority over waves that only hold background pixels (render_body).
    // Interleaved medians, 1080p x 1000: near frame 1117 -> 1108 ms, frame 95 801 -> 785 ms.  Finer grades (the top quarter and sixteenth of
    // the order above the rest) were tried in round 2 and moved nothing consistently (profiles/r02/ab_issue_priority.jsonl); they are gone.
    a.hot = (flags & DSRT_TUNE_NO_PRIORITY) ? 0 : 1;

    HIP_TRY(hipMemsetAsync(ctx->ctrl.p, 0, kCtrlWords * sizeof(uint32_t), stream));
    // Pre-pass for this camera: costliest-first tile order (scheduling only) and removal of tiles that are provably empty (exact:
    // see dsrt_tile_cost_kernel).  DSRT_TUNE_NATURAL_ORDER switches both off, DSRT_TUNE_NO_CULLING keeps the order but culls nothing; counting builds never
    // cull, so that their counters cover every sample.  The words 32 and 48 entries past the cost array receive the number of
    // tiles that see geometry and the number of tiles in the order.
    const size_t pre_stride = (size_t)t.mine + 64;          // per frame: tile costs / order, then the sched words
    if (pre_stride * (size_t)frames >= ((size_t)1 << 32)) { set_error("dsrt_render_batch: too many tiles in one batch (split the sequence)"); return DSRT_ERR_INVALID; }
    for (DevBuf<uint32_t>* b : {&ctx->tile_cost, &ctx->tile_order, &ctx->tile_work, &ctx->tile_tmp}) {      // each by its own size: a failed
        const size_t want = (b == &ctx->tile_cost || b == &ctx->tile_order) ? pre_stride * (size_t)frames : pre_stride;
        if (b->n < want) { int rc = b->alloc(want); if (rc) return rc; }                                    // allocation leaves no stale sibling
    }
    uint32_t* sched = ctx->tile_cost.p + t.mine + 32;       // {tiles that see geometry, tiles in the order, heavy lanes per wave}
    a.sched = sched;
    if (batch) {
        // Batch: every frame's pre-pass (exact culling + coverage order; no probe: the frames' chains overlap whatever their order), its
        // results left pre_stride apart; then one small kernel turns the counts into the table the render kernel looks work items up in.
        // Slices stay at the host's 8 (resident_lanes = 0: the pool is never short of heavy pixels).
        const bool cull = (flags & 3u) == 0u;
        HIP_TRY(hipMemsetAsync(d_rgb8, 0, out_pixels * 3, stream));                          // culled pixels are never written
        if (d_f32) HIP_TRY(hipMemsetAsync(d_f32, 0, out_pixels * 3 * sizeof(float), stream));
        if (ctx->batch_table.n < 2 * (size_t)frames) { int rc = ctx->batch_table.alloc(2 * (size_t)frames); if (rc) return rc; }
        ctx->batch_host.assign(2 * (size_t)frames, BatchFrame{});
        for (int i = 0; i < frames; ++i) {
            const GPUCamera& bc = batch->cameras[i];
            BatchFrame& e = ctx->batch_host[(size_t)i];
            const float cam12[12] = {bc.origin.x, bc.origin.y, bc.origin.z, bc.lower_left_corner.x, bc.lower_left_corner.y, bc.lower_left_corner.z,
                                     bc.horizontal.x, bc.horizontal.y, bc.horizontal.z, bc.vertical.x, bc.vertical.y, bc.vertical.z};
            static_assert(sizeof cam12 == sizeof e.cam, "camera block");
            std::memcpy(e.cam, cam12, sizeof cam12);
            e.sun_dir[0] = batch->sun_dirs[i].x; e.sun_dir[1] = batch->sun_dirs[i].y; e.sun_dir[2] = batch->sun_dirs[i].z;
            e.order_base = (uint32_t)(pre_stride * (size_t)i);
            e.image_slot = (uint32_t)i;
            ctx->batch_host[(size_t)frames + (size_t)i] = e;                                  // the frame's second entry (device_layout.h, BatchFrame)
        }
        HIP_TRY(hipMemcpyAsync(ctx->batch_table.p, ctx->batch_host.data(), 2 * (size_t)frames * sizeof(BatchFrame), hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_tile_order(a.scene, a.frame, ctx->tile_cost.p, ctx->tile_order.p, sched, (uint32_t)f.chunks, 0u, cull, stream,
                                  ctx->batch_table.p, (uint32_t)frames, (uint32_t)pre_stride));
        // rng_mode 0 at enough samples for a pixel to be a long chain: every frame's heavy tiles re-sorted by measured cost, as for a single
        // frame (the probe launch below in this function) -- one probe per frame, 6 ms each at 1080p, against a frame of a second
        if (!(flags & DSRT_TUNE_NO_PROBE) && desc->rng_mode == 0 && f.spp >= 256) {
            if (ctx->probe_queue.n < 1024) { int rc = ctx->probe_queue.alloc(1024); if (rc) return rc; }
            for (int i = 0; i < frames; ++i) {
                RenderArgs pa = a;
                std::memcpy(pa.frame.cam, ctx->batch_host[(size_t)i].cam, sizeof pa.frame.cam);
                std::memcpy(pa.frame.sun_dir, ctx->batch_host[(size_t)i].sun_dir, 3 * sizeof(float));
                pa.frame.spp = 4; pa.frame.chunks = 1; pa.frame.chunk_len = 4;
                pa.out_f32 = nullptr; pa.accum_fixed = nullptr; pa.counters = nullptr;
                pa.sched = sched + pre_stride * (size_t)i;
                pa.frame.tile_order = ctx->tile_order.p + pre_stride * (size_t)i;
                pa.tile_work = ctx->tile_work.p;
                pa.probe_queue = ctx->probe_queue.p;
                HIP_TRY(hipMemsetAsync(ctx->probe_queue.p, 0, 1024 * sizeof(uint32_t), stream));
                HIP_TRY(hipMemsetAsync(ctx->tile_work.p, 0, (size_t)t.mine * sizeof(uint32_t), stream));
                HIP_TRY(hipMemsetAsync(ctx->ctrl.p, 0, kCtrlWords * sizeof(uint32_t), stream));
                HIP_TRY(libm ? devlibm::launch_probe(pa, blocks, lean, stream) : launch_probe(pa, blocks, lean, stream));
                HIP_TRY(launch_tile_reorder(ctx->tile_work.p, ctx->tile_order.p + pre_stride * (size_t)i, ctx->tile_tmp.p, sched + pre_stride * (size_t)i, stream));
            }
            HIP_TRY(hipMemsetAsync(ctx->ctrl.p, 0, kCtrlWords * sizeof(uint32_t), stream));
        }
        HIP_TRY(launch_batch_table(ctx->batch_table.p, sched, (uint32_t)pre_stride, (uint32_t)frames, (uint32_t)(t.tile * t.tile), desc->rng_mode, f.spp,
                                   f.light_chunk_len, ctx->ctrl.p + 2, stream));
        a.batch = ctx->batch_table.p; a.batch_order = ctx->tile_order.p; a.batch_frames = 2u * (uint32_t)frames;
        a.batch_frame_pixels = (uint32_t)(out_pixels / (size_t)frames);        // whole images, or the padded compact shard buffers, one after another
    } else
    if ((flags & 3u) != DSRT_TUNE_NATURAL_ORDER && t.mine > 0) {
        const bool cull = (flags & 3u) != DSRT_TUNE_NO_CULLING && desc->collect_counters == 0;
        if (cull) {                                             // culled pixels are never written: they are the zeros put here
            HIP_TRY(hipMemsetAsync(d_rgb8, 0, out_pixels * 3, stream));
            if (d_f32) HIP_TRY(hipMemsetAsync(d_f32, 0, out_pixels * 3 * sizeof(float), stream));
        }
        HIP_TRY(launch_tile_order(a.scene, a.frame, ctx->tile_cost.p, ctx->tile_order.p, sched, (uint32_t)f.chunks,
                                  (uint32_t)blocks * (uint32_t)threads_per_block, cull, stream));
        a.frame.tile_order = ctx->tile_order.p;
        // Probe: the render kernel itself at kProbeSpp samples per pixel (reference stream, nothing stored)
        // measures what every tile costs; the heavy tiles are then re-sorted by that.  Worth its 0.5 % only when a pixel is a
        // long chain; DSRT_TUNE_NO_PROBE switches it off.
        constexpr int kProbeSpp = 4;          // x every pixel of the heavy tiles: 7 ms at 1080p, near frame.  (1, 2, 4 or 8 samples order the tiles equally well.)
        // Its work items are tiny, so the probe has 64 queue words of its own (path_machine.h, ST_FETCH): on the frame's single queue word
        // the same launch took 27 ms.
        // Only with the reference's stream: in rng_mode 1 a pixel is cut into slices and lanes share samples, so there is no long chain to start
        // early, and the coverage order alone is better (interleaved medians, near frame: 1046 -> 1037 ms, one of 8 shares 176 -> 167 ms).
        if (!(flags & DSRT_TUNE_NO_PROBE) && desc->rng_mode == 0 && f.spp >= 64 * kProbeSpp) {
            RenderArgs pa = a;
            const int probe_spp = (xp & 64u) ? 2 * kProbeSpp : kProbeSpp;                  // experiment
            pa.frame.spp = probe_spp; pa.frame.chunks = 1; pa.frame.chunk_len = probe_spp;
            pa.out_f32 = nullptr; pa.accum_fixed = nullptr; pa.counters = nullptr;
            pa.tile_work = ctx->tile_work.p;
            if (ctx->probe_queue.n < 1024) { int rc = ctx->probe_queue.alloc(1024); if (rc) return rc; }
            HIP_TRY(hipMemsetAsync(ctx->probe_queue.p, 0, 1024 * sizeof(uint32_t), stream));
            pa.probe_queue = ctx->probe_queue.p;
            HIP_TRY(hipMemsetAsync(ctx->tile_work.p, 0, (size_t)t.mine * sizeof(uint32_t), stream));
            HIP_TRY(libm ? devlibm::launch_probe(pa, blocks, lean, stream) : launch_probe(pa, blocks, lean, stream));
            HIP_TRY(launch_tile_reorder(ctx->tile_work.p, ctx->tile_order.p, ctx->tile_tmp.p, sched, stream));
            HIP_TRY(hipMemsetAsync(ctx->ctrl.p, 0, kCtrlWords * sizeof(uint32_t), stream));
        }
    } else {
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)sched, t.mine, 2, stream));           // every tile in the heavy queue, natural order
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(sched + 2), 64, 1, stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(sched + 3), f.chunks, 1, stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(sched + 4), f.chunk_len, 1, stream));
    }
    if (stats) HIP_TRY(hipEventRecord(ctx->ev0, stream));
    const bool count = desc->collect_counters != 0;
    if (batch) HIP_TRY(libm ? devlibm::launch_render_batch(a, desc->rng_mode, blocks, lean, stream) : launch_render_batch(a, desc->rng_mode, blocks, lean, stream));
    else if (libm) HIP_TRY(devlibm::launch_render(a, K, desc->rng_mode, blocks, count, count || desc->checked != 0, desc->collect_counters != 2, lean, stream));
    else HIP_TRY(launch_render(a, K, desc->rng_mode, blocks, count, count || desc->checked != 0, desc->collect_counters != 2, lean, stream));
    if (desc->rng_mode == 1) HIP_TRY(libm ? devlibm::launch_resolve(a.accum_fixed, f.spp, f.inv_gamma, out_pixels, d_rgb8, d_f32, stream)
                                          : launch_resolve(a.accum_fixed, f.spp, f.inv_gamma, out_pixels, d_rgb8, d_f32, stream));
    HIP_TRY(hipEventRecord(ctx->done, stream));
    ctx->done_valid = true;
    if (stats) {
        HIP_TRY(hipEventRecord(ctx->ev1, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        std::memset(stats, 0, sizeof *stats);
        HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, ctx->ev0, ctx->ev1));
        uint32_t ctrl[kCtrlWords];
        HIP_TRY(hipMemcpy(ctrl, ctx->ctrl.p, sizeof ctrl, hipMemcpyDeviceToHost));
        stats->device_flags = ctrl[1];
        stats->waves_launched = blocks * (threads_per_block / 64);
        stats->lds_stack_entries = K;
        uint64_t cnt[kNumCounters];
        std::memcpy(cnt, &ctrl[4], sizeof cnt);
        stats->samples = cnt[C_SAMPLES]; stats->rays = cnt[C_RAYS]; stats->primary_hits = cnt[C_PRIMARY_HITS];
        stats->box_fetches = cnt[C_BOX_FETCHES]; stats->nodes_entered = cnt[C_NODES_ENTERED]; stats->internal_entered = cnt[C_INTERNAL_ENTERED];
        stats->tri_tests = cnt[C_TRI_TESTS]; stats->hit_updates = cnt[C_HIT_UPDATES]; stats->sphere_tests = cnt[C_SPHERE_TESTS];
        stats->shaded_hits = cnt[C_SHADED_HITS]; stats->tex_fetches = cnt[C_TEX_FETCHES]; stats->stack_spills = cnt[C_STACK_SPILLS];
        stats->max_stack = cnt[C_MAX_STACK];
        stats->node_slots = cnt[C_NODE_SLOTS]; stats->tri_slots = cnt[C_TRI_SLOTS]; stats->adv_slots = cnt[C_ADV_SLOTS]; stats->adv_active = cnt[C_ADV_ACTIVE];
        stats->idle_at_leaf = cnt[C_IDLE_AT_LEAF]; stats->idle_waiting = cnt[C_IDLE_WAITING]; stats->idle_done = cnt[C_IDLE_DONE];
        stats->wave_ticks = cnt[C_WAVE_TICKS];
        stats->certificate_fallbacks = cnt[C_CERT_FALLBACKS];
        stats->certificate_audited = cnt[C_AUDITED]; stats->certificate_audit_mismatches = cnt[C_AUDIT_MISMATCHES];
        stats->certified_tree_used = a.accel;
        {   // time marks relative to the first wave's start, in ms (0 when the mark was never passed)
            const double t0 = (double)~cnt[C_T_FIRST];
            stats->heavy_queue_empty_ms = cnt[C_T_HEAVY_EMPTY] ? (float)(((double)~cnt[C_T_HEAVY_EMPTY] - t0) * 1e-5) : 0.0f;
            stats->light_queue_empty_ms = cnt[C_T_LIGHT_EMPTY] ? (float)(((double)~cnt[C_T_LIGHT_EMPTY] - t0) * 1e-5) : 0.0f;
            stats->last_wave_exit_ms = cnt[C_T_LAST] ? (float)(((double)cnt[C_T_LAST] - t0) * 1e-5) : 0.0f;
        }
        stats->visits_depth_lt6 = cnt[C_VISITS_LT6]; stats->visits_depth_lt9 = cnt[C_VISITS_LT9]; stats->visits_depth_lt12 = cnt[C_VISITS_LT12];
        uint32_t live = 0;
        HIP_TRY(hipMemcpy(&live, sched + 1, sizeof live, hipMemcpyDeviceToHost));
        stats->tiles_total = (uint64_t)t.mine; stats->tiles_culled = (uint64_t)t.mine - live;
        if (stats->device_flags) {
            char buf[96];
            std::snprintf(buf, sizeof buf, "render kernel raised status flags 0x%x", stats->device_flags);
            set_error(buf);
            return DSRT_ERR_DEVICE_FLAG;
        }
    }
    return DSRT_OK;
}

int dsrt_ctx_scene_bounds(const DsrtContext* ctx, float lo[3], float hi[3]) {
    if (!ctx || !lo || !hi) { set_error("dsrt_ctx_scene_bounds: null argument"); return DSRT_ERR_INVALID; }
    if (!ctx->scene || !ctx->scene->valid) { set_error("dsrt_ctx_scene_bounds: no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    for (int a = 0; a < 3; ++a) { lo[a] = ctx->scene->view.root_lo[a]; hi[a] = ctx->scene->view.root_hi[a]; }
    return DSRT_OK;
}

int dsrt_render(DsrtContext* ctx, const DsrtRenderDesc* desc, uint8_t* d_rgb8, float* d_f32, void* stream, DsrtStats* stats) {
    return render_impl(ctx, desc, d_rgb8, d_f32, stream, stats, nullptr);
}

int dsrt_render_batch(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const GPUCamera* cameras, const float* sun_dirs_xyz, uint8_t* d_rgb8, float* d_f32,
                      void* stream, DsrtStats* stats) {
    return dsrt::guarded("dsrt_render_batch", [&]() -> int {
        if (frames < 1 || !cameras || !sun_dirs_xyz) { set_error("dsrt_render_batch: null argument"); return DSRT_ERR_INVALID; }
        std::vector<DsrtF3> suns((size_t)frames);
        for (int i = 0; i < frames; ++i) suns[(size_t)i] = DsrtF3{sun_dirs_xyz[3 * i], sun_dirs_xyz[3 * i + 1], sun_dirs_xyz[3 * i + 2]};
        const BatchInput b{frames, cameras, suns.data()};
        return render_impl(ctx, desc, d_rgb8, d_f32, stream, stats, &b);
    });
}

int dsrt_deinterleave_batch(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const uint8_t* d_gathered, uint8_t* d_rgb8_images, void* stream) {
    Tiling t;
    if (!ctx || !desc || frames < 1 || !d_gathered || !d_rgb8_images || !make_tiling(*desc, t)) { set_error("dsrt_deinterleave_batch: bad argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    const int count = desc->shard_count > 1 ? desc->shard_count : 1;
    const size_t part = (size_t)t.padded * t.tile * t.tile * 3, image = (size_t)desc->width * desc->height * 3;
    // the gather of sharded batch launches: rank r's buffer holds its part of frame 0, of frame 1, ...; rank r + 1's follows `frames` parts on
    for (int f = 0; f < frames; ++f)
        HIP_TRY(launch_deinterleave(d_gathered + (size_t)f * part, d_rgb8_images + (size_t)f * image, desc->width, desc->height, t.tile, t.tiles_x, count,
                                    part * (size_t)frames, (hipStream_t)stream));
    return DSRT_OK;
}

int dsrt_deinterleave_tiles(DsrtContext* ctx, const DsrtRenderDesc* desc, const uint8_t* d_gathered, uint8_t* d_rgb8_image, void* stream) {
    Tiling t;
    if (!ctx || !desc || !d_gathered || !d_rgb8_image || !make_tiling(*desc, t)) { set_error("dsrt_deinterleave_tiles: bad argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    const int count = desc->shard_count > 1 ? desc->shard_count : 1;
    HIP_TRY(launch_deinterleave(d_gathered, d_rgb8_image, desc->width, desc->height, t.tile, t.tiles_x, count, (size_t)t.padded * t.tile * t.tile * 3,
                                (hipStream_t)stream));
    return DSRT_OK;
}

int dsrt_render_batch_to_host(DsrtContext* ctx, const DsrtRenderDesc* desc, int frames, const GPUCamera* cameras, const float* sun_dirs_xyz, uint8_t* h_rgb8,
                              DsrtStats* stats) {
    return dsrt::guarded("dsrt_render_batch_to_host", [&]() -> int {
    if (!ctx || !desc || !h_rgb8 || frames < 1) { set_error("dsrt_render_batch_to_host: null argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)desc->width * desc->height * 3 * (size_t)frames;
    DevBuf<uint8_t> d8;
    int rc = d8.alloc(bytes);
    if (rc) return rc;
    DsrtStats local;
    rc = dsrt_render_batch(ctx, desc, frames, cameras, sun_dirs_xyz, d8.p, nullptr, nullptr, stats ? stats : &local);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h_rgb8, d8.p, bytes, hipMemcpyDeviceToHost));
    return DSRT_OK;
    });
}

int dsrt_render_to_host(DsrtContext* ctx, const DsrtRenderDesc* desc, uint8_t* h_rgb8, float* h_f32, DsrtStats* stats) {
    return dsrt::guarded("dsrt_render_to_host", [&]() -> int {
    if (!ctx || !desc || !h_rgb8) { set_error("dsrt_render_to_host: null argument"); return DSRT_ERR_INVALID; }
    if (desc->shard_count > 1) { set_error("dsrt_render_to_host renders whole images only"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t px = (size_t)desc->width * desc->height;
    DevBuf<uint8_t> d8;
    DevBuf<float> d32;
    int rc = d8.alloc(px * 3);
    if (rc) return rc;
    if (h_f32 && (rc = d32.alloc(px * 3))) return rc;
    DsrtStats local;
    rc = dsrt_render(ctx, desc, d8.p, d32.p, nullptr, stats ? stats : &local);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h_rgb8, d8.p, px * 3, hipMemcpyDeviceToHost));
    if (h_f32) HIP_TRY(hipMemcpy(h_f32, d32.p, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return DSRT_OK;
    });
}

int dsrt_selftest_poke_node_word(DsrtContext* ctx, size_t word_index, uint32_t value, uint32_t* old_value) {
    if (!ctx || !ctx->scene || !ctx->scene->valid) { set_error("dsrt_selftest_poke_node_word: no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    if (word_index >= ctx->scene->pairs.n * 4) { set_error("dsrt_selftest_poke_node_word: word index beyond the node records"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    uint32_t* w = reinterpret_cast<uint32_t*>(ctx->scene->pairs.p) + word_index;
    if (old_value) HIP_TRY(hipMemcpy(old_value, w, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(w, &value, 4, hipMemcpyHostToDevice));
    return DSRT_OK;
}

int dsrt_selftest_math(DsrtContext* ctx, int fn, const float* x, float y, float* out, int n) {
    if (!ctx || !x || !out || n <= 0 || fn < 0 || fn > 2) { set_error("dsrt_selftest_math: bad argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<float> dx, dy;
    int rc;
    if ((rc = dx.alloc((size_t)n)) || (rc = dy.alloc((size_t)n))) return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(launch_math(fn, dx.p, y, dy.p, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dy.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return DSRT_OK;
}

int dsrt_selftest_devkat(DsrtContext* ctx, int fn, const float* in12, float* out12, int n) {
    if (!ctx || !in12 || !out12 || n <= 0 || fn < 0 || fn > 5) { set_error("dsrt_selftest_devkat: bad argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<float> di, dout;
    int rc;
    if ((rc = di.alloc((size_t)n * 12)) || (rc = dout.alloc((size_t)n * 12))) return rc;
    HIP_TRY(hipMemcpy(di.p, in12, (size_t)n * 12 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dout.p, 0, (size_t)n * 12 * sizeof(float)));
    HIP_TRY(launch_devkat(fn, di.p, dout.p, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out12, dout.p, (size_t)n * 12 * sizeof(float), hipMemcpyDeviceToHost));
    return DSRT_OK;
}

int dsrt_selftest_philox(DsrtContext* ctx, uint64_t seed, uint64_t subsequence, int n, uint32_t* ours, uint32_t* rocrand_words) {
    if (!ctx || !ours || !rocrand_words || n <= 0) { set_error("dsrt_selftest_philox: bad argument"); return DSRT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<uint32_t> a, b;
    int rc;
    if ((rc = a.alloc((size_t)n)) || (rc = b.alloc((size_t)n))) return rc;
    HIP_TRY(launch_philox(seed, subsequence, n, a.p, b.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ours, a.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rocrand_words, b.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DSRT_OK;
}

// =========================================================================================
// Drop-in layer
// =========================================================================================
static void gpu_render_scene_body(const GPUScene* scene, int width, int height);
namespace {
std::mutex g_dropin_mutex;
DsrtContext* g_dropin_ctx = nullptr;

// The reference calls build_gpu_scene + gpu_render_scene + free_gpu_scene once per FRAME (src/main.cpp:405-428) although only camera
// and sun change between frames.  gpu_render_scene receives device arrays in the reference layouts; converting them (copy to the
// host, re-layout, upload) costs far more than rendering a far frame.  So the drop-in keeps the converted scene of the previous call
// and re-uses it when the arrays it is handed have the same CONTENT: a 128-bit position-dependent hash of every array, computed on
// the device (one pass over ~150 MB at 1 M triangles: well under a millisecond), plus all the counts.  Pointer equality would not do --
// the reference frees and re-allocates the arrays every frame, so equal pointers can hold different data and vice versa.
struct SceneFingerprint {
    uint64_t h[2] = {0, 0};
    int counts[7] = {0, 0, 0, 0, 0, 0, 0};
    bool valid = false;
    bool operator==(const SceneFingerprint& o) const { return valid && o.valid && h[0] == o.h[0] && h[1] == o.h[1] && !std::memcmp(counts, o.counts, sizeof counts); }
};
SceneFingerprint g_dropin_fp;
// two words of device memory for the hash; like g_dropin_ctx a raw pointer that is never freed: a destructor at static-destruction time
// would call hipFree after the HIP runtime may already be gone
uint64_t* g_dropin_hash = nullptr;

int fingerprint_device_scene(const GPUScene& d, SceneFingerprint& fp) {
    fp = SceneFingerprint{};
    const int counts[7] = {d.num_triangles, d.num_spheres, d.num_materials, d.num_bvh_nodes, d.num_textures, d.texture_pool_floats, d.tri_indices ? 1 : 0};
    std::memcpy(fp.counts, counts, sizeof counts);
    for (int c : counts) if (c < 0) { set_error("scene has a negative count"); return DSRT_ERR_INVALID; }
    if (!g_dropin_hash) HIP_TRY(hipMalloc((void**)&g_dropin_hash, 2 * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(g_dropin_hash, 0, 2 * sizeof(uint64_t), nullptr));
    struct Arr { const void* p; size_t bytes; } arrs[7] = {
        {d.triangles, (size_t)d.num_triangles * sizeof(GPUTriangle)}, {d.spheres, (size_t)d.num_spheres * sizeof(GPUSphere)},
        {d.materials, (size_t)d.num_materials * sizeof(GPUMaterial)}, {d.bvh_nodes, (size_t)d.num_bvh_nodes * sizeof(GPUBVHNode)},
        {d.textures, (size_t)d.num_textures * sizeof(GPUTextureHeader)}, {d.texture_pool, (size_t)d.texture_pool_floats * sizeof(float)},
        {d.tri_indices, d.tri_indices ? (size_t)d.num_triangles * sizeof(int) : 0}};
    for (int a = 0; a < 7; ++a)
        if (arrs[a].p && arrs[a].bytes) HIP_TRY(launch_content_hash((const uint32_t*)arrs[a].p, arrs[a].bytes / 4, 0x9E3779B97F4A7C15ull * (uint64_t)(a + 1), g_dropin_hash, nullptr));
    HIP_TRY(hipMemcpy(fp.h, g_dropin_hash, sizeof fp.h, hipMemcpyDeviceToHost));
    fp.valid = true;
    return DSRT_OK;
}

DsrtContext* dropin_context() {
    if (!g_dropin_ctx) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (dsrt_ctx_create(dev, &g_dropin_ctx) != DSRT_OK) return nullptr;
    }
    return g_dropin_ctx;
}
}  // namespace

int dsrt_dropin_has_certified_tree(void) {
    std::lock_guard<std::mutex> lock(g_dropin_mutex);
    return g_dropin_ctx && g_dropin_ctx->scene && g_dropin_ctx->scene->valid && g_dropin_ctx->scene->has_second_tree ? 1 : 0;
}

int dsrt_build_gpu_scene(const DsrtHostScene* hs, const GPUCamera* cam, const float sun_dir_model[3], GPUScene* out) {
    return dsrt::guarded("dsrt_build_gpu_scene", [&]() -> int {
    if (!hs || !cam || !out) { set_error("dsrt_build_gpu_scene: null argument"); return DSRT_ERR_INVALID; }
    GPUScene h;
    int rc = dsrt_host_scene_view(hs, &h);
    if (rc) return rc;
    dsrt_scene_set_frame(&h, cam, sun_dir_model);
    // The returned header carries DEVICE arrays in the reference layouts, like the reference's own builder.
    GPUScene d = h;
    auto push = [&](const void* src, size_t bytes, void** dst) -> int {
        *dst = nullptr;
        if (!bytes) return DSRT_OK;
        HIP_TRY(hipMalloc(dst, bytes));
        HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return DSRT_OK;
    };
    void* p = nullptr;
    if ((rc = push(h.triangles, (size_t)h.num_triangles * sizeof(GPUTriangle), &p))) return rc; d.triangles = (const GPUTriangle*)p;
    if ((rc = push(h.spheres, (size_t)h.num_spheres * sizeof(GPUSphere), &p))) return rc; d.spheres = (const GPUSphere*)p;
    if ((rc = push(h.materials, (size_t)h.num_materials * sizeof(GPUMaterial), &p))) return rc; d.materials = (const GPUMaterial*)p;
    if ((rc = push(h.tri_indices, (size_t)(h.tri_indices ? h.num_triangles : 0) * sizeof(int), &p))) return rc; d.tri_indices = (const int*)p; d.bvh_tri_indices = (int*)p;
    if ((rc = push(h.bvh_nodes, (size_t)h.num_bvh_nodes * sizeof(GPUBVHNode), &p))) return rc; d.bvh_nodes = (GPUBVHNode*)p;
    if ((rc = push(h.textures, (size_t)h.num_textures * sizeof(GPUTextureHeader), &p))) return rc; d.textures = (const GPUTextureHeader*)p;
    if ((rc = push(h.texture_pool, (size_t)h.texture_pool_floats * sizeof(float), &p))) return rc; d.texture_pool = (const float*)p;
    *out = d;
    return DSRT_OK;
    });
}

void dsrt_free_gpu_scene(GPUScene* s) {                                  // src/gpu_scene_builder.cpp:603-626
    if (!s) return;
    if (s->triangles) (void)hipFree((void*)s->triangles);
    if (s->spheres) (void)hipFree((void*)s->spheres);
    if (s->materials) (void)hipFree((void*)s->materials);
    if (s->bvh_nodes) (void)hipFree((void*)s->bvh_nodes);
    if (s->tri_indices) (void)hipFree((void*)s->tri_indices);
    if (s->textures) (void)hipFree((void*)s->textures);
    if (s->texture_pool) (void)hipFree((void*)s->texture_pool);
    s->triangles = nullptr; s->num_triangles = 0;
    s->spheres = nullptr; s->num_spheres = 0;
    s->materials = nullptr; s->num_materials = 0;
    s->bvh_nodes = nullptr; s->num_bvh_nodes = 0;
    s->tri_indices = nullptr; s->bvh_tri_indices = nullptr;
    s->textures = nullptr; s->num_textures = 0;
    s->texture_pool = nullptr; s->texture_pool_floats = 0;
}

void gpu_render_scene(const GPUScene* scene, int width, int height) {
    (void)dsrt::guarded("gpu_render_scene", [&]() -> int { gpu_render_scene_body(scene, width, height); return DSRT_OK; });
}

static void gpu_render_scene_body(const GPUScene* scene, int width, int height) {
    std::lock_guard<std::mutex> lock(g_dropin_mutex);
    if (!scene) { std::fprintf(stderr, "gpu_render_scene: null scene\n"); return; }
    DsrtContext* ctx = dropin_context();
    if (!ctx) { std::fprintf(stderr, "gpu_render_scene: %s\n", dsrt_last_error()); return; }
    SceneFingerprint fp;
    if (fingerprint_device_scene(*scene, fp) != DSRT_OK) { std::fprintf(stderr, "gpu_render_scene: %s\n", dsrt_last_error()); return; }
    // The reference's entry point has nowhere to ask for the certified second tree either: DSRT_CERTIFIED_TREE=1, looked at on every call here (the one context of the
    // drop-in lives as long as the process); a change of the variable re-converts the scene.
    { const char* e = std::getenv("DSRT_CERTIFIED_TREE"); ctx->want_second_tree = e && e[0] == '1'; }
    if (ctx->scene && ctx->scene->valid && fp == g_dropin_fp && ctx->scene->has_second_tree == (ctx->want_second_tree && ctx->scene->view.root_ref != kRefNone)) {
        // same geometry, materials and textures as the previous call: only the per-frame part of the header is taken over
        ctx->camera = scene->camera;
        ctx->sun_dir = scene->sun_dir; ctx->sun_radiance = scene->sun_radiance; ctx->sun_enabled = scene->sun_enabled ? 1 : 0;
    } else {
        g_dropin_fp.valid = false;
        if (dsrt_scene_upload_device(ctx, scene) != DSRT_OK) { std::fprintf(stderr, "gpu_render_scene: scene upload failed: %s\n", dsrt_last_error()); return; }
        g_dropin_fp = fp;
    }
    DsrtRenderDesc d;
    std::memset(&d, 0, sizeof d);
    d.width = width; d.height = height;
    d.spp = scene->params.samples_per_pixel;
    d.max_depth = scene->params.max_depth;
    d.gamma = scene->params.gamma;
    d.seed = scene->seed;
    d.rng_mode = scene->params.rng_mode == 1 ? 1 : 0;        // the reference always stores 0 here (src/gpu_scene_builder.cpp:577)
    // The reference's entry point has nowhere to say which sinf / cosf / powf (DsrtRenderDesc.math_mode): the environment variable DSRT_MATH_MODE=1 asks this
    // drop-in for the device math library's, i.e. for the very file the reference's own gpu_render_scene writes on this GPU (tests/test_gpu_reference_kernel.py)
    { const char* e = std::getenv("DSRT_MATH_MODE"); d.math_mode = e && std::atoi(e) == 1 ? 1 : 0; }
    std::vector<uint8_t> fb((size_t)width * height * 3);
    if (dsrt_render_to_host(ctx, &d, fb.data(), nullptr, nullptr) != DSRT_OK) { std::fprintf(stderr, "render_kernel failed: %s\n", dsrt_last_error()); return; }
    if (dsrt_write_ppm("image_gpu.ppm", fb.data(), width, height) != DSRT_OK) std::fprintf(stderr, "Failed to open image_gpu.ppm for writing\n");
}

}  // extern "C"

// C++ forms of the reference's builder entry points (declared in host/scene_model.hpp).
namespace dsrt {

GPUScene build_gpu_scene(const hittable_list& world, const camera& cam, const vec3& sun_dir_model) {
    GPUScene out;
    std::memset(&out, 0, sizeof out);
    DsrtHostScene* hs = dsrt_host_scene_create();
    const float sun[3] = {sun_dir_model.x(), sun_dir_model.y(), sun_dir_model.z()};
    const GPUCamera gc = cam.toGPUCamera();
    if (flatten_world(world, hs) != DSRT_OK || dsrt_host_scene_build_bvh(hs) != DSRT_OK || dsrt_build_gpu_scene(hs, &gc, sun, &out) != DSRT_OK)
        std::fprintf(stderr, "build_gpu_scene: %s\n", dsrt_last_error());
    dsrt_host_scene_destroy(hs);
    return out;
}

void free_gpu_scene(GPUScene& scene) { dsrt_free_gpu_scene(&scene); }

}  // namespace dsrt
