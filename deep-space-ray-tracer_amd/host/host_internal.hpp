// host_internal.hpp -- declarations shared by the host-side translation units (not installed).
#pragma once

#include <cstdint>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <cstdlib>
#include <thread>
#include <vector>

#include "scene_model.hpp"

namespace dsrt {

void set_error(const std::string& msg);

// No exception crosses the C ABI: the bodies of the extern "C" entry points that allocate or parse run inside this.
template <class F>
int guarded(const char* where, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { set_error(std::string(where) + ": out of memory"); } catch (...) {} return -8; /* DSRT_ERR_NOMEM */ }
    catch (const std::exception& e) { try { set_error(std::string(where) + ": " + e.what()); } catch (...) {} return -1; /* DSRT_ERR_INVALID */ }
    catch (...) { try { set_error(std::string(where) + ": unknown exception"); } catch (...) {} return -1; }
}

struct RgbImage {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
bool load_rgb8(const std::string& path, bool flip_vertically, RgbImage& img);
bool decode_jpeg(const std::vector<uint8_t>& file_bytes, RgbImage& img);      // jpeg_decode.cpp
bool decode_bmp(const std::vector<uint8_t>& file_bytes, RgbImage& img);       // bmp_tga_decode.cpp
bool decode_tga(const std::vector<uint8_t>& file_bytes, RgbImage& img);

// half-width by which the non-parity BVH builders (bvh_sah.cpp, csrc/bvh_lbvh.hip) widen a triangle box that has zero thickness on an axis
inline float flat_box_pad(float scene_extent) { return scene_extent > 0.0f ? scene_extent * (1.0f / 4096.0f) : 1.0e-6f; }

// threads the host BVH builders may use: the machine's, or DSRT_BUILD_THREADS if the environment sets it (1 = build on the calling thread)
inline unsigned builder_threads() {
    if (const char* e = std::getenv("DSRT_BUILD_THREADS")) { const long v = std::strtol(e, nullptr, 10); if (v >= 1) return (unsigned)v; }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? hw : 1u;
}

// bvh_sah.cpp: the binned-SAH builder on plain arrays (every triangle box widened by pad_all if > 0; triangles marked in `skip` left out)
int build_sah_tree(const GPUTriangle* tris, size_t n, float pad_all, const uint8_t* skip, std::vector<GPUBVHNode>& nodes, std::vector<int>& order, int& height_out, int leaf_max);

// The host half of the certified second tree (include/dsrt.h, dsrt_ctx_set_certified_tree; bvh_sah.cpp): from a scene with the reference's tree,
//   unreachable[t]   1 = triangle t lies under a zero-thickness box of the reference tree: bbox_hit can never pass there (src/gpu_render.cu:312), the reference never reaches it
//   leaf_box[6 t..]  lo, hi of the leaf of the reference tree that holds triangle t
//   nodes, order     a binned-SAH tree over the REACHABLE triangles, every triangle box widened by pad = extent * 2^-16 (order = its tri_indices)
//   origins_near     no sphere of the scene reaches beyond 30 extents of the mesh (rays also start on spheres)
struct SecondTree {
    std::vector<uint8_t> unreachable;
    std::vector<float> leaf_box;
    std::vector<GPUBVHNode> nodes;
    std::vector<int> order;
    int height = 0;
    float extent = 0.0f, pad = 0.0f;
    bool origins_near = true;
};
int prepare_second_tree(const GPUScene& h, int leaf_max, SecondTree& out);

bool texture_flip_latch();
void texture_flip_latch_set(bool v);

}  // namespace dsrt

// The flattened scene: exactly the vectors build_gpu_scene uploads (src/gpu_scene_builder.cpp:183-190, 490-546).
struct DsrtHostScene {
    std::vector<GPUTriangle> tris;
    std::vector<GPUSphere> spheres;
    std::vector<GPUMaterial> mats;
    std::vector<int> tri_indices;
    std::vector<GPUBVHNode> nodes;
    std::vector<GPUTextureHeader> tex_headers;
    std::vector<float> tex_pool;

    std::unordered_map<const dsrt::material*, int> mat_index;      // material object -> table slot
    std::unordered_map<std::string, int> tex_index;                // texture path -> header slot
    std::vector<std::string> tex_failed;                           // paths that could not be decoded (their slot is the reference's 1x1 white)
    std::vector<std::shared_ptr<dsrt::material>> keep_alive;       // keeps mat_index keys unique for the scene's lifetime
    bool bvh_valid = false;
    int bvh_height = 0;                                            // levels; the traversal stack needs height - 1 entries
};
