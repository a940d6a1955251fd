// reference_entry_points.cpp -- the reference's two C++ entry points, at LINK level, on top of libdsrt_hip.so.
//
// The reference's main() (src/main.cpp:405-427) calls
//     GPUScene build_gpu_scene(const hittable_list& world, const camera& cam, const vec3& sun_dir_model);   inc/gpu_scene_builder.h:72
//     void     free_gpu_scene(GPUScene& scene);                                                             inc/gpu_scene_builder.h:73
//     extern "C" void gpu_render_scene(const GPUScene& scene, int width, int height);                       src/main.cpp:24-25
// with ITS OWN classes (inc/hittable_list.h, triangle_mesh.h, sphere.h, material.h, camera.h, vec3.h).  The third symbol is exported by the library
// as it is.  The first two take C++ class types that cannot cross a C ABI, so they are provided here, as the one file a maintainer of the reference
// adds to the build IN PLACE OF src/gpu_scene_builder.cpp and src/gpu_render.cu:
//
//     g++ -std=c++17 -I<reference>/inc -I<cuda include dir, for float3> -I<this repo>/include \
//         <reference>/src/main.cpp <reference>/src/stb_image_impl.cpp <this repo>/integration/reference_entry_points.cpp -ldsrt_hip
//
// (oracle/Makefile, target _ref/ref_main_on_dsrt, does exactly that in this container; tests/test_gpu_reference_main.py runs the result on the GPU box.)
// This file is compiled AGAINST THE REFERENCE'S HEADERS: every scene type below -- hittable_list, triangle_mesh, triangle, sphere, material and its four
// subclasses, camera, vec3, and the POD structs GPUScene / GPUTriangle / GPUSphere / GPUMaterial / GPUCamera -- is the reference's own definition.
// include/dsrt_scene_abi.h declares the same POD structs for hosts that do not have the reference's headers (layouts pinned against them,
// tests/test_host_golden.py); here its include guard is pre-defined so that include/dsrt.h is read with the reference's definitions instead.
//
// What it does: walks the world in the order the reference's collect step does (src/gpu_scene_builder.cpp:252-308: meshes triangle by triangle, single
// triangles, spheres, nested lists depth-first; one material slot per distinct material object, src/gpu_scene_builder.cpp:73-139), hands the flattened
// arrays to the library (dsrt_host_scene_add_arrays / dsrt_host_scene_add_texture_file), lets the library build the reference's median-split BVH
// (dsrt_host_scene_build_bvh) and upload (dsrt_build_gpu_scene).  Nothing of the library's own OBJ / MTL loader is involved: the reference's
// triangle_mesh has already read the file.
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtweekend.h"
#include "camera.h"
#include "hittable_list.h"
#include "triangle_mesh.h"
#include "material.h"
#include "sphere.h"
#include "gpu_scene.h"
#include "gpu_scene_builder.h"

#define DSRT_SCENE_ABI_H            /* the scene structs are the reference's own (see above) */
#include "dsrt.h"

static_assert(sizeof(GPUScene) == 384 && sizeof(GPUTriangle) == 116 && sizeof(GPUSphere) == 24 && sizeof(GPUMaterial) == 48 && sizeof(GPUCamera) == 104,
              "the library was built for these layouts (include/dsrt_scene_abi.h)");

namespace {

float3 as_float3(const vec3& v) { return make_float3((float)v.x(), (float)v.y(), (float)v.z()); }

class WorldFlattener {
public:
    explicit WorldFlattener(DsrtHostScene* scene) : scene_(scene) {}

    void visit(const std::shared_ptr<hittable>& node) {
        if (!node) return;
        if (auto mesh = std::dynamic_pointer_cast<triangle_mesh>(node)) { take_mesh(*mesh); return; }
        if (auto tri = std::dynamic_pointer_cast<triangle>(node)) { take_triangle(*tri, -1); return; }
        if (auto ball = std::dynamic_pointer_cast<sphere>(node)) { take_sphere(*ball); return; }
        if (auto group = std::dynamic_pointer_cast<hittable_list>(node)) {
            for (const auto& child : group->objects) visit(child);
        }
        // anything else has no GPU form and is skipped, as the reference skips it
    }

    int flush() {
        return dsrt_host_scene_add_arrays(scene_, triangles_.data(), (int)triangles_.size(), spheres_.data(), (int)spheres_.size(),
                                          materials_.data(), (int)materials_.size());
    }
    bool has_triangles() const { return !triangles_.empty(); }

private:
    // One table entry per distinct material OBJECT; a missing material is a fresh 0.8-gray lambertian every time it is met.
    int slot_of(const std::shared_ptr<material>& m) {
        GPUMaterial entry{};
        entry.albedo_tex = -1;
        entry.emissive = make_float3(0.0f, 0.0f, 0.0f);
        entry.fuzz = 0.0f;
        entry.ref_idx = 1.5f;
        if (!m) {
            entry.type = MAT_LAMBERTIAN;
            entry.albedo = make_float3(0.8f, 0.8f, 0.8f);
            materials_.push_back(entry);
            return (int)materials_.size() - 1;
        }
        const auto seen = slots_.find(m.get());
        if (seen != slots_.end()) return seen->second;
        if (const auto* diffuse = dynamic_cast<const lambertian*>(m.get())) {
            entry.type = MAT_LAMBERTIAN;
            entry.albedo = as_float3(diffuse->albedo_value());
        } else if (const auto* mirror = dynamic_cast<const metal*>(m.get())) {
            entry.type = MAT_METAL;
            entry.albedo = as_float3(mirror->albedo_value());
            entry.fuzz = (float)mirror->fuzz_value();
        } else if (const auto* glass = dynamic_cast<const dielectric*>(m.get())) {
            entry.type = MAT_DIELECTRIC;
            entry.albedo = make_float3(1.0f, 1.0f, 1.0f);
            entry.ref_idx = (float)glass->ior_value();
        } else if (const auto* lamp = dynamic_cast<const diffuse_light*>(m.get())) {
            entry.type = MAT_DIFFUSE_LIGHT;
            entry.albedo = make_float3(1.0f, 1.0f, 1.0f);
            entry.emissive = as_float3(lamp->emit_value());
            entry.ref_idx = 1.0f;
        } else {
            entry.type = MAT_LAMBERTIAN;
            entry.albedo = make_float3(0.73f, 0.73f, 0.73f);
        }
        materials_.push_back(entry);
        const int slot = (int)materials_.size() - 1;
        slots_.emplace(m.get(), slot);
        return slot;
    }

    void take_triangle(const triangle& t, int texture) {
        const int slot = slot_of(t.mat);
        if (texture >= 0) materials_[slot].albedo = make_float3(1.0f, 1.0f, 1.0f);      // a textured triangle whitens its material's albedo
        GPUTriangle g{};
        g.v0 = as_float3(t.v0); g.v1 = as_float3(t.v1); g.v2 = as_float3(t.v2);
        g.n0 = as_float3(t.n0); g.n1 = as_float3(t.n1); g.n2 = as_float3(t.n2);
        g.uv0 = make_float3((float)t.uv0.x(), (float)t.uv0.y(), 0.0f);
        g.uv1 = make_float3((float)t.uv1.x(), (float)t.uv1.y(), 0.0f);
        g.uv2 = make_float3((float)t.uv2.x(), (float)t.uv2.y(), 0.0f);
        g.material_id = slot;
        g.albedo_tex = texture;
        triangles_.push_back(g);
    }

    void take_mesh(const triangle_mesh& mesh) {
        for (size_t i = 0; i < mesh.triangles.size(); ++i) {
            int texture = -1;
            if (i < mesh.tri_map_Kd.size() && !mesh.tri_map_Kd[i].empty()) {
                // the reference's stb flag is "flip" from the moment its MTL reader met a map (inc/texture.h:133): every file decoded here is
                texture = dsrt_host_scene_add_texture_file(scene_, mesh.tri_map_Kd[i].c_str(), 1);
                if (texture < 0) texture = -1;
            }
            take_triangle(mesh.triangles[i], texture);
        }
    }

    void take_sphere(const sphere& s) {
        GPUSphere g{};
        g.center = as_float3(s.static_center());
        g.radius = (float)s.get_radius();
        g.material_id = slot_of(s.get_material());
        g._pad = 0;
        spheres_.push_back(g);
    }

    DsrtHostScene* scene_;
    std::vector<GPUTriangle> triangles_;
    std::vector<GPUSphere> spheres_;
    std::vector<GPUMaterial> materials_;
    std::unordered_map<const material*, int> slots_;
};

void complain(const char* where) { std::fprintf(stderr, "%s: %s\n", where, dsrt_last_error()); }

}  // namespace

GPUScene build_gpu_scene(const hittable_list& world, const camera& cam, const vec3& sun_dir_model) {
    GPUScene scene{};
    DsrtHostScene* host = dsrt_host_scene_create();
    if (!host) { complain("build_gpu_scene"); return scene; }
    WorldFlattener flat(host);
    for (const auto& object : world.objects) flat.visit(object);
    const GPUCamera gpu_cam = cam.toGPUCamera();
    const float sun[3] = {(float)sun_dir_model.x(), (float)sun_dir_model.y(), (float)sun_dir_model.z()};
    if (flat.flush() != DSRT_OK || (flat.has_triangles() && dsrt_host_scene_build_bvh(host) != DSRT_OK) ||
        dsrt_build_gpu_scene(host, &gpu_cam, sun, &scene) != DSRT_OK) {
        complain("build_gpu_scene");
        scene = GPUScene{};                   // an empty scene renders black, like a reference scene without geometry
    }
    dsrt_host_scene_destroy(host);
    return scene;
}

void free_gpu_scene(GPUScene& scene) { dsrt_free_gpu_scene(&scene); }
