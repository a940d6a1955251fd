// scene_model.hpp -- host-side scene objects with the reference's names and meaning.
//
// The reference's main() assembles a `hittable_list` of `triangle_mesh` / `sphere` / `triangle` objects
// carrying `lambertian` / `metal` / `dielectric` / `diffuse_light` materials, points a `camera`, and calls
//     GPUScene build_gpu_scene(const hittable_list&, const camera&, const vec3& sun_dir_model);
//     void     free_gpu_scene(GPUScene&);                         (inc/gpu_scene_builder.h:72-73)
//     extern "C" void gpu_render_scene(const GPUScene&, int, int); (src/main.cpp:24-25)
// This header provides those names so that such a driver compiles against this library unchanged.
// Unlike the reference's book-style classes (inc/hittable.h, sphere.h, triangle.h, material.h ...) these
// are plain data carriers: there is no virtual hit()/scatter() here, because nothing on the GPU path ever
// calls one -- the builder only reads geometry and material parameters out of them
// (src/gpu_scene_builder.cpp:39-139, 252-308).  The numerics that DO matter for bit parity are kept:
//   * vec3 is float[3] and `v / t` multiplies by 1.0f/t            (inc/vec3.h:14-22, 49-55, 99-103)
//   * triangle flat normal = unit_vector(cross(v1-v0, v2-v0))       (inc/triangle.h:70-73)
//   * camera::initialize arithmetic order                           (inc/camera.h:91-116)
#pragma once

#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dsrt.h"

namespace dsrt {

struct vec3 {
    float e[3];
    vec3() : e{0.0f, 0.0f, 0.0f} {}
    vec3(float a, float b, float c) : e{a, b, c} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float operator[](int i) const { return e[i]; }
    float& operator[](int i) { return e[i]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return sqrtf(length_squared()); }
};
using point3 = vec3;
using color = vec3;

inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(float t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, float t) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator/(const vec3& v, float t) { float inv = 1.0f / t; return vec3(v.e[0] * inv, v.e[1] * inv, v.e[2] * inv); }
inline float dot(const vec3& a, const vec3& b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

// ---- materials: parameters only ---------------------------------------------------------------
struct material {
    int   type = MAT_LAMBERTIAN;        // MaterialType, inc/gpu_scene.h:21-26
    vec3  albedo{0.73f, 0.73f, 0.73f};
    vec3  emissive{0.0f, 0.0f, 0.0f};
    float fuzz = 0.0f;
    float ref_idx = 1.5f;
    virtual ~material() = default;
};
// Values below are what upsert_material writes for each class (src/gpu_scene_builder.cpp:96-126).
struct lambertian : material {
    explicit lambertian(const color& a) { type = MAT_LAMBERTIAN; albedo = a; }
    // textured form: albedo_value() of a non-solid texture is the gray fallback (inc/material.h:99-105)
    struct textured_tag {};
    explicit lambertian(textured_tag) { type = MAT_LAMBERTIAN; albedo = color(0.8f, 0.8f, 0.8f); }
};
struct metal : material {
    metal(const color& a, double f) { type = MAT_METAL; albedo = a; fuzz = (float)(f < 1 ? f : 1); }   // inc/material.h:117-118
};
struct dielectric : material {
    explicit dielectric(double ior) { type = MAT_DIELECTRIC; albedo = color(1.0f, 1.0f, 1.0f); ref_idx = (float)ior; }
};
struct diffuse_light : material {
    explicit diffuse_light(const color& c) { type = MAT_DIFFUSE_LIGHT; albedo = color(1.0f, 1.0f, 1.0f); emissive = c; ref_idx = 1.0f; }
};

// ---- hittables: geometry only -----------------------------------------------------------------
struct hittable {
    enum class kind { sphere, triangle, mesh, list };
    virtual ~hittable() = default;
    virtual kind what() const = 0;
};

struct sphere : hittable {
    point3 center;
    double radius;
    std::shared_ptr<material> mat;
    sphere(const point3& c, double r, std::shared_ptr<material> m) : center(c), radius(std::fmax(0.0, r)), mat(std::move(m)) {}
    kind what() const override { return kind::sphere; }
};

struct triangle : hittable {
    vec3 v0, v1, v2, n0, n1, n2, uv0, uv1, uv2;
    std::shared_ptr<material> mat;
    triangle() = default;
    triangle(const vec3& a, const vec3& b, const vec3& c, std::shared_ptr<material> m) : v0(a), v1(b), v2(c), mat(std::move(m)) { flat_normal(); }
    triangle(const vec3& a, const vec3& b, const vec3& c, const vec3& ta, const vec3& tb, const vec3& tc, std::shared_ptr<material> m)
        : v0(a), v1(b), v2(c), uv0(ta), uv1(tb), uv2(tc), mat(std::move(m)) { flat_normal(); }
    kind what() const override { return kind::triangle; }
private:
    void flat_normal() { n0 = n1 = n2 = unit_vector(cross(v1 - v0, v2 - v0)); }
};

// OBJ + MTL mesh.  Loader semantics follow inc/triangle_mesh.h:75-255 (implemented in obj_mesh.cpp).
struct triangle_mesh : hittable {
    std::vector<vec3> verts;
    std::vector<vec3> uvs;                    // (u, 1-v, 0)
    std::vector<triangle> triangles;
    std::vector<std::string> tri_map_Kd;      // per-triangle diffuse texture path, "" if none
    std::shared_ptr<material> fallback;
    bool loaded = false;                      // false: file could not be opened (the reference silently yields an empty mesh)
    triangle_mesh(const std::string& obj_path, std::shared_ptr<material> fallback_mat, double scale = 1.0);
    kind what() const override { return kind::mesh; }
};

struct hittable_list : hittable {
    std::vector<std::shared_ptr<hittable>> objects;
    hittable_list() = default;
    explicit hittable_list(std::shared_ptr<hittable> o) { add(std::move(o)); }
    void clear() { objects.clear(); }
    void add(std::shared_ptr<hittable> o) { objects.push_back(std::move(o)); }
    kind what() const override { return kind::list; }
};

// ---- camera (inc/camera.h:66-134) ---------------------------------------------------------------
class camera {
public:
    int   image_width = 800;
    int   image_height = 450;
    int   samples_per_pixel = 10;
    int   max_depth = 50;
    vec3  lookfrom, lookat;
    vec3  vup = vec3(0.0f, 1.0f, 0.0f);
    float vfov = 40.0f;
    float aperture = 0.0f;
    float focus_dist = 1.0f;
    vec3  origin, horizontal, vertical, lower_left_corner, u, v, w;
    float lens_radius = 0.0f;

    void initialize();
    GPUCamera toGPUCamera() const;
};

// ---- the reference's three entry points, C++ forms ------------------------------------------------
// build: flatten + BVH + upload (device pointers in the returned header); caller owns it and must free it.
GPUScene build_gpu_scene(const hittable_list& world, const camera& cam, const vec3& sun_dir_model);
void free_gpu_scene(GPUScene& scene);

// Flatten a world into a DsrtHostScene (the CPU half of build_gpu_scene), for callers that want to keep the
// scene resident across frames instead of rebuilding it per frame as src/main.cpp:405 does.
int flatten_world(const hittable_list& world, DsrtHostScene* into);

}  // namespace dsrt
