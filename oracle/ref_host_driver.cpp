// ref_host_driver.cpp -- drives the REFERENCE'S OWN host code to produce golden vectors.
//
// TEST INFRASTRUCTURE ONLY (see dsrt_oracle.h).  This file is ours; everything it calls is the
// reference's, compiled from the sources where they lie (the three #includes below).  Nothing from
// the reference is copied into this repository: the only things committed are the OUTPUTS
// (tests/golden/*, made by tests/golden/make_golden.py running the binary built from this file).
//
// Reference code exercised (file:line in /root/reference):
//   src/main.cpp               read_pose_file :139-173, rotate_yaw_deg_d :118-128, dnormalize :78-82,
//                              to_float_vec3 :84-86, point_camera_at :178-187 (static functions; `main`
//                              itself is renamed away and garbage-collected at link time)
//   inc/camera.h               camera::initialize :91-116, toGPUCamera :118-133
//   inc/triangle_mesh.h        OBJ / MTL loader :75-255
//   src/gpu_scene_builder.cpp  collect_world :310-317, upsert_material :71-139, HostTextureRegistry :199-246,
//                              build_bvh_for_triangles :444-459  (build_gpu_scene / free_gpu_scene, the
//                              only functions needing libcudart, are never referenced and are dropped by
//                              --gc-sections)
//   inc/aabb.h, sphere.h, triangle.h, hittable_list.h    the CPU hit classes, for ray-level cross-checks
//   inc/vec3.h reflect / refract :136-147, inc/material.h reflectance :28-32, metal::scatter :123-137,
//   dielectric::scatter :153-180, inc/onb.h build_from_w :47-56     material / frame known answers (cmd_matkat)
//
// NOT exercised, because it cannot be built here: src/gpu_render.cu (the CUDA kernel).

#define main dsrt_unused_reference_main
#include "src/main.cpp"
#undef main
#include "src/gpu_scene_builder.cpp"
#include "src/stb_image_impl.cpp"

#include <cinttypes>
#include <cstring>
#include <map>

static void hex_bytes(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) std::printf("%02x", b[i]);
}
static uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

template <typename T>
static void dump_vec(const std::string& path, const std::vector<T>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    if (!v.empty()) std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}

#include "ref_world_loader.hpp"

static int cmd_scene(const std::string& world_path, const std::string& prefix) {
    hittable_list world = load_world(world_path);
    HostBuild B;
    HostTextureRegistry texreg;
    collect_world(world, B, texreg);
    std::vector<int> idx;
    std::vector<GPUBVHNode> nodes;
    build_bvh_for_triangles(B.h_tris, idx, nodes);

    std::vector<GPUTextureHeader> headers;
    std::vector<float> pool;
    int off = 0;
    for (const auto& ht : texreg.textures) {              // as src/gpu_scene_builder.cpp:513-535
        GPUTextureHeader h{}; h.width = ht.width; h.height = ht.height; h.offset = off;
        headers.push_back(h);
        pool.insert(pool.end(), ht.data.begin(), ht.data.end());
        off += (int)ht.data.size();
    }
    dump_vec(prefix + ".tris.bin", B.h_tris);
    dump_vec(prefix + ".spheres.bin", B.h_spheres);
    dump_vec(prefix + ".mats.bin", B.h_mats);
    dump_vec(prefix + ".idx.bin", idx);
    dump_vec(prefix + ".nodes.bin", nodes);
    dump_vec(prefix + ".texhdr.bin", headers);
    dump_vec(prefix + ".texpool.bin", pool);
    std::printf("{\"num_triangles\": %zu, \"num_spheres\": %zu, \"num_materials\": %zu, \"num_bvh_nodes\": %zu, "
                "\"num_textures\": %zu, \"texture_pool_floats\": %zu}\n",
                B.h_tris.size(), B.h_spheres.size(), B.h_mats.size(), nodes.size(), headers.size(), pool.size());
    return 0;
}

static void print_camera_json(const camera& cam) {
    GPUCamera g = cam.toGPUCamera();
    std::printf("\"gpu_camera\": \"");
    hex_bytes(&g, sizeof g);
    std::printf("\"");
}

static int cmd_camera(int argc, char** argv) {
    if (argc < 13) return 2;
    camera cam;
    cam.image_width = std::atoi(argv[9]);
    cam.image_height = std::atoi(argv[10]);
    cam.samples_per_pixel = std::atoi(argv[11]);
    cam.max_depth = std::atoi(argv[12]);
    cam.vfov = (float)std::atof(argv[8]);
    cam.aperture = 0.0;
    vec3 from((float)std::atof(argv[2]), (float)std::atof(argv[3]), (float)std::atof(argv[4]));
    vec3 at((float)std::atof(argv[5]), (float)std::atof(argv[6]), (float)std::atof(argv[7]));
    point_camera_at(cam, from, at);                        // src/main.cpp:178-187
    std::printf("{");
    print_camera_json(cam);
    std::printf("}\n");
    return 0;
}

// The per-frame arithmetic of src/main.cpp:310-399, restated as CALLS to the reference's own functions.
static int cmd_poses(int argc, char** argv) {
    if (argc < 8) return 2;
    std::vector<PoseEntry> poses;
    if (!read_pose_file(argv[2], poses)) { std::printf("{\"error\": \"no poses\"}\n"); return 1; }
    camera cam;
    cam.image_width = std::atoi(argv[3]);
    cam.image_height = std::atoi(argv[4]);
    cam.samples_per_pixel = std::atoi(argv[5]);
    cam.max_depth = std::atoi(argv[6]);
    cam.vfov = (float)std::atof(argv[7]);
    cam.aperture = 0.0;
    dvec3 light_pos_world_d(0.0, 0.0, 0.0);
    std::printf("[\n");
    for (size_t i = 0; i < poses.size(); ++i) {
        const PoseEntry& p = poses[i];
        double yaw_deg = p.model_euler_deg.x();
        dvec3 cam_rel_world_d = p.cam_pos_world_d - p.model_pos_world_d;
        dvec3 light_rel_world_d = light_pos_world_d - p.model_pos_world_d;
        double sep_m = dlength(cam_rel_world_d);
        bool skipped = sep_m < 1.0;
        dvec3 cam_in_model_d = rotate_yaw_deg_d(cam_rel_world_d, -yaw_deg);
        dvec3 light_in_model_d = rotate_yaw_deg_d(light_rel_world_d, -yaw_deg);
        vec3 cam_in_model = to_float_vec3(cam_in_model_d);
        dvec3 sun_dir_model_d = dnormalize(light_in_model_d);
        vec3 sun_dir_model = to_float_vec3(sun_dir_model_d);
        std::printf("{\"frame\": %zu, \"skipped\": %s, \"sep_m\": %.17g, "
                    "\"cam_in_model\": [%" PRIu32 ", %" PRIu32 ", %" PRIu32 "], "
                    "\"sun_dir_model\": [%" PRIu32 ", %" PRIu32 ", %" PRIu32 "], ",
                    i, skipped ? "true" : "false", sep_m,
                    fbits(cam_in_model.x()), fbits(cam_in_model.y()), fbits(cam_in_model.z()),
                    fbits(sun_dir_model.x()), fbits(sun_dir_model.y()), fbits(sun_dir_model.z()));
        if (!skipped) point_camera_at(cam, cam_in_model, vec3(0, 0, 0));
        print_camera_json(cam);
        std::printf("}%s\n", i + 1 < poses.size() ? "," : "");
    }
    std::printf("]\n");
    return 0;
}

static int cmd_abi() {
#define SZ(T) std::printf("\"sizeof_" #T "\": %zu, ", sizeof(T))
#define OFF(T, m) std::printf("\"" #T "." #m "\": %zu, ", offsetof(T, m))
    std::printf("{");
    SZ(GPUTextureHeader); SZ(GPUMaterial); SZ(GPUSphere); SZ(GPUTriangle); SZ(GPUBVHNode);
    SZ(GPURenderParams); SZ(GPUCamera); SZ(GPUScene);
    OFF(GPUMaterial, albedo); OFF(GPUMaterial, emissive); OFF(GPUMaterial, fuzz); OFF(GPUMaterial, ref_idx);
    OFF(GPUSphere, radius); OFF(GPUSphere, material_id);
    OFF(GPUTriangle, v0); OFF(GPUTriangle, n0); OFF(GPUTriangle, uv0); OFF(GPUTriangle, material_id); OFF(GPUTriangle, albedo_tex);
    OFF(GPUBVHNode, bbox_min); OFF(GPUBVHNode, bbox_max); OFF(GPUBVHNode, left); OFF(GPUBVHNode, right);
    OFF(GPUBVHNode, tri_offset); OFF(GPUBVHNode, tri_count);
    OFF(GPURenderParams, samples_per_pixel); OFF(GPURenderParams, max_depth); OFF(GPURenderParams, rng_mode);
    OFF(GPURenderParams, tile_size); OFF(GPURenderParams, gamma); OFF(GPURenderParams, exposure);
    OFF(GPUCamera, origin); OFF(GPUCamera, lower_left_corner); OFF(GPUCamera, horizontal); OFF(GPUCamera, vertical);
    OFF(GPUCamera, u); OFF(GPUCamera, lens_radius); OFF(GPUCamera, image_width); OFF(GPUCamera, max_depth);
    OFF(GPUScene, spheres); OFF(GPUScene, num_spheres); OFF(GPUScene, triangles); OFF(GPUScene, tri_indices);
    OFF(GPUScene, num_triangles); OFF(GPUScene, bvh_nodes); OFF(GPUScene, num_bvh_nodes); OFF(GPUScene, bvh_tri_indices);
    OFF(GPUScene, materials); OFF(GPUScene, num_materials); OFF(GPUScene, textures); OFF(GPUScene, num_textures);
    OFF(GPUScene, texture_pool); OFF(GPUScene, texture_pool_floats); OFF(GPUScene, camera); OFF(GPUScene, sky_type);
    OFF(GPUScene, env_tex_id); OFF(GPUScene, sky_solid); OFF(GPUScene, sky_top); OFF(GPUScene, sky_bottom);
    OFF(GPUScene, params); OFF(GPUScene, seed); OFF(GPUScene, sun_enabled); OFF(GPUScene, sun_dir); OFF(GPUScene, sun_radiance);
    std::printf("\"end\": 0}\n");
    return 0;
}

// Ray-level known answers from the reference's CPU classes.  Rays come from the same 32-bit LCG the
// kernel uses (so the test can regenerate them); values are printed as float/double bit patterns.
static float lcg01(uint32_t& s) { s = s * 1664525u + 1013904223u; return (s & 0x00FFFFFFu) / 16777216.0f; }
static int cmd_hitkat(int n) {
    uint32_t s = 2024u;
    auto m = std::make_shared<lambertian>(color(0.5, 0.5, 0.5));
    sphere sph(point3(0.25f, -0.125f, -1.0f), 0.5, m);
    triangle tri(vec3(-1.0f, -0.5f, -2.0f), vec3(1.5f, -0.25f, -2.5f), vec3(0.0f, 1.25f, -1.5f), m);
    aabb box(point3(-0.5f, -0.25f, -1.75f), point3(0.75f, 0.5f, -1.0f));
    std::printf("[\n");
    for (int i = 0; i < n; ++i) {
        vec3 o(lcg01(s) - 0.5f, lcg01(s) - 0.5f, lcg01(s) * 0.5f);
        vec3 d(lcg01(s) * 2.0f - 1.0f, lcg01(s) * 2.0f - 1.0f, -(lcg01(s) + 0.25f));
        if (i % 7 == 3) d = vec3(0.0f, d.y(), d.z());     // axis-parallel component: the 1/0 slab path
        ray r(o, d);
        hit_record rs, rt;
        bool hs = sph.hit(r, interval(0.001, 1e9), rs);
        bool ht = tri.hit(r, interval(0.001, 1e9), rt);
        bool hb = box.hit(r, interval(0.001, 1e9));
        uint64_t ts = 0, tt = 0;
        if (hs) std::memcpy(&ts, &rs.t, 8);
        if (ht) std::memcpy(&tt, &rt.t, 8);
        std::printf("{\"o\": [%u, %u, %u], \"d\": [%u, %u, %u], \"sphere\": %d, \"sphere_t\": %" PRIu64
                    ", \"tri\": %d, \"tri_t\": %" PRIu64 ", \"box\": %d}%s\n",
                    fbits(o.x()), fbits(o.y()), fbits(o.z()), fbits(d.x()), fbits(d.y()), fbits(d.z()),
                    hs ? 1 : 0, ts, ht ? 1 : 0, tt, hb ? 1 : 0, i + 1 < n ? "," : "");
    }
    std::printf("]\n");
    return 0;
}

// Known answers from the reference's own DEVICE helpers, which compile on the host (CUDA_D is empty off nvcc, inc/cuda_compat.h):
//   random_float_device              inc/rtweekend.h:126-133   the LCG (same constants as src/gpu_render.cu:77-80)
//   random_in_unit_sphere_device     :164-171                  the rejection loop (candidates -1 + 2 r: exact in float and in double)
//   random_cosine_direction_device   :190-202                  cosine direction in local coordinates, DOUBLE math rounded to float
//   generate_camera_ray_device       inc/camera.h:35-61        ray set-up, lens_radius = 0
// The kernel does not call these (it has its own float copies, src/gpu_render.cu:77-109, 941-968), but they are the
// same formulas executed by reference code, which is what the oracle's rand01 / rejection loop / cosine direction / camera ray can be
// pinned against: bit for bit where the arithmetic is float or exact, to rounding where the reference helper computes in double.
static int cmd_devkat(int n) {
    std::printf("{\"lcg\": [\n");
    const uint32_t seeds[4] = {1337u, 0u, 0xFFFFFFFFu, 123456789u};
    for (int k = 0; k < 4; ++k) {
        uint32_t s = seeds[k];
        std::printf("  {\"seed\": %u, \"draws\": [", seeds[k]);
        for (int i = 0; i < 16; ++i) { float f = random_float_device(s); std::printf("[%u, %u]%s", s, fbits(f), i < 15 ? ", " : ""); }
        std::printf("]}%s\n", k < 3 ? "," : "");
    }
    std::printf("],\n\"unit_sphere\": [\n");
    for (int i = 0; i < n; ++i) {
        uint32_t s0 = 1000003u * (uint32_t)i + 17u, s = s0;
        vec3 p = random_in_unit_sphere_device(s);
        std::printf("  {\"state_in\": %u, \"state_out\": %u, \"p\": [%u, %u, %u]}%s\n", s0, s, fbits(p.x()), fbits(p.y()), fbits(p.z()), i + 1 < n ? "," : "");
    }
    std::printf("],\n\"cosine_direction\": [\n");
    for (int i = 0; i < n; ++i) {
        uint32_t s0 = 7919u * (uint32_t)i + 5u, s = s0;
        vec3 d = random_cosine_direction_device(s);
        std::printf("  {\"state_in\": %u, \"state_out\": %u, \"d\": [%u, %u, %u]}%s\n", s0, s, fbits(d.x()), fbits(d.y()), fbits(d.z()), i + 1 < n ? "," : "");
    }
    std::printf("],\n\"camera_ray\": [\n");
    camera cam;
    cam.image_width = 200; cam.image_height = 112; cam.samples_per_pixel = 4; cam.max_depth = 5; cam.vfov = 40; cam.aperture = 0.0;
    point_camera_at(cam, vec3(-0.72611237f, 0.0f, 1786.9741f), vec3(0, 0, 0));
    const GPUCamera g = cam.toGPUCamera();
    std::printf("  {\"camera\": \""); hex_bytes(&g, sizeof g); std::printf("\", \"rays\": [\n");
    for (int i = 0; i < n; ++i) {
        uint32_t s0 = 2654435761u * (uint32_t)(i + 1), s = s0;
        const int px = (i * 37) % 200, py = (i * 11) % 112;
        uint32_t sj = s0;
        const float jx = (float)random_double_device(sj), jy = (float)random_double_device(sj);      // the jitter the helper is about to draw
        ray r = generate_camera_ray_device(g, px, py, s);
        std::printf("    {\"px\": %d, \"py\": %d, \"state_in\": %u, \"state_out\": %u, \"jx\": %u, \"jy\": %u, \"orig\": [%u, %u, %u], \"dir\": [%u, %u, %u]}%s\n",
                    px, py, s0, s, fbits(jx), fbits(jy), fbits(r.origin().x()), fbits(r.origin().y()), fbits(r.origin().z()),
                    fbits(r.direction().x()), fbits(r.direction().y()), fbits(r.direction().z()), i + 1 < n ? "," : "");
    }
    std::printf("  ]}\n]}\n");
    return 0;
}

// Known answers from the reference's MATERIAL and FRAME helpers, executed: the host counterparts of what ray_color's specular branches and
// the cosine sampler are made of (src/gpu_render.cu:112-118, 195-212, 603-661).  vec3 is float-based (inc/vec3.h:14-70: unit_vector is
// `v * (1.0f / length)`, like the kernel's f3_norm), so wherever the helper's own arithmetic is float the answers are bit patterns:
//   reflect                 inc/vec3.h:136-139      float: bit for bit
//   refract                 inc/vec3.h:141-147      float: bit for bit (the kernel's copy re-normalises its argument first, :199-206;
//                                                   inputs here come out of unit_vector, the test compares where that is idempotent)
//   metal::scatter, fuzz 0  inc/material.h:123-137  reflect(unit_vector(dir), n) + 0.0f * (a rand() vector): the direction and the
//                                                   accept test dot(dir, n) > 0
//   dielectric::scatter     inc/material.h:153-180  on the total-internal-reflection branch only (the decision is taken in double there:
//                                                   cases keep ratio * sin_theta at least 5 % above 1; the other branch draws rand());
//                                                   what is pinned: the direction, and that NO random number is drawn
//   reflectance             inc/material.h:28-32    Schlick in double: the kernel's float copy to rounding
//   onb::build_from_w       inc/onb.h:47-56         w and v bit for bit; u = cross(w, v) is the NEGATIVE of the kernel's cross(v, w)
static int cmd_matkat(int n) {
    uint32_t s = 4242u;
    auto sgn = [&](float m) { return (lcg01(s) * 2.0f - 1.0f) * m; };
    std::printf("{\"reflect\": [\n");
    for (int i = 0; i < n; ++i) {
        const vec3 v(sgn(3.0f), sgn(3.0f), sgn(3.0f));
        const vec3 nn = unit_vector(vec3(sgn(1.0f), sgn(1.0f), sgn(1.0f) + 1e-3f));
        const vec3 r = reflect(v, nn);
        std::printf("  {\"v\": [%u, %u, %u], \"n\": [%u, %u, %u], \"out\": [%u, %u, %u]}%s\n", fbits(v.x()), fbits(v.y()), fbits(v.z()), fbits(nn.x()), fbits(nn.y()),
                    fbits(nn.z()), fbits(r.x()), fbits(r.y()), fbits(r.z()), i + 1 < n ? "," : "");
    }
    std::printf("],\n\"refract\": [\n");
    const float etas[6] = {1.0f / 1.5f, 1.5f, 1.0f / 1.33f, 1.0f, 2.4f, 1.0f / 2.4f};
    for (int i = 0; i < n; ++i) {
        const vec3 uv = unit_vector(vec3(sgn(1.0f), sgn(1.0f), sgn(1.0f) + 1e-3f));
        vec3 nn = unit_vector(vec3(sgn(1.0f), sgn(1.0f), sgn(1.0f) + 1e-3f));
        if (dot(uv, nn) > 0.0f) nn = -nn;                                  // the normal faces the incoming ray, as set_face_normal leaves it
        const float eta = etas[i % 6];
        const vec3 r = refract(uv, nn, eta);
        std::printf("  {\"uv\": [%u, %u, %u], \"n\": [%u, %u, %u], \"eta\": %u, \"out\": [%u, %u, %u]}%s\n", fbits(uv.x()), fbits(uv.y()), fbits(uv.z()), fbits(nn.x()),
                    fbits(nn.y()), fbits(nn.z()), fbits(eta), fbits(r.x()), fbits(r.y()), fbits(r.z()), i + 1 < n ? "," : "");
    }
    std::printf("],\n\"metal_fuzz0\": [\n");
    {
        metal m(color(0.8f, 0.6f, 0.2f), 0.0);
        for (int i = 0; i < n; ++i) {
            const vec3 d(sgn(40.0f), sgn(40.0f), sgn(40.0f));              // ray directions are NOT unit length in this renderer (camera rays: |dir| = focus distance)
            const vec3 nn = unit_vector(vec3(sgn(1.0f), sgn(1.0f), sgn(1.0f) + 1e-3f));      // either side: the accept test must see both signs
            hit_record rec;
            rec.p = point3(sgn(5.0f), sgn(5.0f), sgn(5.0f));
            rec.normal = nn;
            scatter_record srec;
            const bool ok = m.scatter(ray(point3(0, 0, 0), d), rec, srec);
            const vec3 o = srec.specular_ray.direction();
            std::printf("  {\"dir\": [%u, %u, %u], \"n\": [%u, %u, %u], \"out\": [%u, %u, %u], \"ok\": %d}%s\n", fbits(d.x()), fbits(d.y()), fbits(d.z()), fbits(nn.x()),
                        fbits(nn.y()), fbits(nn.z()), fbits(o.x()), fbits(o.y()), fbits(o.z()), ok ? 1 : 0, i + 1 < n ? "," : "");
        }
    }
    std::printf("],\n\"dielectric_tir\": [\n");
    {
        int emitted = 0;
        for (int i = 0; emitted < n && i < 100 * n; ++i) {
            const bool front = (i & 1) != 0;
            const double ir = front ? 0.55 : 1.5;                          // ratio = 1 / 0.55 = 1.82 (front face) or 1.5 (leaving the glass)
            const vec3 d(sgn(40.0f), sgn(40.0f), sgn(40.0f));
            vec3 nn = unit_vector(vec3(sgn(1.0f), sgn(1.0f), sgn(1.0f) + 1e-3f));
            const vec3 ud = unit_vector(d);
            if (dot(ud, nn) > 0.0f) nn = -nn;
            const double ratio = front ? 1.0 / ir : ir;
            const double c = fmin((double)dot(-ud, nn), 1.0), sn = std::sqrt(1.0 - c * c);
            if (!(ratio * sn > 1.05)) continue;                            // well inside the total-internal-reflection regime
            dielectric g(ir);
            hit_record rec;
            rec.p = point3(sgn(5.0f), sgn(5.0f), sgn(5.0f));
            rec.normal = nn;
            rec.front_face = front;
            scatter_record srec;
            (void)g.scatter(ray(point3(0, 0, 0), d), rec, srec);
            const vec3 o = srec.specular_ray.direction();
            const float irf = (float)ir;
            ++emitted;
            std::printf("  {\"dir\": [%u, %u, %u], \"n\": [%u, %u, %u], \"front\": %d, \"ref_idx\": %u, \"out\": [%u, %u, %u]}%s\n", fbits(d.x()), fbits(d.y()), fbits(d.z()),
                        fbits(nn.x()), fbits(nn.y()), fbits(nn.z()), front ? 1 : 0, fbits(irf), fbits(o.x()), fbits(o.y()), fbits(o.z()), emitted < n ? "," : "");
        }
    }
    std::printf("],\n\"reflectance\": [\n");
    for (int i = 0; i < n; ++i) {
        const float c = lcg01(s), ratio = etas[i % 6];
        const double r = reflectance((double)c, (double)ratio);
        uint64_t rb; std::memcpy(&rb, &r, 8);
        std::printf("  {\"cos\": %u, \"ratio\": %u, \"out_f64\": %" PRIu64 "}%s\n", fbits(c), fbits(ratio), rb, i + 1 < n ? "," : "");
    }
    std::printf("],\n\"onb\": [\n");
    for (int i = 0; i < n; ++i) {
        vec3 nn(sgn(2.0f), sgn(2.0f), sgn(2.0f) + 1e-3f);
        if (i % 5 == 0) nn = vec3(i % 10 ? 1.0f : -1.0f, sgn(0.3f), sgn(0.3f));          // |w.x| > 0.9: the other helper axis
        onb f;
        f.build_from_w(nn);
        std::printf("  {\"n\": [%u, %u, %u], \"u\": [%u, %u, %u], \"v\": [%u, %u, %u], \"w\": [%u, %u, %u]}%s\n", fbits(nn.x()), fbits(nn.y()), fbits(nn.z()),
                    fbits(f.u().x()), fbits(f.u().y()), fbits(f.u().z()), fbits(f.v().x()), fbits(f.v().y()), fbits(f.v().z()), fbits(f.w().x()), fbits(f.w().y()),
                    fbits(f.w().z()), i + 1 < n ? "," : "");
    }
    std::printf("]}\n");
    return 0;
}

// The reference's texture decoder on one file: stbi_load(path, &w, &h, &n, 3), the call of src/gpu_scene_builder.cpp:215
// (the flip flag of inc/texture.h:133 is a separate, global switch; `flip` sets it like image_texture::load does).
static int cmd_decode(const char* path, int flip) {
    stbi_set_flip_vertically_on_load(flip);
    int w = 0, h = 0, n = 0;
    unsigned char* img = stbi_load(path, &w, &h, &n, 3);
    if (!img) { std::printf("{\"ok\": 0}\n"); return 0; }
    std::printf("{\"ok\": 1, \"w\": %d, \"h\": %d, \"file_channels\": %d, \"rgb\": \"", w, h, n);
    hex_bytes(img, (size_t)w * h * 3);
    std::printf("\"}\n");
    stbi_image_free(img);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: ref_host abi | scene <world.txt> <prefix> | camera fx fy fz ax ay az vfov W H spp depth |"
                             " poses <pose.txt> W H spp depth vfov | hitkat <n> | devkat <n> | matkat <n> | decode <image> [flip]\n");
        return 2;
    }
    std::string c = argv[1];
    if (c == "abi") return cmd_abi();
    if (c == "scene" && argc >= 4) return cmd_scene(argv[2], argv[3]);
    if (c == "camera") return cmd_camera(argc, argv);
    if (c == "poses") return cmd_poses(argc, argv);
    if (c == "hitkat" && argc >= 3) return cmd_hitkat(std::atoi(argv[2]));
    if (c == "devkat" && argc >= 3) return cmd_devkat(std::atoi(argv[2]));
    if (c == "matkat" && argc >= 3) return cmd_matkat(std::atoi(argv[2]));
    if (c == "decode" && argc >= 3) return cmd_decode(argv[2], argc >= 4 ? std::atoi(argv[3]) : 0);
    return 2;
}
