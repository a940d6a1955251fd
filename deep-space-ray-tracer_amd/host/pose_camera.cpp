// pose_camera.cpp -- pose-.txt reader, world->model transform, and the pinhole camera.
//
// Reference behaviour reproduced (file:line in the reference repository):
//   src/main.cpp:139-173   one pose per non-empty, non-'#' line: nine whitespace-separated doubles
//                          cam_xyz model_xyz yaw pitch roll; a line that does not yield nine numbers is skipped;
//                          yaw/pitch/roll are narrowed to float when stored.
//   src/main.cpp:334-357   cam_rel = cam - model and light_rel = 0 - model in double; both rotated about +Y by
//                          -yaw (degrees; rad = deg * pi / 180 with pi = 3.1415926535897932385, inc/rtweekend.h:27-34);
//                          narrowed to float only at the end; sun_dir = normalize(light_in_model) in double, then float;
//                          frames with |cam_rel| < 1 m are skipped (:342-345).
//   src/main.cpp:178-187   vup = (0,1,0), focus_dist = |lookfrom - lookat| (float), then camera::initialize.
//   inc/camera.h:91-116    all-float basis construction (tanf of half the vertical field of view).
// Every expression keeps the reference's association order; this file is compiled with -ffp-contract=off.
#include <cmath>
#include <fstream>
#include <sstream>

#include "host_internal.hpp"

namespace dsrt {

void camera::initialize() {
    const float aspect = float(image_width) / float(image_height);
    const float theta = (float)((double)vfov * 3.1415926535897932385 / 180.0);
    const float half = tanf(theta / 2.0f);
    const float viewport_h = 2.0f * half;
    const float viewport_w = aspect * viewport_h;

    w = unit_vector(lookfrom - lookat);
    u = unit_vector(cross(vup, w));
    v = cross(w, u);

    origin = lookfrom;
    horizontal = focus_dist * viewport_w * u;
    vertical = focus_dist * viewport_h * v;
    lower_left_corner = origin - horizontal * 0.5f - vertical * 0.5f - focus_dist * w;
    lens_radius = aperture * 0.5f;
}

GPUCamera camera::toGPUCamera() const {
    GPUCamera g;
    auto put = [](DsrtF3& d, const vec3& s) { d.x = s.x(); d.y = s.y(); d.z = s.z(); };
    put(g.origin, origin);
    put(g.lower_left_corner, lower_left_corner);
    put(g.horizontal, horizontal);
    put(g.vertical, vertical);
    put(g.u, u);
    put(g.v, v);
    put(g.w, w);
    g.lens_radius = lens_radius;
    g.image_width = image_width;
    g.image_height = image_height;
    g.samples_per_pixel = samples_per_pixel;
    g.max_depth = max_depth;
    return g;
}

}  // namespace dsrt

namespace {

struct D3 { double x, y, z; };

D3 yaw_about_y(const D3& p, double yaw_deg) {
    const double rad = yaw_deg * 3.1415926535897932385 / 180.0;
    const double c = std::cos(rad), s = std::sin(rad);
    return D3{c * p.x + s * p.z, p.y, -s * p.x + c * p.z};
}

}  // namespace

extern "C" {

int dsrt_read_pose_file(const char* path, DsrtPose* out, int cap, int* count) {
    return dsrt::guarded("dsrt_read_pose_file", [&]() -> int {
    if (!path || !count || cap < 0 || (cap > 0 && !out)) { dsrt::set_error("dsrt_read_pose_file: bad argument"); return DSRT_ERR_INVALID; }
    *count = 0;
    std::ifstream in(path);
    if (!in) { dsrt::set_error(std::string("cannot open pose file ") + path); return DSRT_ERR_IO; }
    std::string line;
    int n = 0;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream iss(line);
        double v[9];
        bool ok = true;
        for (double& q : v) if (!(iss >> q)) { ok = false; break; }
        if (!ok) continue;
        if (n < cap) {
            DsrtPose& p = out[n];
            for (int k = 0; k < 3; ++k) { p.cam_pos_world[k] = v[k]; p.model_pos_world[k] = v[3 + k]; p.model_euler_deg[k] = (float)v[6 + k]; }
        }
        ++n;
    }
    *count = n;
    if (n == 0) { dsrt::set_error(std::string("no valid pose in ") + path); return DSRT_ERR_IO; }
    return DSRT_OK;
    });
}

int dsrt_pose_to_frame(const DsrtPose* pose, DsrtFrame* out) {
    if (!pose || !out) { dsrt::set_error("dsrt_pose_to_frame: null argument"); return DSRT_ERR_INVALID; }
    const double yaw_deg = (double)pose->model_euler_deg[0];
    const D3 cam_rel{pose->cam_pos_world[0] - pose->model_pos_world[0], pose->cam_pos_world[1] - pose->model_pos_world[1],
                     pose->cam_pos_world[2] - pose->model_pos_world[2]};
    const D3 light_rel{0.0 - pose->model_pos_world[0], 0.0 - pose->model_pos_world[1], 0.0 - pose->model_pos_world[2]};
    out->sep_m = std::sqrt(cam_rel.x * cam_rel.x + cam_rel.y * cam_rel.y + cam_rel.z * cam_rel.z);
    out->skipped = out->sep_m < 1.0 ? 1 : 0;
    const D3 cam_m = yaw_about_y(cam_rel, -yaw_deg);
    const D3 light_m = yaw_about_y(light_rel, -yaw_deg);
    out->cam_in_model[0] = (float)cam_m.x; out->cam_in_model[1] = (float)cam_m.y; out->cam_in_model[2] = (float)cam_m.z;
    const double L = std::sqrt(light_m.x * light_m.x + light_m.y * light_m.y + light_m.z * light_m.z);
    D3 sun{0.0, 0.0, 0.0};
    if (L != 0.0) { const double inv = 1.0 / L; sun = D3{inv * light_m.x, inv * light_m.y, inv * light_m.z}; }
    out->sun_dir_model[0] = (float)sun.x; out->sun_dir_model[1] = (float)sun.y; out->sun_dir_model[2] = (float)sun.z;
    return DSRT_OK;
}

int dsrt_camera_look_at(GPUCamera* out, const float from[3], const float at[3], float vfov_deg, int width, int height, int spp,
                        int max_depth) {
    if (!out || !from || !at || width < 2 || height < 2) { dsrt::set_error("dsrt_camera_look_at: bad argument"); return DSRT_ERR_INVALID; }
    dsrt::camera cam;
    cam.image_width = width;
    cam.image_height = height;
    cam.samples_per_pixel = spp;
    cam.max_depth = max_depth;
    cam.vfov = vfov_deg;
    cam.aperture = 0.0f;
    cam.lookfrom = dsrt::vec3(from[0], from[1], from[2]);
    cam.lookat = dsrt::vec3(at[0], at[1], at[2]);
    cam.vup = dsrt::vec3(0.0f, 1.0f, 0.0f);
    cam.focus_dist = (cam.lookfrom - cam.lookat).length();
    cam.initialize();
    *out = cam.toGPUCamera();
    return DSRT_OK;
}

}  // extern "C"
