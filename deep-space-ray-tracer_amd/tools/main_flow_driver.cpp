// main_flow_driver -- the reference's main() flow against this library's C++ boundary, one frame.
//
// What src/main.cpp does per frame, with the reference's own class names (scene_model.hpp provides them):
//     triangle_mesh(obj, lambertian(0.73), 1.0)          src/main.cpp:238-245
//     camera cam; width/height/spp/depth/vfov/aperture   :254-260
//     pose -> cam_in_model, sun_dir_model                :310-357   (dsrt_read_pose_file / dsrt_pose_to_frame)
//     point_camera_at(cam, cam_in_model, origin)         :178-187, :399
//     GPUScene s = build_gpu_scene(world, cam, sun)      :405
//     gpu_render_scene(s, W, H)  -> image_gpu.ppm        :413
//     rename image_gpu.ppm -> frame file                 :425
//     free_gpu_scene(s)                                  :428
// Exists so that the C++ entry points (dsrt::build_gpu_scene / dsrt::free_gpu_scene) and INTEGRATION.md's "main.cpp compiles
// against this" are exercised by a test (tests/test_gpu_parity.py::test_cpp_boundary_main_flow), not only described.
// The reference declares the render entry as `extern "C" void gpu_render_scene(const GPUScene&, int, int)` (src/main.cpp:24-25); a
// reference parameter and a pointer are the same thing at the ABI, include/dsrt.h spells it as a pointer for C.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../host/scene_model.hpp"

using namespace dsrt;

static void point_camera_at(camera& cam, const vec3& cam_pos, const vec3& target_pos) {      // src/main.cpp:178-187
    cam.lookfrom = cam_pos;
    cam.lookat = target_pos;
    cam.vup = vec3(0, 1, 0);
    cam.focus_dist = (cam.lookfrom - cam.lookat).length();
    cam.initialize();
}

int main(int argc, char** argv) {
    if (argc < 9) { std::fprintf(stderr, "usage: main_flow_driver mesh.obj poses.txt frame W H spp depth out.ppm [repeat]\n"); return 2; }
    const std::string obj = argv[1], pose_file = argv[2], out = argv[8];
    const int frame = std::atoi(argv[3]), W = std::atoi(argv[4]), H = std::atoi(argv[5]), spp = std::atoi(argv[6]), depth = std::atoi(argv[7]);
    const int repeat = argc > 9 ? std::atoi(argv[9]) : 1;

    auto fallbackM = std::make_shared<lambertian>(vec3(0.73f, 0.73f, 0.73f));
    auto mesh = std::make_shared<triangle_mesh>(obj, fallbackM, 1.0);
    if (!mesh->loaded) { std::fprintf(stderr, "cannot load %s\n", obj.c_str()); return 1; }

    camera cam;
    cam.image_width = W;
    cam.image_height = H;
    cam.samples_per_pixel = spp;
    cam.max_depth = depth;
    cam.vfov = 40;
    cam.aperture = 0.0f;

    int n = 0;
    if (dsrt_read_pose_file(pose_file.c_str(), nullptr, 0, &n) != DSRT_OK || frame < 0 || frame >= n) { std::fprintf(stderr, "bad pose file or frame: %s\n", dsrt_last_error()); return 1; }
    std::vector<DsrtPose> poses((size_t)n);
    dsrt_read_pose_file(pose_file.c_str(), poses.data(), n, &n);
    DsrtFrame fr;
    dsrt_pose_to_frame(&poses[(size_t)frame], &fr);
    const vec3 cam_in_model(fr.cam_in_model[0], fr.cam_in_model[1], fr.cam_in_model[2]);
    const vec3 sun_dir_model(fr.sun_dir_model[0], fr.sun_dir_model[1], fr.sun_dir_model[2]);

    for (int r = 0; r < repeat; ++r) {              // the reference runs this body once per pose; repeating it exercises the per-frame rebuild
        hittable_list frame_world;
        frame_world.add(mesh);
        point_camera_at(cam, cam_in_model, vec3(0, 0, 0));
        GPUScene gpu_scene = build_gpu_scene(frame_world, cam, sun_dir_model);
        if (!gpu_scene.triangles) { std::fprintf(stderr, "build_gpu_scene failed: %s\n", dsrt_last_error()); return 1; }
        gpu_render_scene(&gpu_scene, cam.image_width, cam.image_height);
        if (std::rename("image_gpu.ppm", out.c_str()) != 0) { std::fprintf(stderr, "no image_gpu.ppm was written\n"); return 1; }
        free_gpu_scene(gpu_scene);
        if (gpu_scene.triangles || gpu_scene.num_triangles) { std::fprintf(stderr, "free_gpu_scene left the header populated\n"); return 1; }
    }
    std::printf("Saved %s\n", out.c_str());
    return 0;
}
