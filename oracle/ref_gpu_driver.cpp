// ref_gpu_driver.cpp -- runs the REFERENCE'S OWN renderer on the GPU: its scene builder and its render kernel, end to end.
//
// TEST INFRASTRUCTURE ONLY (see dsrt_oracle.h).  This file is ours; everything it calls is the reference's:
//   src/main.cpp               point_camera_at :178-187, the scene classes it includes (`main` itself renamed away)
//   src/gpu_scene_builder.cpp  build_gpu_scene :464-601, free_gpu_scene :603-626        (separate object)
//   src/gpu_render.cu          gpu_render_scene :1037-1108 -> render_kernel :973-1031 and everything below it (separate object)
// The two CUDA translation units and inc/gpu_scene.h name the CUDA runtime (<cuda_runtime.h>, cudaMalloc, <<<>>>), which this image does not
// have; it does have AMD's own CUDA-to-HIP source translator, hipify-perl (/opt/rocm/bin).  oracle/Makefile (target _ref/ref_gpu) copies the
// reference's inc/ and src/ into a scratch directory under /tmp, runs hipify-perl over exactly those three files there, compiles the result
// with hipcc for gfx950 (-ffp-contract=off -fno-fast-math, like everything else here; inc/cuda_compat.h's three macros are given on the
// command line as the __host__ / __device__ spellings its nvcc branch defines), links this driver, writes ONE binary to oracle/_ref/ and
// deletes the scratch directory.  Nothing of the reference enters the repository, no header, library or tool is stood in for by hand, and no
// line of the reference's arithmetic or control flow is touched: the translator renames runtime API calls and includes.
//
// What it is for: the one thing nothing else in this pipeline could do -- EXECUTE ray_color, scene_hit and bvh_hit_closest as the reference
// wrote them.  tests/test_gpu_reference_kernel.py renders the test scenes with this binary and with libdsrt_hip in math_mode 1 (the same kernels
// compiled with the platform's sinf / cosf / powf, which is what this build gets, where the default mode uses include/dsrt_detmath.h) and compares
// the images byte for byte.
//
// usage: ref_gpu <world.txt> W H spp max_depth  from_x from_y from_z  at_x at_y at_z  vfov  sun_x sun_y sun_z  <out.ppm> [1 = render twice and time the second call]
#define main dsrt_unused_reference_main
#include "src/main.cpp"
#undef main

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ref_world_loader.hpp"

int main(int argc, char** argv) {
    if (argc < 17) {
        std::fprintf(stderr, "usage: ref_gpu <world.txt> W H spp max_depth fx fy fz ax ay az vfov sx sy sz <out.ppm>\n");
        return 2;
    }
    hittable_list world = load_world(argv[1]);
    camera cam;
    cam.image_width = std::atoi(argv[2]);
    cam.image_height = std::atoi(argv[3]);
    cam.samples_per_pixel = std::atoi(argv[4]);
    cam.max_depth = std::atoi(argv[5]);
    cam.vfov = (float)std::atof(argv[12]);
    cam.aperture = 0.0;
    const vec3 from((float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]));
    const vec3 at((float)std::atof(argv[9]), (float)std::atof(argv[10]), (float)std::atof(argv[11]));
    point_camera_at(cam, from, at);                                         // src/main.cpp:178-187
    const vec3 sun((float)std::atof(argv[13]), (float)std::atof(argv[14]), (float)std::atof(argv[15]));
    const auto t0 = std::chrono::steady_clock::now();
    GPUScene scene = build_gpu_scene(world, cam, sun);                     // src/main.cpp:405
    const auto t1 = std::chrono::steady_clock::now();
    std::remove("image_gpu.ppm");
    gpu_render_scene(scene, cam.image_width, cam.image_height);            // src/main.cpp:413 (first call: includes the runtime's one-time kernel load)
    const auto t2 = std::chrono::steady_clock::now();
    double again_ms = -1.0;
    if (argc > 17 && std::atoi(argv[17]) > 0) {                             // optional: time a second call (same scene, everything warm)
        gpu_render_scene(scene, cam.image_width, cam.image_height);
        again_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    }
    std::printf("{\"triangles\": %d, \"spheres\": %d, \"materials\": %d, \"bvh_nodes\": %d, \"textures\": %d, \"build_gpu_scene_ms\": %.3f, "
                "\"gpu_render_scene_ms\": %.3f, \"gpu_render_scene_second_call_ms\": %.3f}\n", scene.num_triangles, scene.num_spheres,
                scene.num_materials, scene.num_bvh_nodes, scene.num_textures, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count(), again_ms);
    free_gpu_scene(scene);                                                  // src/main.cpp:428
    if (std::rename("image_gpu.ppm", argv[16]) != 0) { std::fprintf(stderr, "ref_gpu: the reference wrote no image_gpu.ppm\n"); return 1; }
    return 0;
}
