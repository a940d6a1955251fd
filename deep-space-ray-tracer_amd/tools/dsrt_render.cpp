// dsrt_render -- frame driver over the C ABI, taking the reference executable's flags.
//
// Mirrors what src/main.cpp of the reference does per run: --input_txt <pose file>, --output_dir <dir>
// (:194-215), one frame per pose named frame_%04zu.ppm (:418-425), frames closer than 1 m skipped (:342-345),
// camera at cam_in_model looking at the model origin with vfov 40 (:254-260, :399).  Differences, all deliberate:
//   * the mesh path, image size, spp and depth are flags (--obj --width --height --spp --depth) instead of
//     constants (:238, :255-258);
//   * the scene is flattened, BVH-built and uploaded ONCE; only camera and sun change per frame (the reference
//     rebuilds and re-uploads everything per frame, :405);
//   * the output directory is created if missing but never emptied (the reference deletes its contents, :41-50);
//   * PPM only: there is no ImageMagick shell-out (:28-36) and no upscaling step (:438-447).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../../include/dsrt.h"

static int fail(const char* what) {
    std::fprintf(stderr, "dsrt_render: %s: %s\n", what, dsrt_last_error());
    return 1;
}

int main(int argc, char** argv) {
    std::string pose_file, out_dir = "output", obj;
    int width = 800, height = 450, spp = 1000, depth = 50, first = 0, count = -1, rng_mode = 0, math_mode = 0;
    bool sah = false, lbvh = false, png = false, strict_textures = false, certified = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "dsrt_render: %s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (a == "--input_txt") pose_file = next("--input_txt");
        else if (a == "--output_dir") out_dir = next("--output_dir");
        else if (a == "--obj") obj = next("--obj");
        else if (a == "--width") width = std::atoi(next("--width"));
        else if (a == "--height") height = std::atoi(next("--height"));
        else if (a == "--spp") spp = std::atoi(next("--spp"));
        else if (a == "--depth") depth = std::atoi(next("--depth"));
        else if (a == "--frame") first = std::atoi(next("--frame"));
        else if (a == "--frames") count = std::atoi(next("--frames"));
        else if (a == "--fast") { sah = true; rng_mode = 1; }      // non-parity fast mode: SAH tree + Philox stream per sample (include/dsrt.h)
        else if (a == "--bvh") { const std::string k = next("--bvh"); sah = k == "sah"; lbvh = k == "lbvh"; }      // lbvh: built on the GPU (milliseconds), non-parity like sah
        else if (a == "--rng-mode") rng_mode = std::atoi(next("--rng-mode"));
        else if (a == "--reference-math") math_mode = 1;          // sinf / cosf / powf from the device math library: the reference's own kernel's bytes on this GPU (include/dsrt.h)
        else if (a == "--certified-tree") certified = true;        // rays walk the certified second tree: the reference's bytes, a third fewer node visits (include/dsrt.h)
        else if (a == "--strict-textures") strict_textures = true;  // refuse a mesh whose texture maps this library cannot decode (include/dsrt.h)
        else if (a == "--png") png = true;                          // frames as PNG instead of PPM (the reference converts with ImageMagick)
        else if (a == "--upscale") std::fprintf(stderr, "dsrt_render: --upscale is not supported (post-process outside this library)\n");
        else { std::fprintf(stderr, "usage: dsrt_render --obj mesh.obj [--input_txt poses.txt] [--output_dir dir] [--width W --height H --spp N --depth D] [--frame i --frames n] [--bvh median|sah|lbvh] [--rng-mode 0|1] [--reference-math] [--certified-tree] [--fast] [--png] [--strict-textures]\n"); return 2; }
    }
    if (obj.empty()) { std::fprintf(stderr, "dsrt_render: --obj is required\n"); return 2; }
    mkdir(out_dir.c_str(), 0777);

    std::vector<DsrtPose> poses;
    if (!pose_file.empty()) {
        int n = 0;
        if (dsrt_read_pose_file(pose_file.c_str(), nullptr, 0, &n) == DSRT_OK) {
            poses.resize((size_t)n);
            dsrt_read_pose_file(pose_file.c_str(), poses.data(), n, &n);
        }
    }
    if (poses.empty()) {                                   // the reference's default pose, src/main.cpp:275-284
        std::printf("No valid pose file found; using single default pose.\n");
        DsrtPose p{};
        p.cam_pos_world[0] = 0.0; p.cam_pos_world[1] = 50.0; p.cam_pos_world[2] = 200.0;
        p.model_pos_world[0] = 0.0; p.model_pos_world[1] = -100.0; p.model_pos_world[2] = 0.0;
        poses.push_back(p);
    } else {
        std::printf("Loaded %zu poses.\n", poses.size());
    }

    DsrtHostScene* hs = dsrt_host_scene_create();
    if (dsrt_host_scene_add_obj(hs, obj.c_str(), 1.0) != DSRT_OK) return fail("loading the mesh");
    {
        char names[4096];
        const int bad = dsrt_host_scene_texture_failures(hs, names, sizeof names);
        if (bad) {
            std::fprintf(stderr, "dsrt_render: %d texture map(s) could not be decoded (GIF, PSD, PIC, HDR and CMYK JPEG are not read here; the reference's stb_image reads them);\n"
                                 "they render as the reference's 1x1 white fallback, i.e. NOT like the reference would:\n%s", bad, names);
            if (strict_textures) return 3;
        }
    }
    {
        float build_ms = 0.0f, total_ms = 0.0f;
        const int rc = lbvh ? dsrt_host_scene_build_bvh_gpu(hs, 0, &build_ms, &total_ms) : (sah ? dsrt_host_scene_build_bvh_sah(hs) : dsrt_host_scene_build_bvh(hs));
        if (rc != DSRT_OK) return fail("building the BVH");
        if (lbvh) std::printf("BVH built on the GPU: %.2f ms of kernels, %.2f ms with upload and copy-back\n", build_ms, total_ms);
    }
    GPUScene scene;
    if (dsrt_host_scene_view(hs, &scene) != DSRT_OK) return fail("viewing the scene");
    std::printf("mesh: %d triangles, %d BVH nodes, %d materials\n", scene.num_triangles, scene.num_bvh_nodes, scene.num_materials);

    DsrtContext* ctx = nullptr;
    if (dsrt_ctx_create(0, &ctx) != DSRT_OK) return fail("creating the device context");
    if (certified && dsrt_ctx_set_certified_tree(ctx, 1) != DSRT_OK) return fail("asking for the certified second tree");
    // The reference's loop renders frame after frame (src/main.cpp:310-431).  Here the poses of a run are rendered as batch launches
    // (dsrt_render_batch_to_host: up to 32 frames as ONE pool of work, include/dsrt.h) and written out in pose order afterwards: each
    // frame's image is byte for byte what a launch of its own would give, the run is simply not held up by every frame's tail.
    const size_t last = count < 0 ? poses.size() : std::min(poses.size(), (size_t)(first + count));
    std::vector<size_t> ids;
    std::vector<GPUCamera> cams;
    std::vector<float> suns;
    for (size_t i = (size_t)first; i < last; ++i) {
        DsrtFrame fr;
        dsrt_pose_to_frame(&poses[i], &fr);
        std::printf("\n=== Frame %zu ===\n  sep(cam, model) = %g m\n", i, fr.sep_m);
        if (fr.skipped) { std::printf("  [!] Camera is inside/too close to ISS mesh. Skipping frame.\n"); continue; }
        GPUCamera cam;
        const float origin[3] = {0.0f, 0.0f, 0.0f};
        if (dsrt_camera_look_at(&cam, fr.cam_in_model, origin, 40.0f, width, height, spp, depth) != DSRT_OK) return fail("camera");
        if (ids.empty()) {
            dsrt_scene_set_frame(&scene, &cam, fr.sun_dir_model);
            if (dsrt_scene_upload(ctx, &scene) != DSRT_OK) return fail("uploading the scene");
        }
        ids.push_back(i); cams.push_back(cam);
        suns.insert(suns.end(), fr.sun_dir_model, fr.sun_dir_model + 3);
    }
    DsrtRenderDesc d;
    std::memset(&d, 0, sizeof d);
    d.width = width; d.height = height; d.spp = spp; d.max_depth = depth; d.gamma = 2.0f; d.seed = 1337; d.rng_mode = rng_mode; d.math_mode = math_mode;
    const size_t image_bytes = (size_t)width * height * 3;
    size_t per_launch = 32;
    while (per_launch > 1 && (unsigned long long)per_launch * width * height * (rng_mode == 1 ? 16ull : 1ull) >= (1ull << 32)) per_launch /= 2;
    std::vector<uint8_t> fb(image_bytes * std::min(per_launch, std::max<size_t>(ids.size(), 1)));
    for (size_t k = 0; k < ids.size(); k += per_launch) {
        const size_t n = std::min(per_launch, ids.size() - k);
        DsrtStats st;
        if (dsrt_render_batch_to_host(ctx, &d, (int)n, cams.data() + k, suns.data() + 3 * k, fb.data(), &st) != DSRT_OK) return fail("rendering");
        std::printf("\nframes %zu..%zu: kernel %.3f ms (%.1f Msamples/s)\n", ids[k], ids[k + n - 1], st.kernel_ms, (double)n * width * height * spp / (st.kernel_ms * 1e3));
        for (size_t q = 0; q < n; ++q) {
            char name[64];
            std::snprintf(name, sizeof name, png ? "/frame_%04zu.png" : "/frame_%04zu.ppm", ids[k + q]);
            const std::string path = out_dir + name;
            const uint8_t* img = fb.data() + q * image_bytes;
            if ((png ? dsrt_write_png(path.c_str(), img, width, height) : dsrt_write_ppm(path.c_str(), img, width, height)) != DSRT_OK) return fail("writing the frame");
            std::printf("Saved %s\n", path.c_str());
        }
    }
    dsrt_ctx_destroy(ctx);
    dsrt_host_scene_destroy(hs);
    std::printf("Done.\n");
    return 0;
}
