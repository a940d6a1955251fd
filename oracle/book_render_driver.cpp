// book_render_driver.cpp -- the "book-style CPU hittable_list render" BASELINE.json names as the reported CPU baseline.
//
// TEST / BASELINE INFRASTRUCTURE ONLY (see dsrt_oracle.h); built by oracle/Makefile into oracle/_ref/book_render where
// /root/reference exists.  Everything that touches geometry or materials is the REFERENCE'S OWN code, compiled from the
// headers where they lie: hittable_list::hit (inc/hittable_list.h:29-48), sphere::hit (inc/sphere.h:67-102),
// triangle_mesh::hit -- a linear scan over all triangles -- (inc/triangle_mesh.h:31-47), triangle::hit (inc/triangle.h:31-61),
// lambertian/metal/dielectric/diffuse_light::scatter/emitted (inc/material.h:70-226), cosine_pdf (inc/pdf.h:22-39),
// camera::initialize (inc/camera.h:91-116), and their rand()-based host RNG (inc/rtweekend.h:55-110).
//
// What the reference does NOT contain is a caller for any of it: no ray_color, no pixel loop.  The loop
// below is ours, in the style of the book the classes come from ("The Rest of Your Life": emitted + attenuation *
// scattering_pdf * L / pdf with the material's own cosine pdf; specular materials skip the pdf), black background plus the
// same directional sun term the GPU path adds at Lambertian hits, so that the two renders show the same scene.  It is a
// timing baseline, not a parity target: different RNG, double/float mix of the book classes, recursive evaluation.
//
// usage: book_render <world.txt> <W> <H> <spp> <max_depth> <threads> fx fy fz ax ay az vfov sx sy sz [out.ppm|- [y0 y1]]
//        (the book classes draw from rand(), which takes a process-wide lock in glibc: threads of ONE process do not scale, so
//        callers that want all cores start one process per row band, y0..y1, and add the samples up)
//        prints one JSON line {"samples":..,"seconds":..,"threads":..,"msamples_per_s":..}; the TSV schema of the reference's
//        scripts/performance.py (num_threads\tduration_ns) is what tests/bench assemble from several runs.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <vector>

#include "rtweekend.h"
#include "camera.h"
#include "hittable_list.h"
#include "triangle_mesh.h"
#include "material.h"
#include "sphere.h"
#include "stb_image_impl.cpp"

static hittable_list load_world(const std::string& path) {
    hittable_list world;
    std::map<std::string, std::shared_ptr<material>> mats;
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream iss(line);
        std::string tag; iss >> tag;
        if (tag == "mat") {
            std::string name, kind; iss >> name >> kind;
            double a, b, c, d;
            if (kind == "lambertian") { iss >> a >> b >> c; mats[name] = std::make_shared<lambertian>(color(a, b, c)); }
            else if (kind == "metal") { iss >> a >> b >> c >> d; mats[name] = std::make_shared<metal>(color(a, b, c), d); }
            else if (kind == "dielectric") { iss >> a; mats[name] = std::make_shared<dielectric>(a); }
            else if (kind == "light") { iss >> a >> b >> c; mats[name] = std::make_shared<diffuse_light>(color(a, b, c)); }
        } else if (tag == "sphere") {
            double x, y, z, r; std::string m; iss >> x >> y >> z >> r >> m;
            world.add(std::make_shared<sphere>(point3(x, y, z), r, mats.at(m)));
        } else if (tag == "tri") {
            double v[9]; for (double& q : v) iss >> q; std::string m; iss >> m;
            world.add(std::make_shared<triangle>(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]), mats.at(m)));
        } else if (tag == "obj") {
            std::string p; double scale = 1.0; iss >> p; iss >> scale;
            world.add(std::make_shared<triangle_mesh>(p, std::make_shared<lambertian>(vec3(0.73, 0.73, 0.73)), scale));
        }
    }
    return world;
}

struct Sun { vec3 to_light; vec3 radiance; };

static vec3 ray_color(const ray& r, int depth, const hittable_list& world, const Sun& sun) {
    if (depth <= 0) return vec3(0, 0, 0);
    hit_record rec;
    if (!world.hit(r, interval(0.001, 1e9), rec)) return vec3(0, 0, 0);          // black sky, as the GPU path
    vec3 emitted = rec.mat_ptr->emitted(r, rec, rec.u, rec.v, rec.p);
    scatter_record srec;
    if (!rec.mat_ptr->scatter(r, rec, srec)) return emitted;
    if (srec.skip_pdf) return srec.attenuation * ray_color(srec.specular_ray, depth - 1, world, sun);

    vec3 direct(0, 0, 0);                                                        // directional sun at diffuse hits
    float cos_l = dot(rec.normal, sun.to_light);
    if (cos_l > 0.0f) {
        hit_record shadow;
        if (!world.hit(ray(rec.p + 1e-3f * rec.normal, sun.to_light), interval(0.001, 1e9), shadow))
            direct = srec.attenuation * sun.radiance * (cos_l / (float)rt_pi());
    }
    ray scattered(rec.p, srec.pdf_ptr->generate());
    double pdf = srec.pdf_ptr->value(scattered.direction());
    if (pdf <= 0.0) return emitted + direct;
    double spdf = rec.mat_ptr->scattering_pdf(r, rec, scattered);
    return emitted + direct + (srec.attenuation * ray_color(scattered, depth - 1, world, sun)) * (float)(spdf / pdf);
}

int main(int argc, char** argv) {
    if (argc < 17) { std::fprintf(stderr, "usage: book_render world.txt W H spp depth threads fx fy fz ax ay az vfov sx sy sz [out.ppm]\n"); return 2; }
    hittable_list world = load_world(argv[1]);
    camera cam;
    cam.image_width = std::atoi(argv[2]);
    cam.image_height = std::atoi(argv[3]);
    cam.samples_per_pixel = std::atoi(argv[4]);
    cam.max_depth = std::atoi(argv[5]);
    const int threads = std::max(1, std::atoi(argv[6]));
    cam.lookfrom = point3(std::atof(argv[7]), std::atof(argv[8]), std::atof(argv[9]));
    cam.lookat = point3(std::atof(argv[10]), std::atof(argv[11]), std::atof(argv[12]));
    cam.vfov = (float)std::atof(argv[13]);
    cam.vup = vec3(0, 1, 0);
    cam.aperture = 0.0;
    cam.focus_dist = (cam.lookfrom - cam.lookat).length();
    cam.initialize();
    vec3 sun_dir(std::atof(argv[14]), std::atof(argv[15]), std::atof(argv[16]));   // ISS -> Sun as in the pose pipeline; light arrives from -sun_dir
    Sun sun{unit_vector(-sun_dir), vec3(100000.0f, 95000.0f, 90000.0f)};

    const int W = cam.image_width, H = cam.image_height, spp = cam.samples_per_pixel;
    std::vector<unsigned char> img((size_t)W * H * 3, 0);
    const int y0 = argc > 19 ? std::max(0, std::atoi(argv[18])) : 0, y1 = argc > 19 ? std::min(H, std::atoi(argv[19])) : H;
    std::atomic<int> next_row{y0};
    auto worker = [&]() {
        for (int j = next_row++; j < y1; j = next_row++) {
            for (int i = 0; i < W; ++i) {
                vec3 acc(0, 0, 0);
                for (int s = 0; s < spp; ++s) {
                    float u = ((float)i + (float)random_double_host()) / (float)(W - 1);
                    float v = ((float)j + (float)random_double_host()) / (float)(H - 1);
                    ray r(cam.origin, cam.lower_left_corner + u * cam.horizontal + v * cam.vertical - cam.origin);
                    vec3 c = ray_color(r, cam.max_depth, world, sun);
                    acc += vec3(fminf(1.0f, fmaxf(0.0f, c.x())), fminf(1.0f, fmaxf(0.0f, c.y())), fminf(1.0f, fmaxf(0.0f, c.z())));
                }
                rgb8 px = pack_color(acc, spp);                                    // inc/color.h:56-62
                size_t o = ((size_t)(H - 1 - j) * W + i) * 3;
                img[o] = px.r; img[o + 1] = px.g; img[o + 2] = px.b;
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (argc > 17 && std::string(argv[17]) != "-") {
        FILE* f = std::fopen(argv[17], "wb");
        if (f) { std::fprintf(f, "P6\n%d %d\n255\n", W, H); std::fwrite(img.data(), 1, img.size(), f); std::fclose(f); }
    }
    const double samples = (double)W * (y1 - y0) * spp;
    std::printf("{\"samples\": %.0f, \"seconds\": %.6f, \"threads\": %d, \"duration_ns\": %.0f, \"msamples_per_s\": %.6f}\n", samples, sec, threads, sec * 1e9,
                samples / sec / 1e6);
    return 0;
}
