// ref_world_loader.hpp -- the "world description" reader shared by the two drivers of the reference's own code (ref_host_driver.cpp,
// ref_gpu_driver.cpp).  TEST INFRASTRUCTURE ONLY.  The format is ours; every object it creates is an instance of the REFERENCE'S classes
// (included by the driver before this file: src/main.cpp pulls in hittable_list, sphere, triangle, triangle_mesh, material).
#pragma once
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

// ---- "world description" text format (ours): one object per line, in insertion order -------------
//   mat <name> lambertian r g b | metal r g b fuzz | dielectric ior | light r g b
//   sphere cx cy cz radius <mat>
//   tri x0 y0 z0 x1 y1 z1 x2 y2 z2 <mat>
//   obj <path> [scale]            (fallback material lambertian(0.73), src/main.cpp:240)
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
            auto fallbackM = std::make_shared<lambertian>(vec3(0.73, 0.73, 0.73));
            world.add(std::make_shared<triangle_mesh>(p, fallbackM, scale));
        }
    }
    return world;
}

