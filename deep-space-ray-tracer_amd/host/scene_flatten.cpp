// scene_flatten.cpp -- world objects -> the flat arrays the renderer consumes.
//
// Follows the CPU half of the reference's build_gpu_scene:
//   collect_from_hittable   src/gpu_scene_builder.cpp:252-308  (mesh -> triangle -> sphere -> list, in that order)
//   upsert_material         :71-139   (one table slot per distinct material OBJECT; a null material gets a fresh
//                                      0.8-gray lambertian slot every time)
//   make_gpu_triangle/sphere:39-66
//   HostTextureRegistry     :199-246  (one slot per distinct path; failure -> 1x1 white; sRGB -> linear powf(c/255, 2.2))
//   textured triangle       :274-278  (its material's albedo is overwritten with white)
//   defaults                :560-598  (dsrt_scene_set_frame)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "host_internal.hpp"

namespace dsrt {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

namespace {

DsrtF3 f3(const vec3& v) { return DsrtF3{v.x(), v.y(), v.z()}; }

int material_slot(DsrtHostScene& hs, const std::shared_ptr<material>& m) {
    GPUMaterial gm;
    std::memset(&gm, 0, sizeof gm);
    gm.albedo_tex = -1;
    if (!m) {
        gm.type = MAT_LAMBERTIAN;
        gm.albedo = DsrtF3{0.8f, 0.8f, 0.8f};
        gm.ref_idx = 1.5f;
        hs.mats.push_back(gm);
        return (int)hs.mats.size() - 1;
    }
    auto known = hs.mat_index.find(m.get());
    if (known != hs.mat_index.end()) return known->second;
    gm.type = m->type;
    gm.albedo = f3(m->albedo);
    gm.emissive = f3(m->emissive);
    gm.fuzz = m->fuzz;
    gm.ref_idx = m->ref_idx;
    hs.mats.push_back(gm);
    const int slot = (int)hs.mats.size() - 1;
    hs.mat_index.emplace(m.get(), slot);
    hs.keep_alive.push_back(m);
    return slot;
}

int texture_slot(DsrtHostScene& hs, const std::string& path, int flip = -1) {      // flip < 0: as the loader's latch says
    if (path.empty()) return -1;
    auto known = hs.tex_index.find(path);
    if (known != hs.tex_index.end()) return known->second;
    GPUTextureHeader h;
    h.offset = (int)hs.tex_pool.size();
    RgbImage img;
    if (!load_rgb8(path, flip < 0 ? texture_flip_latch() : flip != 0, img)) {
        std::fprintf(stderr, "WARN: failed to load texture '%s'\n", path.c_str());
        hs.tex_failed.push_back(path);
        h.width = h.height = 1;
        hs.tex_pool.insert(hs.tex_pool.end(), {1.0f, 1.0f, 1.0f});
    } else {
        h.width = img.width;
        h.height = img.height;
        float lut[256];
        for (int c = 0; c < 256; ++c) lut[c] = powf((float)c / 255.0f, 2.2f);
        hs.tex_pool.reserve(hs.tex_pool.size() + img.rgb.size());
        for (uint8_t b : img.rgb) hs.tex_pool.push_back(lut[b]);
    }
    hs.tex_headers.push_back(h);
    const int slot = (int)hs.tex_headers.size() - 1;
    hs.tex_index[path] = slot;
    return slot;
}

GPUTriangle pack_triangle(const triangle& t, int mat, int tex) {
    GPUTriangle g;
    std::memset(&g, 0, sizeof g);
    g.v0 = f3(t.v0); g.v1 = f3(t.v1); g.v2 = f3(t.v2);
    g.n0 = f3(t.n0); g.n1 = f3(t.n1); g.n2 = f3(t.n2);
    g.uv0 = DsrtF3{t.uv0.x(), t.uv0.y(), 0.0f};
    g.uv1 = DsrtF3{t.uv1.x(), t.uv1.y(), 0.0f};
    g.uv2 = DsrtF3{t.uv2.x(), t.uv2.y(), 0.0f};
    g.material_id = mat;
    g.albedo_tex = tex;
    return g;
}

void collect(DsrtHostScene& hs, const std::shared_ptr<hittable>& obj) {
    if (!obj) return;
    switch (obj->what()) {
    case hittable::kind::mesh: {
        const auto& mesh = static_cast<const triangle_mesh&>(*obj);
        for (size_t i = 0; i < mesh.triangles.size(); ++i) {
            const triangle& t = mesh.triangles[i];
            const int mat = material_slot(hs, t.mat);
            int tex = -1;
            if (i < mesh.tri_map_Kd.size() && !mesh.tri_map_Kd[i].empty()) tex = texture_slot(hs, mesh.tri_map_Kd[i]);
            if (tex >= 0) hs.mats[mat].albedo = DsrtF3{1.0f, 1.0f, 1.0f};
            hs.tris.push_back(pack_triangle(t, mat, tex));
        }
        break;
    }
    case hittable::kind::triangle: {
        const auto& t = static_cast<const triangle&>(*obj);
        hs.tris.push_back(pack_triangle(t, material_slot(hs, t.mat), -1));
        break;
    }
    case hittable::kind::sphere: {
        const auto& s = static_cast<const sphere&>(*obj);
        GPUSphere g;
        g.center = f3(s.center);
        g.radius = (float)s.radius;
        g.material_id = material_slot(hs, s.mat);
        g._pad = 0;
        hs.spheres.push_back(g);
        break;
    }
    case hittable::kind::list:
        for (const auto& child : static_cast<const hittable_list&>(*obj).objects) collect(hs, child);
        break;
    }
}

}  // namespace

int flatten_world(const hittable_list& world, DsrtHostScene* into) {
    if (!into) { set_error("flatten_world: null scene"); return DSRT_ERR_INVALID; }
    for (const auto& obj : world.objects) collect(*into, obj);
    into->bvh_valid = false;
    return DSRT_OK;
}

}  // namespace dsrt

using namespace dsrt;

extern "C" {

const char* dsrt_last_error(void) { return dsrt::g_last_error.c_str(); }
int dsrt_abi_version(void) { return DSRT_ABI_VERSION; }     // include/dsrt.h: the one place the number is written
size_t dsrt_sizeof(int which) {                              // what a binding's hand-written mirror of a struct must measure
    switch (which) {
        case DSRT_SIZEOF_RENDER_DESC: return sizeof(DsrtRenderDesc);
        case DSRT_SIZEOF_STATS: return sizeof(DsrtStats);
        case DSRT_SIZEOF_GPU_SCENE: return sizeof(GPUScene);
        case DSRT_SIZEOF_GPU_CAMERA: return sizeof(GPUCamera);
        case DSRT_SIZEOF_POSE: return sizeof(DsrtPose);
        case DSRT_SIZEOF_FRAME: return sizeof(DsrtFrame);
        default: return 0;
    }
}

DsrtHostScene* dsrt_host_scene_create(void) { return new DsrtHostScene(); }
void dsrt_host_scene_destroy(DsrtHostScene* hs) { delete hs; }

int dsrt_host_scene_add_obj(DsrtHostScene* hs, const char* obj_path, double scale) {
    return dsrt::guarded("dsrt_host_scene_add_obj", [&]() -> int {
    if (!hs || !obj_path) { set_error("dsrt_host_scene_add_obj: null argument"); return DSRT_ERR_INVALID; }
    auto fallback = std::make_shared<lambertian>(vec3(0.73f, 0.73f, 0.73f));      // src/main.cpp:240
    auto mesh = std::make_shared<triangle_mesh>(std::string(obj_path), fallback, scale);
    if (!mesh->loaded) { set_error(std::string("cannot open OBJ file ") + obj_path); return DSRT_ERR_IO; }
    hittable_list world(mesh);
    return flatten_world(world, hs);
    });
}

int dsrt_host_scene_add_world_file(DsrtHostScene* hs, const char* world_path) {
    return dsrt::guarded("dsrt_host_scene_add_world_file", [&]() -> int {
    if (!hs || !world_path) { set_error("dsrt_host_scene_add_world_file: null argument"); return DSRT_ERR_INVALID; }
    std::ifstream in(world_path);
    if (!in) { set_error(std::string("cannot open world file ") + world_path); return DSRT_ERR_IO; }
    hittable_list world;
    std::map<std::string, std::shared_ptr<material>> mats;
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        if (line.empty() || line[0] == '#') continue;
        std::istringstream iss(line);
        std::string tag;
        if (!(iss >> tag)) continue;
        auto bad = [&](const char* why) {
            set_error(std::string(world_path) + ":" + std::to_string(lineno) + ": " + why);
            return DSRT_ERR_INVALID;
        };
        if (tag == "mat") {
            std::string name, kind;
            double a = 0, b = 0, c = 0, d = 0;
            iss >> name >> kind;
            if (kind == "lambertian" && (iss >> a >> b >> c)) mats[name] = std::make_shared<lambertian>(color((float)a, (float)b, (float)c));
            else if (kind == "metal" && (iss >> a >> b >> c >> d)) mats[name] = std::make_shared<metal>(color((float)a, (float)b, (float)c), d);
            else if (kind == "dielectric" && (iss >> a)) mats[name] = std::make_shared<dielectric>(a);
            else if (kind == "light" && (iss >> a >> b >> c)) mats[name] = std::make_shared<diffuse_light>(color((float)a, (float)b, (float)c));
            else return bad("malformed mat line");
        } else if (tag == "sphere") {
            double x, y, z, r;
            std::string m;
            if (!(iss >> x >> y >> z >> r >> m) || !mats.count(m)) return bad("malformed sphere line or unknown material");
            world.add(std::make_shared<sphere>(point3((float)x, (float)y, (float)z), r, mats[m]));
        } else if (tag == "tri") {
            double v[9];
            std::string m;
            for (double& q : v) if (!(iss >> q)) return bad("malformed tri line");
            if (!(iss >> m) || !mats.count(m)) return bad("unknown material on tri line");
            world.add(std::make_shared<triangle>(vec3((float)v[0], (float)v[1], (float)v[2]), vec3((float)v[3], (float)v[4], (float)v[5]),
                                                 vec3((float)v[6], (float)v[7], (float)v[8]), mats[m]));
        } else if (tag == "obj") {
            std::string p;
            double scale = 1.0;
            if (!(iss >> p)) return bad("obj line without a path");
            iss >> scale;
            auto fallback = std::make_shared<lambertian>(vec3(0.73f, 0.73f, 0.73f));
            auto mesh = std::make_shared<triangle_mesh>(p, fallback, scale);
            if (!mesh->loaded) { set_error("cannot open OBJ file " + p); return DSRT_ERR_IO; }
            world.add(mesh);
        } else {
            return bad("unknown tag");
        }
    }
    return flatten_world(world, hs);
    });
}

int dsrt_host_scene_add_arrays(DsrtHostScene* hs, const GPUTriangle* tris, int num_tris, const GPUSphere* spheres, int num_spheres,
                               const GPUMaterial* mats, int num_mats) {
    return dsrt::guarded("dsrt_host_scene_add_arrays", [&]() -> int {
    if (!hs || num_tris < 0 || num_spheres < 0 || num_mats < 0 || (num_tris && !tris) || (num_spheres && !spheres) || (num_mats && !mats)) {
        set_error("dsrt_host_scene_add_arrays: bad argument");
        return DSRT_ERR_INVALID;
    }
    const int base = (int)hs->mats.size();
    for (int i = 0; i < num_tris; ++i) if (tris[i].material_id < 0 || tris[i].material_id >= num_mats) { set_error("triangle material id out of range"); return DSRT_ERR_INVALID; }
    for (int i = 0; i < num_spheres; ++i) if (spheres[i].material_id < 0 || spheres[i].material_id >= num_mats) { set_error("sphere material id out of range"); return DSRT_ERR_INVALID; }
    hs->mats.insert(hs->mats.end(), mats, mats + num_mats);
    for (int i = 0; i < num_tris; ++i) { GPUTriangle t = tris[i]; t.material_id += base; hs->tris.push_back(t); }
    for (int i = 0; i < num_spheres; ++i) { GPUSphere s = spheres[i]; s.material_id += base; hs->spheres.push_back(s); }
    hs->bvh_valid = false;
    return DSRT_OK;
    });
}

int dsrt_host_scene_add_texture_file(DsrtHostScene* hs, const char* path, int flip_vertically) {
    return dsrt::guarded("dsrt_host_scene_add_texture_file", [&]() -> int {
    if (!hs || !path || !path[0]) { set_error("dsrt_host_scene_add_texture_file: bad argument"); return DSRT_ERR_INVALID; }
    return texture_slot(*hs, path, flip_vertically != 0 ? 1 : 0);
    });
}

int dsrt_host_scene_view(const DsrtHostScene* hs, GPUScene* out) {
    if (!hs || !out) { set_error("dsrt_host_scene_view: null argument"); return DSRT_ERR_INVALID; }
    if (!hs->tris.empty() && !hs->bvh_valid) { set_error("dsrt_host_scene_view: call dsrt_host_scene_build_bvh first"); return DSRT_ERR_INVALID; }
    std::memset(out, 0, sizeof *out);
    out->spheres = hs->spheres.empty() ? nullptr : hs->spheres.data();
    out->num_spheres = (int)hs->spheres.size();
    out->triangles = hs->tris.empty() ? nullptr : hs->tris.data();
    out->num_triangles = (int)hs->tris.size();
    out->tri_indices = hs->tri_indices.empty() ? nullptr : hs->tri_indices.data();
    out->bvh_tri_indices = const_cast<int*>(out->tri_indices);
    out->bvh_nodes = hs->nodes.empty() ? nullptr : const_cast<GPUBVHNode*>(hs->nodes.data());
    out->num_bvh_nodes = (int)hs->nodes.size();
    out->materials = hs->mats.empty() ? nullptr : hs->mats.data();
    out->num_materials = (int)hs->mats.size();
    out->textures = hs->tex_headers.empty() ? nullptr : hs->tex_headers.data();
    out->num_textures = (int)hs->tex_headers.size();
    out->texture_pool = hs->tex_pool.empty() ? nullptr : hs->tex_pool.data();
    out->texture_pool_floats = (int)hs->tex_pool.size();
    return DSRT_OK;
}

int dsrt_host_scene_texture_failures(const DsrtHostScene* hs, char* names, size_t cap) {
    if (!hs) return 0;
    if (names && cap) {
        size_t at = 0;
        names[0] = 0;
        for (const std::string& p : hs->tex_failed) {
            if (at + p.size() + 2 > cap) break;
            std::memcpy(names + at, p.data(), p.size());
            at += p.size();
            names[at++] = '\n';
            names[at] = 0;
        }
    }
    return (int)hs->tex_failed.size();
}

int dsrt_host_scene_bvh_stack_need(const DsrtHostScene* hs) {
    if (!hs || !hs->bvh_valid) return 0;
    return hs->bvh_height > 0 ? hs->bvh_height - 1 : 0;
}

void dsrt_scene_set_frame(GPUScene* scene, const GPUCamera* cam, const float sun_dir_model[3]) {
    if (!scene) return;
    if (cam) scene->camera = *cam;
    scene->sky_type = SKY_SOLID;
    scene->env_tex_id = -1;
    scene->sky_solid = scene->sky_top = scene->sky_bottom = DsrtF3{0.0f, 0.0f, 0.0f};
    GPURenderParams p;
    std::memset(&p, 0, sizeof p);
    p.img_width = scene->camera.image_width;
    p.img_height = scene->camera.image_height;
    p.samples_per_pixel = scene->camera.samples_per_pixel;
    p.max_depth = scene->camera.max_depth;
    p.use_bvh = 1;
    p.gamma = 2.0f;
    p.exposure = 50.0f;
    scene->params = p;
    scene->seed = 1337ULL;
    scene->sun_enabled = 1;
    if (sun_dir_model) scene->sun_dir = DsrtF3{sun_dir_model[0], sun_dir_model[1], sun_dir_model[2]};
    scene->sun_radiance = DsrtF3{100000.0f, 95000.0f, 90000.0f};
}

}  // extern "C"
