// obj_mesh.cpp -- OBJ + MTL ingestion with the reference loader's observable behaviour.
//
// Behaviour followed (inc/triangle_mesh.h in the reference):
//   :171-255  line tags mtllib / usemtl / v / vt / f; `v` scaled in double then narrowed to float; `vt` stored
//             as (u, 1-v, 0); faces fan-triangulated from their first vertex; face tokens v, v/vt, v//vn, v/vt/vn
//             (vn is parsed and ignored: normals are always the flat face normal, inc/triangle.h:70-73);
//             a material object is created the first time a `usemtl` name is used by a face and reused by name.
//   :114-168  MTL keys newmtl / Kd / Ks / Ke / Ns / d / Ni / map_Kd / map_Ke, everything else ignored.
//   :75-112   material choice, in this order: emissive (Ke != 0 or map_Ke) -> diffuse_light; map_Kd -> textured
//             lambertian; d < 0.999 -> dielectric(Ni if 0.1 < Ni < 10 else 1.5); |Ks| > 0.05 -> metal(Ks,
//             fuzz = clamp(100/(Ns+100), 0, 1)); else lambertian(Kd).
// Deliberate differences, all on inputs where the reference has undefined behaviour: a face index that is
// negative or beyond the vertices read so far makes the reference index out of bounds (:224, :232-233); here
// that triangle (or, for the fan pivot, the face) is skipped.
//
// Texture orientation (SURVEY.md note T): constructing the reference's image_texture calls
// stbi_set_flip_vertically_on_load(true) (inc/texture.h:133), a process-global switch, so every image the
// builder decodes afterwards (src/gpu_scene_builder.cpp:215) arrives bottom row first.  texture_flip_latch()
// records that the switch has been thrown, and image_io.cpp honours it.
#include "scene_model.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace dsrt {

static bool g_flip_latch = false;
bool texture_flip_latch() { return g_flip_latch; }
void texture_flip_latch_set(bool v) { g_flip_latch = v; }

namespace {

struct MtlEntry {
    vec3 Kd{0.8f, 0.8f, 0.8f}, Ks{0.0f, 0.0f, 0.0f}, Ke{0.0f, 0.0f, 0.0f};
    double Ns = 0.0, d = 1.0, Ni = 1.5;
    std::string map_Kd, map_Ke;
};

std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out;
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && std::isspace((unsigned char)line[i])) ++i;
        size_t j = i;
        while (j < n && !std::isspace((unsigned char)line[j])) ++j;
        if (j > i) out.emplace_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

// Parse a whole token as a double the way `istream >> double` accepts it (leading numeric prefix).
bool to_double(const std::string& s, double& v) {
    char* end = nullptr;
    v = std::strtod(s.c_str(), &end);
    return end != s.c_str();
}
bool to_float(const std::string& s, float& v) {
    char* end = nullptr;
    v = std::strtof(s.c_str(), &end);
    return end != s.c_str();
}
bool three_doubles(const std::vector<std::string>& t, double& a, double& b, double& c) {
    return t.size() >= 4 && to_double(t[1], a) && to_double(t[2], b) && to_double(t[3], c);
}

std::unordered_map<std::string, MtlEntry> read_mtl(const std::string& path) {
    std::unordered_map<std::string, MtlEntry> table;
    std::ifstream in(path);
    if (!in) return table;
    std::string line, current;
    MtlEntry e;
    auto commit = [&]() { if (!current.empty()) table[current] = e; };
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        auto t = split_ws(line);
        if (t.empty()) continue;
        const std::string& key = t[0];
        double a, b, c;
        if (key == "newmtl") { commit(); e = MtlEntry{}; if (t.size() > 1) current = t[1]; }
        else if (key == "Kd") { if (three_doubles(t, a, b, c)) e.Kd = vec3((float)a, (float)b, (float)c); }
        else if (key == "Ks") { if (three_doubles(t, a, b, c)) e.Ks = vec3((float)a, (float)b, (float)c); }
        else if (key == "Ke") { if (three_doubles(t, a, b, c)) e.Ke = vec3((float)a, (float)b, (float)c); }
        else if (key == "Ns") { if (t.size() > 1 && to_double(t[1], a)) e.Ns = a; }
        else if (key == "d")  { if (t.size() > 1 && to_double(t[1], a)) e.d = a; }
        else if (key == "Ni") { if (t.size() > 1 && to_double(t[1], a)) e.Ni = a; }
        else if (key == "map_Kd") { if (t.size() > 1) e.map_Kd = t[1]; }
        else if (key == "map_Ke") { if (t.size() > 1) e.map_Ke = t[1]; }
    }
    commit();
    return table;
}

std::shared_ptr<material> material_from_mtl(const MtlEntry& m) {
    const bool emissive = (m.Ke.x() != 0.0f || m.Ke.y() != 0.0f || m.Ke.z() != 0.0f);
    if (emissive || !m.map_Ke.empty()) {
        if (!m.map_Ke.empty()) {
            g_flip_latch = true;                                   // image_texture constructed
            return std::make_shared<diffuse_light>(color(1.0f, 1.0f, 1.0f));   // emit_value() of a non-solid texture
        }
        return std::make_shared<diffuse_light>(m.Ke);
    }
    if (!m.map_Kd.empty()) {
        g_flip_latch = true;                                       // image_texture constructed
        return std::make_shared<lambertian>(lambertian::textured_tag{});
    }
    if (m.d < 0.999) {
        double ior = (m.Ni > 0.1 && m.Ni < 10.0) ? m.Ni : 1.5;
        return std::make_shared<dielectric>(ior);
    }
    const double ks_mag = (double)m.Ks.length();
    if (ks_mag > 0.05) {
        double fuzz = std::clamp(100.0 / (m.Ns + 100.0), 0.0, 1.0);
        return std::make_shared<metal>(m.Ks, fuzz);
    }
    return std::make_shared<lambertian>(m.Kd);
}

// "v", "v/vt", "v//vn", "v/vt/vn" -> (v, vt); 0 where absent or unparsable.
void face_indices(const std::string& tok, int& v, int& vt) {
    v = vt = 0;
    const char* p = tok.c_str();
    char* end = nullptr;
    long a = std::strtol(p, &end, 10);
    if (end == p) return;
    v = (int)a;
    if (*end != '/') return;
    p = end + 1;
    if (*p == '/') return;                 // v//vn
    long b = std::strtol(p, &end, 10);
    if (end == p) return;
    vt = (int)b;
}

}  // namespace

triangle_mesh::triangle_mesh(const std::string& obj_path, std::shared_ptr<material> fallback_mat, double scale)
    : fallback(std::move(fallback_mat)) {
    std::ifstream in(obj_path);
    if (!in) return;
    loaded = true;
    const std::string base_dir = obj_path.substr(0, obj_path.find_last_of("/\\") + 1);
    std::unordered_map<std::string, MtlEntry> mtl;
    std::unordered_map<std::string, std::shared_ptr<material>> made;
    std::string line, active;

    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        auto t = split_ws(line);
        if (t.empty()) continue;
        const std::string& tag = t[0];
        if (tag == "mtllib") {
            if (t.size() > 1) for (auto& kv : read_mtl(base_dir + t[1])) mtl[kv.first] = kv.second;
            else for (auto& kv : read_mtl(base_dir)) mtl[kv.first] = kv.second;
        } else if (tag == "usemtl") {
            if (t.size() > 1) active = t[1];      // a bare `usemtl` leaves the current name as it was (failed extraction)
        } else if (tag == "v") {
            double x, y, z;
            if (three_doubles(t, x, y, z)) verts.emplace_back((float)(scale * x), (float)(scale * y), (float)(scale * z));
        } else if (tag == "vt") {
            float a, b;
            if (t.size() >= 3 && to_float(t[1], a) && to_float(t[2], b)) uvs.emplace_back(a, 1.0f - b, 0.0f);
        } else if (tag == "f") {
            if (t.size() < 4) continue;
            std::shared_ptr<material> use = fallback;
            const MtlEntry* entry = nullptr;
            if (!active.empty()) {
                auto hit = mtl.find(active);
                if (hit != mtl.end()) entry = &hit->second;
                auto cached = made.find(active);
                if (cached != made.end()) use = cached->second;
                else if (entry) { use = material_from_mtl(*entry); made[active] = use; }
            }
            const int nverts = (int)verts.size(), nuv = (int)uvs.size();
            auto uv_of = [&](int it) { return (it > 0 && it <= nuv) ? uvs[it - 1] : vec3(0.0f, 0.0f, 0.0f); };
            int i0, t0;
            face_indices(t[1], i0, t0);
            if (i0 <= 0 || i0 > nverts) continue;
            const vec3 p0 = verts[i0 - 1], q0 = uv_of(t0);
            for (size_t k = 2; k + 1 < t.size(); ++k) {
                int i1, t1, i2, t2;
                face_indices(t[k], i1, t1);
                face_indices(t[k + 1], i2, t2);
                if (i1 <= 0 || i2 <= 0 || i1 > nverts || i2 > nverts) continue;
                triangles.emplace_back(p0, verts[i1 - 1], verts[i2 - 1], q0, uv_of(t1), uv_of(t2), use);
                tri_map_Kd.push_back((entry && !entry->map_Kd.empty()) ? base_dir + entry->map_Kd : std::string());
            }
        }
    }
}

}  // namespace dsrt
