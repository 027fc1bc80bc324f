// scene_loader.cpp -- OBJ/MTL ingest producing exactly the three arrays the reference uploads.
//
// Mirrors loadFromFile (main.cpp:28-58).  tinyobjloader is not vendored in the reference
// checkout, so its behaviour on this path is restated: `v`, `f` with 1-based or negative
// indices and v/vt/vn forms, fan triangulation of polygons (PTH_QUAD_SHORTER_DIAGONAL: the quad rule of
// newer tinyobjloader releases), `mtllib`, `usemtl`, and from the MTL `newmtl`, `Kd`, `Ke`.  Shapes/groups do not matter: the reference concatenates all
// shapes in file order (main.cpp:38-57) and material ids are per face.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/pt_host.h"

namespace {

// A material DECLARED by newmtl starts all-zero, as tinyobjloader's InitMaterial leaves it, and a Kd / Ke line with
// fewer than three numbers leaves the missing ones 0 (its parseReal3 defaults).  Faces with NO material
// (material id -1: no usemtl yet, unknown name, missing MTL file) make the reference index materials[-1]
// (main.cpp:49, undefined); here they get kNoMaterial.
struct Material {
    float kd[3] = { 0.f, 0.f, 0.f };
    float ke[3] = { 0.f, 0.f, 0.f };
};
constexpr Material kNoMaterial = { { 0.6f, 0.6f, 0.6f }, { 0.f, 0.f, 0.f } };

bool read_file(const std::string &path, std::string &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

inline void skip_ws(const char *&p, const char *e)
{
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
}

inline std::string_view token(const char *&p, const char *e)
{
    skip_ws(p, e);
    const char *b = p;
    while (p < e && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n' && *p != '#') p++;
    return { b, (size_t)(p - b) };
}

inline bool parse_float(const char *&p, const char *e, float &v)
{
    skip_ws(p, e);
    if (p >= e || *p == '\n' || *p == '#') return false;
    char *end = nullptr;
    v = std::strtof(p, &end);  // text is NUL-terminated (std::string), so this cannot overrun
    if (end == p) return false;
    p = end;
    return true;
}

inline void next_line(const char *&p, const char *e)
{
    while (p < e && *p != '\n') p++;
    if (p < e) p++;
}

bool load_mtl(const std::string &path, std::vector<Material> &mats, std::unordered_map<std::string, int> &names)
{
    std::string text;
    if (!read_file(path, text)) return false;
    const char *p = text.c_str(), *e = p + text.size();
    Material *cur = nullptr;
    while (p < e) {
        std::string_view k = token(p, e);
        if (k == "newmtl") {
            std::string_view n = token(p, e);
            names[std::string(n)] = (int)mats.size();
            mats.emplace_back();
            cur = &mats.back();
        } else if ((k == "Kd" || k == "Ke") && cur) {
            float *dst = k == "Kd" ? cur->kd : cur->ke;
            float v[3] = { 0.f, 0.f, 0.f };
            int got = 0;
            while (got < 3 && parse_float(p, e, v[got])) got++;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        }
        next_line(p, e);
    }
    return true;
}

void set_err(char *err, size_t n, const std::string &msg)
{
    if (err && n) std::snprintf(err, n, "%s", msg.c_str());
}

}  // namespace

extern "C" int pth_load_obj(const char *obj_path, const char *mtl_dir, pth_scene *out, char *err, size_t err_len)
{
    return pth_load_obj_ex(obj_path, mtl_dir, 0u, out, err, err_len);
}

extern "C" int pth_load_obj_ex(const char *obj_path, const char *mtl_dir, uint32_t flags, pth_scene *out, char *err, size_t err_len)
{
    if (!obj_path || !out) { set_err(err, err_len, "null argument"); return 1; }
    std::memset(out, 0, sizeof(*out));
    std::string text;
    if (!read_file(obj_path, text)) { set_err(err, err_len, std::string("cannot open ") + obj_path); return 2; }
    std::string base = mtl_dir ? std::string(mtl_dir) : std::string(obj_path);
    if (!mtl_dir) {
        const size_t slash = base.find_last_of('/');
        base = slash == std::string::npos ? std::string(".") : base.substr(0, slash);
    }

    std::vector<float> pos;           // attrib.vertices
    std::vector<uint32_t> tri_idx;    // triangulated vertex indices
    std::vector<int> tri_mat;         // shape.mesh.material_ids
    std::vector<Material> mats;
    std::unordered_map<std::string, int> mat_names;
    int cur_mat = -1;
    std::vector<long> poly;

    const char *p = text.c_str(), *e = p + text.size();
    size_t line_no = 0;
    while (p < e) {
        line_no++;
        std::string_view k = token(p, e);
        if (k == "v") {
            float v[3] = { 0.f, 0.f, 0.f };
            for (int i = 0; i < 3; i++)
                if (!parse_float(p, e, v[i])) {
                    set_err(err, err_len, "bad vertex at line " + std::to_string(line_no));
                    return 3;
                }
            pos.insert(pos.end(), v, v + 3);
        } else if (k == "f") {
            poly.clear();
            for (;;) {
                std::string_view t = token(p, e);
                if (t.empty()) break;
                char *end = nullptr;
                errno = 0;
                const long i = std::strtol(t.data(), &end, 10);  // "a", "a/b", "a//c", "a/b/c": vertex index first
                if (end == t.data()) { set_err(err, err_len, "bad face at line " + std::to_string(line_no)); return 3; }
                const long nv = (long)(pos.size() / 3);
                const long vi = i > 0 ? i - 1 : nv + i;  // negative = relative to the vertices so far
                if (i == 0 || vi < 0 || vi >= nv) {
                    set_err(err, err_len, "face index out of range at line " + std::to_string(line_no));
                    return 3;
                }
                poly.push_back(vi);
            }
            if (poly.size() < 3) { set_err(err, err_len, "face with < 3 vertices at line " + std::to_string(line_no)); return 3; }
            if (poly.size() == 4 && (flags & PTH_QUAD_SHORTER_DIAGONAL)) {
                // tinyobjloader >= v2.0.0rc9 (as far as it is known here; the reference's submodule is an unpinned,
                // empty directory): a quad is cut along its SHORTER diagonal, (0,1,2)(0,2,3) when |v0v2|^2 < |v1v3|^2,
                // else (0,1,3)(1,2,3)
                auto d2 = [&](long a, long b) {
                    float s2 = 0.f;
                    for (int c = 0; c < 3; c++) { const float d = pos[3 * (size_t)a + c] - pos[3 * (size_t)b + c]; s2 += d * d; }
                    return s2;
                };
                const bool fan = d2(poly[0], poly[2]) < d2(poly[1], poly[3]);
                const int order[2][6] = { { 0, 1, 3, 1, 2, 3 }, { 0, 1, 2, 0, 2, 3 } };
                for (int c = 0; c < 6; c++) tri_idx.push_back((uint32_t)poly[(size_t)order[fan ? 1 : 0][c]]);
                tri_mat.push_back(cur_mat);
                tri_mat.push_back(cur_mat);
                next_line(p, e);
                continue;
            }
            for (size_t c = 1; c + 1 < poly.size(); c++) {  // fan: (0,1,2) (0,2,3) ...
                tri_idx.push_back((uint32_t)poly[0]);
                tri_idx.push_back((uint32_t)poly[c]);
                tri_idx.push_back((uint32_t)poly[c + 1]);
                tri_mat.push_back(cur_mat);
            }
        } else if (k == "usemtl") {
            std::string_view n = token(p, e);
            auto it = mat_names.find(std::string(n));
            cur_mat = it == mat_names.end() ? -1 : it->second;
        } else if (k == "mtllib") {
            std::string_view n = token(p, e);
            if (!load_mtl(base + "/" + std::string(n), mats, mat_names)) {
                // tinyobjloader only warns; faces then have material id -1
            }
        }
        next_line(p, e);
    }
    if (tri_mat.empty()) { set_err(err, err_len, "no faces in OBJ"); return 4; }

    const size_t nt = tri_mat.size();
    out->n_tris = (uint32_t)nt;
    out->n_verts = (uint32_t)(3 * nt);
    out->vertices = (float *)std::malloc(sizeof(float) * 9 * nt);
    out->indices = (uint32_t *)std::malloc(sizeof(uint32_t) * 3 * nt);
    out->faces = (float *)std::malloc(sizeof(float) * 6 * nt);
    if (!out->vertices || !out->indices || !out->faces) {
        pth_free_scene(out);
        set_err(err, err_len, "out of memory");
        return 5;
    }
    for (size_t i = 0; i < 3 * nt; i++) {  // main.cpp:39-46
        const uint32_t vi = tri_idx[i];
        out->vertices[3 * i + 0] = pos[3 * (size_t)vi + 0];
        out->vertices[3 * i + 1] = -pos[3 * (size_t)vi + 1];  // Y flipped
        out->vertices[3 * i + 2] = pos[3 * (size_t)vi + 2];
        out->indices[i] = (uint32_t)i;
    }
    for (size_t t = 0; t < nt; t++) {  // main.cpp:47-56
        const Material &m = tri_mat[t] >= 0 ? mats[(size_t)tri_mat[t]] : kNoMaterial;
        for (int c = 0; c < 3; c++) {
            out->faces[6 * t + c] = m.kd[c];
            out->faces[6 * t + 3 + c] = m.ke[c];
        }
    }
    return 0;
}

extern "C" void pth_free_scene(pth_scene *s)
{
    if (!s) return;
    std::free(s->vertices);
    std::free(s->indices);
    std::free(s->faces);
    std::memset(s, 0, sizeof(*s));
}
