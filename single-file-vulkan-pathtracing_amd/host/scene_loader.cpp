// scene_loader.cpp -- OBJ/MTL ingest producing exactly the three arrays the reference uploads.
//
// Mirrors loadFromFile (main.cpp:28-58).  tinyobjloader is not vendored in the reference
// checkout, so its behaviour on this path is restated: `v`, `f` with 1-based or negative
// indices and v/vt/vn forms, fan triangulation of polygons (PTH_QUAD_SHORTER_DIAGONAL: the quad rule of
// newer tinyobjloader releases), `mtllib`, `usemtl`, and from the MTL `newmtl`, `Kd`, `Ke`.  Shapes/groups do not matter: the reference concatenates all
// shapes in file order (main.cpp:38-57) and material ids are per face.
#include <sys/stat.h>
#include <exception>
#include <new>
#include <stdexcept>
#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
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
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode)) {  // (a directory opens for reading and "is" LONG_MAX bytes long: `mtllib` without a name)
        std::fclose(f);
        return false;
    }
    bool ok = std::fseek(f, 0, SEEK_END) == 0;
    const long n = ok ? std::ftell(f) : -1;
    ok = ok && n >= 0 && std::fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out.resize((size_t)n);
        ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    }
    std::fclose(f);
    return ok;
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
    // std::from_chars: correctly rounded like glibc's strtof, several times faster, no locale.  What it does not take goes to
    // strtof: a leading '+', hexadecimal floats, values out of float's range (strtof's HUGE_VALF / denormal results are kept)
    const char *q = p;
    const bool neg = *q == '-';
    const char *d = (neg || *q == '+') ? q + 1 : q;
    const bool plain = *q != '+' && !(d + 1 < e && d[0] == '0' && (d[1] == 'x' || d[1] == 'X'));
    if (plain) {
        float r = 0.f;
        const auto res = std::from_chars(q, e, r);
        if (res.ec == std::errc()) { v = r; p = res.ptr; return true; }
        if (res.ec == std::errc::invalid_argument) return false;
    }
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

// ---- the OBJ text in parallel ----------------------------------------------------------------------------------------
// The text is cut at line starts into one chunk per thread.  Pass A (per chunk): the `v` lines into a local array, the `f`
// lines as raw index lists (a relative index needs the number of vertices BEFORE the line, which a chunk only knows locally),
// `usemtl` / `mtllib` as events with their place among the chunk's faces.  In between, in file order (little work): vertex /
// line counts before each chunk, the MTL files, the material in force at each chunk's start.  Pass B (per chunk): indices
// resolved and range-checked, polygons cut into triangles, materials attached; pass C writes the three arrays.  The result --
// and, for a bad file, the error and its line -- is what one thread reading line by line produces (the loop this replaces),
// checked against tests/obj_ref.py with chunks of a few hundred bytes.  1 M triangles, 139 MB of text, on the GPU box's host
// (16 threads): 0.60 s -> 0.09 s (profiles/r04ac_load_obj.log).
namespace {

struct Event { uint32_t face; bool lib; std::string_view name; size_t off; };
struct ObjError { size_t line = 0; uint32_t token = 0; int code = 0; const char *what = nullptr; bool set = false; };
inline void note(ObjError &e, size_t line, uint32_t token, int code, const char *what)
{
    if (!e.set || line < e.line || (line == e.line && token < e.token)) e = { line, token, code, what, true };
}

struct Chunk {
    const char *b = nullptr, *e = nullptr;
    std::vector<float> pos;
    std::vector<long> idx;            // raw indices of all polygons, one after the other
    std::vector<uint32_t> start;      // polygon f: idx[start[f] .. start[f + 1])
    std::vector<uint32_t> nv_at;      // vertices of THIS chunk seen before polygon f's line
    std::vector<uint32_t> line_of;    // its line within the chunk (1-based)
    std::vector<Event> ev;
    size_t lines = 0;
    ObjError err;                     // first error of pass A (line within the chunk)
    size_t v_before = 0, line_before = 0, tri_before = 0, n_tri = 0;
    int mat_in = -1;
    std::vector<int> ev_mat;          // resolved material of every usemtl event
    std::vector<uint32_t> tri_idx;    // pass B
    std::vector<int> tri_mat;
};

void parse_chunk(Chunk &c)
{
    const char *p = c.b, *e = c.e;
    c.start.push_back(0u);
    while (p < e) {
        c.lines++;
        std::string_view k = token(p, e);
        if (k == "v") {
            float v[3] = { 0.f, 0.f, 0.f };
            for (int i = 0; i < 3; i++)
                if (!parse_float(p, e, v[i])) { note(c.err, c.lines, 0u, 3, "bad vertex"); return; }
            c.pos.insert(c.pos.end(), v, v + 3);
        } else if (k == "f") {
            uint32_t n = 0;
            for (;;) {
                std::string_view t = token(p, e);
                if (t.empty()) break;
                char *end = nullptr;
                const long i = std::strtol(t.data(), &end, 10);  // "a", "a/b", "a//c", "a/b/c": vertex index first
                // (the index ends the token or is followed by '/': "1x" is not index 1)
                if (end == t.data() || (end != t.data() + t.size() && *end != '/')) { note(c.err, c.lines, n, 3, "bad face"); break; }
                if (i == 0) { note(c.err, c.lines, n, 3, "face index out of range"); break; }
                c.idx.push_back(i);
                n++;
            }
            if (!c.err.set && n < 3) note(c.err, c.lines, n, 3, "face with < 3 vertices");
            // (a line that ends the parse is recorded as far as it got: an index before the offending token may be out of range,
            // which pass B finds and which a reader going token by token would have reported first)
            c.start.push_back((uint32_t)c.idx.size());
            c.nv_at.push_back((uint32_t)(c.pos.size() / 3));
            c.line_of.push_back((uint32_t)c.lines);
            if (c.err.set) return;
        } else if (k == "usemtl" || k == "mtllib") {
            std::string_view n = token(p, e);
            c.ev.push_back({ (uint32_t)c.nv_at.size(), k == "mtllib", n, (size_t)(n.data() - c.b) });
        }
        next_line(p, e);
    }
}

template <class F>
void for_chunks(std::vector<Chunk> &cs, F f)
{
    if (cs.size() == 1) { f(cs[0]); return; }
    // an exception of a worker (std::bad_alloc on a huge file) is carried to the caller: left inside the thread it would end the process
    std::vector<std::exception_ptr> failed(cs.size());
    auto guarded = [&cs, &f, &failed](size_t k) {
        try { f(cs[k]); } catch (...) { failed[k] = std::current_exception(); }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < cs.size(); k++) th.emplace_back(guarded, k);
    guarded(0);
    for (auto &t : th) t.join();
    for (const auto &e : failed)
        if (e) std::rethrow_exception(e);
}

}  // namespace

static int load_obj_impl(const char *obj_path, const char *mtl_dir, uint32_t flags, pth_scene *out, char *err, size_t err_len);

// (nothing may leave the C-ABI as an exception -- the reference's loader throws, main.cpp:35; here it is a status and a message)
extern "C" int pth_load_obj_ex(const char *obj_path, const char *mtl_dir, uint32_t flags, pth_scene *out, char *err, size_t err_len)
{
    try {
        return load_obj_impl(obj_path, mtl_dir, flags, out, err, err_len);
    } catch (const std::bad_alloc &) {
        set_err(err, err_len, "out of memory");
    } catch (const std::exception &e) {
        set_err(err, err_len, std::string("internal error: ") + e.what());
    } catch (...) {
        set_err(err, err_len, "internal error");
    }
    if (out) pth_free_scene(out);
    return 3;
}

static int load_obj_impl(const char *obj_path, const char *mtl_dir, uint32_t flags, pth_scene *out, char *err, size_t err_len)
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

    // chunks: about one per hardware thread (<= 16), none below 1 MB; PTH_SMALL_CHUNKS (tests): 256 bytes each, <= 64 threads
    const char *tb = text.c_str(), *te = tb + text.size();
    const bool tiny = (flags & PTH_SMALL_CHUNKS) != 0;
    const size_t hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t want = tiny ? std::max<size_t>(1, std::min<size_t>(64, text.size() / 256)) : std::max<size_t>(1, std::min(hw, text.size() >> 20));
    std::vector<Chunk> cs;
    {
        const char *p = tb;
        for (size_t k = 0; k < want && p < te; k++) {
            const char *q = k + 1 == want ? te : tb + text.size() * (k + 1) / want;
            if (q < p) q = p;
            while (q < te && q > tb && q[-1] != '\n') q++;  // to the next line start
            if (q == p && q < te) continue;
            cs.emplace_back();
            cs.back().b = p; cs.back().e = q;
            p = q;
        }
        if (cs.empty()) cs.emplace_back(), cs.back().b = cs.back().e = tb;
        cs.back().e = te;
    }
    for_chunks(cs, parse_chunk);

    // in file order: counts before each chunk, the material libraries, the material at each chunk's start
    ObjError first;
    std::vector<Material> mats;
    std::unordered_map<std::string, std::vector<std::pair<size_t, int>>> defs;  // name -> (place of the mtllib line, index): later ones override
    size_t vb = 0, lb = 0;
    for (Chunk &c : cs) {
        c.v_before = vb; c.line_before = lb;
        if (c.err.set) note(first, lb + c.err.line, c.err.token, c.err.code, c.err.what);
        vb += c.pos.size() / 3; lb += c.lines;
        for (const Event &ev : c.ev)
            if (ev.lib) {
                std::vector<Material> m2;
                std::unordered_map<std::string, int> n2;
                if (load_mtl(base + "/" + std::string(ev.name), m2, n2)) {  // (tinyobjloader only warns when the file is missing)
                    const size_t at = (size_t)(c.b - tb) + ev.off;
                    for (auto &kv : n2) defs[kv.first].push_back({ at, (int)mats.size() + kv.second });
                    mats.insert(mats.end(), m2.begin(), m2.end());
                }
            }
    }
    for (auto &kv : defs) std::sort(kv.second.begin(), kv.second.end());
    for_chunks(cs, [&](Chunk &c) {
        c.ev_mat.assign(c.ev.size(), -1);
        std::string_view last_name;
        size_t last_at = 0;
        int last_id = -1;
        bool have_last = false;
        for (size_t k = 0; k < c.ev.size(); k++) {
            const Event &ev = c.ev[k];
            if (ev.lib) { have_last = false; continue; }
            const size_t at = (size_t)(c.b - tb) + ev.off;
            if (have_last && ev.name == last_name && defs.size() && at >= last_at) { c.ev_mat[k] = last_id; continue; }
            int id = -1;
            auto it = defs.find(std::string(ev.name));
            if (it != defs.end())
                for (const auto &d : it->second) { if (d.first < at) id = d.second; else break; }
            c.ev_mat[k] = id;
            last_name = ev.name; last_at = at; last_id = id; have_last = true;
        }
    });
    {
        int cur = -1;
        for (Chunk &c : cs) {
            c.mat_in = cur;
            for (size_t k = 0; k < c.ev.size(); k++)
                if (!c.ev[k].lib) cur = c.ev_mat[k];
        }
    }
    // all vertices in one array (relative indices reach back across chunks; the quad rule reads positions)
    std::vector<float> pos(3 * vb);
    for_chunks(cs, [&](Chunk &c) { if (!c.pos.empty()) std::memcpy(pos.data() + 3 * c.v_before, c.pos.data(), sizeof(float) * c.pos.size()); });
    // pass B: indices resolved, polygons cut into triangles
    for_chunks(cs, [&](Chunk &c) {
        int cur = c.mat_in;
        size_t next_ev = 0;
        std::vector<long> poly;
        for (size_t f = 0; f + 1 < c.start.size(); f++) {
            while (next_ev < c.ev.size() && c.ev[next_ev].face <= f) { if (!c.ev[next_ev].lib) cur = c.ev_mat[next_ev]; next_ev++; }
            const long nv = (long)(c.v_before + c.nv_at[f]);
            poly.clear();
            bool bad = false;
            for (uint32_t t = c.start[f]; t < c.start[f + 1]; t++) {
                const long i = c.idx[t];
                const long vi = i > 0 ? i - 1 : nv + i;  // negative = relative to the vertices so far
                if (vi < 0 || vi >= nv) { note(c.err, c.line_of[f], t - c.start[f], 3, "face index out of range"); bad = true; break; }
                poly.push_back(vi);
            }
            if (bad) return;
            if (poly.size() == 4 && (flags & PTH_QUAD_SHORTER_DIAGONAL)) {
                // tinyobjloader >= v2.0.0rc9 (as far as it is known here; the reference's submodule is an unpinned,
                // empty directory): a quad is cut along its SHORTER diagonal, (0,1,2)(0,2,3) when |v0v2|^2 < |v1v3|^2,
                // else (0,1,3)(1,2,3)
                auto d2 = [&](long a, long b) {
                    float s2 = 0.f;
                    for (int k = 0; k < 3; k++) { const float d = pos[3 * (size_t)a + k] - pos[3 * (size_t)b + k]; s2 += d * d; }
                    return s2;
                };
                const bool fan = d2(poly[0], poly[2]) < d2(poly[1], poly[3]);
                const int order[2][6] = { { 0, 1, 3, 1, 2, 3 }, { 0, 1, 2, 0, 2, 3 } };
                for (int k = 0; k < 6; k++) c.tri_idx.push_back((uint32_t)poly[(size_t)order[fan ? 1 : 0][k]]);
                c.tri_mat.push_back(cur);
                c.tri_mat.push_back(cur);
                continue;
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {  // fan: (0,1,2) (0,2,3) ...
                c.tri_idx.push_back((uint32_t)poly[0]);
                c.tri_idx.push_back((uint32_t)poly[k]);
                c.tri_idx.push_back((uint32_t)poly[k + 1]);
                c.tri_mat.push_back(cur);
            }
        }
    });
    size_t nt = 0;
    for (Chunk &c : cs) {
        if (c.err.set) note(first, c.line_before + c.err.line, c.err.token, c.err.code, c.err.what);
        c.tri_before = nt;
        c.n_tri = c.tri_mat.size();
        nt += c.n_tri;
    }
    if (first.set) { set_err(err, err_len, std::string(first.what) + " at line " + std::to_string(first.line)); return first.code; }
    if (nt == 0) { set_err(err, err_len, "no faces in OBJ"); return 4; }
    if (nt > 0xFFFFFFFFull / 3) { set_err(err, err_len, "more than 2^32 / 3 triangles"); return 5; }

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
    for_chunks(cs, [&](Chunk &c) {
        for (size_t j = 0; j < 3 * c.n_tri; j++) {  // main.cpp:39-46
            const size_t i = 3 * c.tri_before + j;
            const uint32_t vi = c.tri_idx[j];
            out->vertices[3 * i + 0] = pos[3 * (size_t)vi + 0];
            out->vertices[3 * i + 1] = -pos[3 * (size_t)vi + 1];  // Y flipped
            out->vertices[3 * i + 2] = pos[3 * (size_t)vi + 2];
            out->indices[i] = (uint32_t)i;
        }
        for (size_t j = 0; j < c.n_tri; j++) {  // main.cpp:47-56
            const size_t t = c.tri_before + j;
            const Material &m = c.tri_mat[j] >= 0 ? mats[(size_t)c.tri_mat[j]] : kNoMaterial;
            for (int k = 0; k < 3; k++) {
                out->faces[6 * t + k] = m.kd[k];
                out->faces[6 * t + 3 + k] = m.ke[k];
            }
        }
    });
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
