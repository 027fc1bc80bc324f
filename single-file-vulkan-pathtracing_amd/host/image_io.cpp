// image_io.cpp -- image output replacing the swapchain copy + present (main.cpp:661-679), and
// the generator of the synthetic triangle soup (BASELINE.json config 5).
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <memory>

#include "../../include/pt_host.h"

namespace {
// what the try / catch wrappers of the C-ABI need so that an exception (std::bad_alloc of a row buffer) leaks neither a FILE nor a malloc block
struct FileCloser { void operator()(FILE *f) const { if (f) std::fclose(f); } };
using File = std::unique_ptr<FILE, FileCloser>;
struct Freer { void operator()(void *p) const { std::free(p); } };
template <class T> using Block = std::unique_ptr<T, Freer>;
// the explicit close whose result the writers report (the guard then holds nothing)
int close_checked(File &f) { return std::fclose(f.release()) == 0 ? 0 : 3; }
}  // namespace

extern "C" int pth_write_ppm_bgra8(const char *path, const uint8_t *bgra, uint32_t w, uint32_t h)
try {
    if (!path || !bgra || !w || !h) return 1;
    File fg(std::fopen(path, "wb"));
    if (!fg) return 2;
    FILE *f = fg.get();
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    std::vector<uint8_t> row(3 * (size_t)w);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *s = bgra + 4 * (size_t)y * w;
        for (uint32_t x = 0; x < w; x++) {
            row[3 * x + 0] = s[4 * x + 2];
            row[3 * x + 1] = s[4 * x + 1];
            row[3 * x + 2] = s[4 * x + 0];
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    return close_checked(fg);
} catch (...) {
    return 3;  // (std::bad_alloc and the like: nothing leaves the C-ABI as an exception)
}

extern "C" int pth_write_pfm(const char *path, const float *rgb, uint32_t w, uint32_t h)
try {
    if (!path || !rgb || !w || !h) return 1;
    File fg(std::fopen(path, "wb"));
    if (!fg) return 2;
    FILE *f = fg.get();
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for (uint32_t y = h; y-- > 0;) std::fwrite(rgb + 3 * (size_t)y * w, sizeof(float), 3 * (size_t)w, f);
    return close_checked(fg);
} catch (...) {
    return 3;  // (std::bad_alloc and the like: nothing leaves the C-ABI as an exception)
}

namespace {
// PCG-RXS-M-XS-32, same generator family the shaders use (common.glsl:13-19)
struct Pcg {
    uint32_t s;
    uint32_t next()
    {
        s = s * 747796405u + 2891336453u;
        const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (w >> 22u) ^ w;
    }
    float uni() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }  // [0,1), 24 bits
};
const char *const kSoupNames[8] = { "white", "red", "green", "blue", "yellow", "cyan", "magenta", "grey" };
const float kSoupKd[8][3] = { { .725f, .71f, .68f }, { .63f, .065f, .05f }, { .14f, .45f, .091f }, { .1f, .2f, .7f },
                              { .7f, .7f, .1f },     { .1f, .7f, .7f },    { .7f, .1f, .7f },      { .4f, .4f, .4f } };
const float kSoupLightKd[3] = { 0.78f, 0.78f, 0.78f }, kSoupLightKe[3] = { 17.f, 12.f, 4.f };

// one triangle of the soup (OBJ space), consuming the generator exactly as the recipe below says
void soup_triangle(Pcg &rng, float (&v)[3][3])
{
    for (;;) {
        const float c[3] = { rng.uni() * 2.f - 1.f, rng.uni() * 2.f, rng.uni() * 2.f - 1.f };
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) v[k][a] = c[a] + 0.02f * (rng.uni() - 0.5f);
        const float e1[3] = { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] };
        const float e2[3] = { v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2] };
        const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        if (cx * cx + cy * cy + cz * cz >= 1e-16f) return;
    }
}
}  // namespace

// The same soup as pth_write_soup_obj + pth_load_obj would give, without the detour through ~140 bytes of text per
// triangle: for the scenes larger than the Infinity Cache (bench.py --config c5x, 8 M triangles = 1.1 GB of OBJ).
extern "C" int pth_make_soup(uint32_t n_tris, uint32_t seed, pth_scene *out)
try {
    if (!out || !n_tris || n_tris > 0x0FFFFFFFu) return 1;
    *out = pth_scene{};
    Block<float> vert_g(static_cast<float *>(std::malloc(sizeof(float) * 9 * (size_t)n_tris)));
    Block<uint32_t> idx_g(static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * 3 * (size_t)n_tris)));
    Block<float> faces_g(static_cast<float *>(std::malloc(sizeof(float) * 6 * (size_t)n_tris)));
    if (!vert_g || !idx_g || !faces_g) return 2;
    float *vert = vert_g.get(), *faces = faces_g.get();
    uint32_t *idx = idx_g.get();
    Pcg rng{ seed };
    for (uint32_t i = 0; i < n_tris; i++) {
        float v[3][3];
        soup_triangle(rng, v);
        for (int k = 0; k < 3; k++) {
            vert[9 * (size_t)i + 3 * k + 0] = v[k][0];
            vert[9 * (size_t)i + 3 * k + 1] = -v[k][1];  // main.cpp:42
            vert[9 * (size_t)i + 3 * k + 2] = v[k][2];
            idx[3 * (size_t)i + k] = 3u * i + (uint32_t)k;
        }
        const bool light = i % 64u == 7u;
        const float *kd = light ? kSoupLightKd : kSoupKd[i % 8u];
        for (int a = 0; a < 3; a++) {
            faces[6 * (size_t)i + a] = kd[a];
            faces[6 * (size_t)i + 3 + a] = light ? kSoupLightKe[a] : 0.f;
        }
    }
    out->vertices = vert_g.release(); out->n_verts = 3u * n_tris; out->indices = idx_g.release(); out->n_tris = n_tris; out->faces = faces_g.release();
    return 0;
} catch (...) {
    return 3;  // (std::bad_alloc and the like: nothing leaves the C-ABI as an exception)
}

// ---- the "teapot in a stadium" stress scene: primitive sizes over four orders of magnitude ------------------------------
// A Cornell-sized room (OBJ space x in [-1,1], y in [0,2], z in [-1,1]; Y is negated like loadFromFile does, main.cpp:42)
// whose FLOOR is a floor_side x floor_side grid of quads with PCG-jittered heights (+- a tenth of a tile; 2 floor_side^2
// triangles of edge 2/floor_side), four walls and a ceiling of two triangles each, the Cornell light, two rotated boxes of 12 triangles
// and a finely tessellated sphere ("the teapot": radius 0.25 on the short box, sphere_seg x 2 sphere_seg quads).  Frozen
// recipe: floor heights from the PCG stream seeded 7 in row-major vertex order; everything else is closed form.  Triangles
// are wound so that the shader's geometric normal (closesthit.rchit:43-48, after the Y flip) faces the room / away from
// the solids.  What it is for: spatial-median (Morton) splits cut the sphere and the floor tiles with planes chosen by
// the room's size; a surface-area-class builder isolates them first (pt_scene_set_bvh_quality).
namespace {
struct StadiumOut {
    std::vector<float> vert, faces;
    // one triangle in OBJ space; `toward`: the point the shader normal should face (inside = true) or face away from
    void tri(const float a[3], const float b[3], const float c[3], const float kd[3], const float ke[3], const float ref[3], bool toward)
    {
        // flipped space (y negated), shader normal n = -cross(b - a, c - a)
        const float A[3] = { a[0], -a[1], a[2] }, B[3] = { b[0], -b[1], b[2] }, C[3] = { c[0], -c[1], c[2] }, R[3] = { ref[0], -ref[1], ref[2] };
        const float e1[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] }, e2[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
        const float n[3] = { -(e1[1] * e2[2] - e1[2] * e2[1]), -(e1[2] * e2[0] - e1[0] * e2[2]), -(e1[0] * e2[1] - e1[1] * e2[0]) };
        const float g[3] = { (A[0] + B[0] + C[0]) / 3.f, (A[1] + B[1] + C[1]) / 3.f, (A[2] + B[2] + C[2]) / 3.f };
        const float d = n[0] * (R[0] - g[0]) + n[1] * (R[1] - g[1]) + n[2] * (R[2] - g[2]);
        const bool swap = toward ? d < 0.f : d > 0.f;
        const float *p1 = swap ? C : B, *p2 = swap ? B : C;
        for (int k = 0; k < 3; k++) vert.push_back(A[k]);
        for (int k = 0; k < 3; k++) vert.push_back(p1[k]);
        for (int k = 0; k < 3; k++) vert.push_back(p2[k]);
        for (int k = 0; k < 3; k++) faces.push_back(kd[k]);
        for (int k = 0; k < 3; k++) faces.push_back(ke[k]);
    }
    void quad(const float a[3], const float b[3], const float c[3], const float d[3], const float kd[3], const float ke[3], const float ref[3], bool toward)
    {
        tri(a, b, c, kd, ke, ref, toward);
        tri(a, c, d, kd, ke, ref, toward);
    }
};
}  // namespace

extern "C" int pth_make_stadium(uint32_t floor_side, uint32_t sphere_seg, pth_scene *out)
try {
    if (!out || floor_side < 1u || sphere_seg < 3u || floor_side > 8192u || sphere_seg > 4096u) return 1;
    *out = pth_scene{};
    StadiumOut o;
    try {
        const float zero[3] = { 0.f, 0.f, 0.f }, white[3] = { .725f, .71f, .68f }, red[3] = { .63f, .065f, .05f }, green[3] = { .14f, .45f, .091f };
        const float grey[3] = { .4f, .4f, .4f }, blue[3] = { .1f, .2f, .7f };
        const float centre[3] = { 0.f, 1.f, 0.f };
        // floor: jittered height field, two greys in a checker pattern
        Pcg rng{ 7u };
        const uint32_t nv = floor_side + 1u;
        std::vector<float> hgt((size_t)nv * nv);
        const float step = 2.0f / (float)floor_side;
        for (auto &h : hgt) h = 0.2f * step * (rng.uni() - 0.5f);   // +- a tenth of a tile
        for (uint32_t j = 0; j < floor_side; j++)
            for (uint32_t i = 0; i < floor_side; i++) {
                const float x0 = -1.f + step * (float)i, x1 = -1.f + step * (float)(i + 1u), z0 = -1.f + step * (float)j, z1 = -1.f + step * (float)(j + 1u);
                const float a[3] = { x0, hgt[(size_t)j * nv + i], z0 }, b[3] = { x1, hgt[(size_t)j * nv + i + 1u], z0 };
                const float c[3] = { x1, hgt[(size_t)(j + 1u) * nv + i + 1u], z1 }, d[3] = { x0, hgt[(size_t)(j + 1u) * nv + i], z1 };
                o.quad(a, b, c, d, ((i + j) & 1u) ? grey : white, zero, centre, true);
            }
        // ceiling, back wall, right (green) and left (red) walls: two triangles each -- the large primitives
        {
            const float c0[3] = { -1, 2, -1 }, c1[3] = { 1, 2, -1 }, c2[3] = { 1, 2, 1 }, c3[3] = { -1, 2, 1 };
            o.quad(c0, c1, c2, c3, white, zero, centre, true);
            const float b0[3] = { -1, 0, -1 }, b1[3] = { 1, 0, -1 }, b2[3] = { 1, 2, -1 }, b3[3] = { -1, 2, -1 };
            o.quad(b0, b1, b2, b3, white, zero, centre, true);
            const float r0[3] = { 1, 0, -1 }, r1[3] = { 1, 0, 1 }, r2[3] = { 1, 2, 1 }, r3[3] = { 1, 2, -1 };
            o.quad(r0, r1, r2, r3, green, zero, centre, true);
            const float l0[3] = { -1, 0, -1 }, l1[3] = { -1, 0, 1 }, l2[3] = { -1, 2, 1 }, l3[3] = { -1, 2, -1 };
            o.quad(l0, l1, l2, l3, red, zero, centre, true);
            // the Cornell light (Ke 17 12 4), facing down
            const float q0[3] = { -0.24f, 1.98f, -0.22f }, q1[3] = { 0.23f, 1.98f, -0.22f }, q2[3] = { 0.23f, 1.98f, 0.16f }, q3[3] = { -0.24f, 1.98f, 0.16f };
            o.quad(q0, q1, q2, q3, kSoupLightKd, kSoupLightKe, centre, true);
        }
        // two boxes (half extents hx, hy, hz about (cx, hy, cz), rotated by `ang` about y): 12 triangles each
        auto box = [&](float cx, float cz, float hx, float hy, float hz, float ang) {
            const float ca = std::cos(ang), sa = std::sin(ang);
            float v[8][3];
            for (int k = 0; k < 8; k++) {
                const float lx = (k & 1) ? hx : -hx, ly = (k & 2) ? 2.f * hy : 0.004f, lz = (k & 4) ? hz : -hz;
                v[k][0] = cx + ca * lx + sa * lz; v[k][1] = ly; v[k][2] = cz - sa * lx + ca * lz;
            }
            const float mid[3] = { cx, hy, cz };
            const int f[6][4] = { { 0, 1, 3, 2 }, { 4, 5, 7, 6 }, { 0, 1, 5, 4 }, { 2, 3, 7, 6 }, { 0, 2, 6, 4 }, { 1, 3, 7, 5 } };
            for (auto &q : f) o.quad(v[q[0]], v[q[1]], v[q[2]], v[q[3]], white, zero, mid, false);
        };
        box(0.33f, 0.35f, 0.3f, 0.3f, 0.3f, -0.29f);    // short box
        box(-0.35f, -0.3f, 0.3f, 0.6f, 0.3f, 0.3f);     // tall box
        // the "teapot": a sphere of radius 0.25 resting on the short box
        const float sc[3] = { 0.33f, 0.6f + 0.25f, 0.35f }, rad = 0.25f;
        const uint32_t nlat = sphere_seg, nlon = 2u * sphere_seg;
        auto sp = [&](uint32_t a, uint32_t b, float p[3]) {
            const double th = 3.14159265358979323846 * (double)a / (double)nlat, ph = 2.0 * 3.14159265358979323846 * (double)(b % nlon) / (double)nlon;
            p[0] = sc[0] + rad * (float)(std::sin(th) * std::cos(ph));
            p[1] = sc[1] + rad * (float)std::cos(th);
            p[2] = sc[2] + rad * (float)(std::sin(th) * std::sin(ph));
        };
        for (uint32_t a = 0; a < nlat; a++)
            for (uint32_t b = 0; b < nlon; b++) {
                float p00[3], p01[3], p10[3], p11[3];
                sp(a, b, p00); sp(a, b + 1u, p01); sp(a + 1u, b, p10); sp(a + 1u, b + 1u, p11);
                if (a > 0u) o.tri(p00, p10, p01, blue, zero, sc, false);           // (the pole caps are single triangles)
                if (a + 1u < nlat) o.tri(p01, p10, p11, blue, zero, sc, false);
            }
    } catch (...) {
        return 2;
    }
    const size_t nt = o.faces.size() / 6;
    if (nt == 0 || nt > 0x0FFFFFFFu) return 1;
    Block<float> vert_g(static_cast<float *>(std::malloc(sizeof(float) * o.vert.size())));
    Block<uint32_t> idx_g(static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * 3 * nt)));
    Block<float> faces_g(static_cast<float *>(std::malloc(sizeof(float) * o.faces.size())));
    if (!vert_g || !idx_g || !faces_g) return 2;
    float *vert = vert_g.get(), *faces = faces_g.get();
    uint32_t *idx = idx_g.get();
    std::memcpy(vert, o.vert.data(), sizeof(float) * o.vert.size());   // (already in loaded space: y negated by StadiumOut::tri)
    std::memcpy(faces, o.faces.data(), sizeof(float) * o.faces.size());
    for (size_t i = 0; i < 3 * nt; i++) idx[i] = (uint32_t)i;
    out->vertices = vert_g.release(); out->n_verts = (uint32_t)(3 * nt); out->indices = idx_g.release(); out->n_tris = (uint32_t)nt; out->faces = faces_g.release();
    return 0;
} catch (...) {
    return 3;  // (std::bad_alloc and the like: nothing leaves the C-ABI as an exception)
}

// Recipe (frozen): centre c ~ U([-1,1] x [0,2] x [-1,1]) in OBJ space (Y is negated at load, so
// the soup fills the Cornell box volume [-1,1] x [-2,0] x [-1,1] the camera looks into); the
// three vertices are c + 0.02 * (U(-1/2,1/2))^3; triangles with |cross| < 1e-8 are re-drawn;
// material of triangle i = palette[i mod 8], except `light` when i mod 64 == 7.
extern "C" int pth_write_soup_obj(const char *obj_path, uint32_t n_tris, uint32_t seed)
try {
    if (!obj_path || !n_tris) return 1;
    std::string obj(obj_path);
    std::string mtl = obj.size() > 4 && obj.substr(obj.size() - 4) == ".obj" ? obj.substr(0, obj.size() - 4) + ".mtl" : obj + ".mtl";
    const size_t slash = mtl.find_last_of('/');
    const std::string mtl_name = slash == std::string::npos ? mtl : mtl.substr(slash + 1);
    File fmg(std::fopen(mtl.c_str(), "w"));
    if (!fmg) return 2;
    FILE *fm = fmg.get();
    const char *const *names = kSoupNames;
    const float (&kd)[8][3] = kSoupKd;
    for (int i = 0; i < 8; i++) std::fprintf(fm, "newmtl %s\nKd %g %g %g\nKe 0 0 0\n\n", names[i], kd[i][0], kd[i][1], kd[i][2]);
    std::fprintf(fm, "newmtl light\nKd 0.78 0.78 0.78\nKe 17 12 4\n");
    if (close_checked(fmg)) return 3;

    std::vector<char> buf(1 << 20);   // (declared before the FILE that uses it as its buffer: destroyed after the FILE is closed)
    File fg(std::fopen(obj.c_str(), "w"));
    if (!fg) return 2;
    FILE *f = fg.get();
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::fprintf(f, "# synthetic triangle soup, %u triangles, seed %u\nmtllib %s\n", n_tris, seed, mtl_name.c_str());
    Pcg rng{ seed };
    int last = -1;
    for (uint32_t i = 0; i < n_tris; i++) {
        float v[3][3];
        soup_triangle(rng, v);
        const int m = (i % 64u == 7u) ? 8 : (int)(i % 8u);
        if (m != last) {
            std::fprintf(f, "usemtl %s\n", m == 8 ? "light" : names[m]);
            last = m;
        }
        for (int k = 0; k < 3; k++) std::fprintf(f, "v %.9g %.9g %.9g\n", v[k][0], v[k][1], v[k][2]);
        std::fprintf(f, "f -3 -2 -1\n");
    }
    return close_checked(fg);
} catch (...) {
    return 3;  // (std::bad_alloc and the like: nothing leaves the C-ABI as an exception)
}
