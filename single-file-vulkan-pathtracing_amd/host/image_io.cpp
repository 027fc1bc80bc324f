// image_io.cpp -- image output replacing the swapchain copy + present (main.cpp:661-679), and
// the generator of the synthetic triangle soup (BASELINE.json config 5).
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/pt_host.h"

extern "C" int pth_write_ppm_bgra8(const char *path, const uint8_t *bgra, uint32_t w, uint32_t h)
{
    if (!path || !bgra || !w || !h) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
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
    return std::fclose(f) == 0 ? 0 : 3;
}

extern "C" int pth_write_pfm(const char *path, const float *rgb, uint32_t w, uint32_t h)
{
    if (!path || !rgb || !w || !h) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for (uint32_t y = h; y-- > 0;) std::fwrite(rgb + 3 * (size_t)y * w, sizeof(float), 3 * (size_t)w, f);
    return std::fclose(f) == 0 ? 0 : 3;
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
{
    if (!out || !n_tris || n_tris > 0x0FFFFFFFu) return 1;
    *out = pth_scene{};
    float *vert = static_cast<float *>(std::malloc(sizeof(float) * 9 * (size_t)n_tris));
    uint32_t *idx = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * 3 * (size_t)n_tris));
    float *faces = static_cast<float *>(std::malloc(sizeof(float) * 6 * (size_t)n_tris));
    if (!vert || !idx || !faces) { std::free(vert); std::free(idx); std::free(faces); return 2; }
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
    out->vertices = vert; out->n_verts = 3u * n_tris; out->indices = idx; out->n_tris = n_tris; out->faces = faces;
    return 0;
}

// Recipe (frozen): centre c ~ U([-1,1] x [0,2] x [-1,1]) in OBJ space (Y is negated at load, so
// the soup fills the Cornell box volume [-1,1] x [-2,0] x [-1,1] the camera looks into); the
// three vertices are c + 0.02 * (U(-1/2,1/2))^3; triangles with |cross| < 1e-8 are re-drawn;
// material of triangle i = palette[i mod 8], except `light` when i mod 64 == 7.
extern "C" int pth_write_soup_obj(const char *obj_path, uint32_t n_tris, uint32_t seed)
{
    if (!obj_path || !n_tris) return 1;
    std::string obj(obj_path);
    std::string mtl = obj.size() > 4 && obj.substr(obj.size() - 4) == ".obj" ? obj.substr(0, obj.size() - 4) + ".mtl" : obj + ".mtl";
    const size_t slash = mtl.find_last_of('/');
    const std::string mtl_name = slash == std::string::npos ? mtl : mtl.substr(slash + 1);
    FILE *fm = std::fopen(mtl.c_str(), "w");
    if (!fm) return 2;
    const char *const *names = kSoupNames;
    const float (&kd)[8][3] = kSoupKd;
    for (int i = 0; i < 8; i++) std::fprintf(fm, "newmtl %s\nKd %g %g %g\nKe 0 0 0\n\n", names[i], kd[i][0], kd[i][1], kd[i][2]);
    std::fprintf(fm, "newmtl light\nKd 0.78 0.78 0.78\nKe 17 12 4\n");
    std::fclose(fm);

    FILE *f = std::fopen(obj.c_str(), "w");
    if (!f) return 2;
    std::vector<char> buf(1 << 20);
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
    return std::fclose(f) == 0 ? 0 : 3;
}
