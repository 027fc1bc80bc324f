// Sanitizer harness (tests/test_sanitizers.py builds it with -fsanitize=address,undefined): the C++20 host loader and the C oracle -- the two
// pieces of plain CPU code on either side of the HIP path -- over the reference's scene, instanced and NEE modes, a batch of rays through both
// closest-hit modes, and a set of malformed / ragged OBJ texts read whole and in 256-byte chunks.  Exit status 0 and no sanitizer report = pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_host.h"
extern "C" {
#include "../../oracle/pt_oracle.h"
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "san_main: %s failed (line %d)\n", #c, __LINE__); fails++; } } while (0)

static void write_file(const std::string &path, const std::string &text)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: san_main CornellBox-Original.obj scratch_dir\n"); return 2; }
    const std::string dir = argv[2];
    char err[512];
    // ---- the loader on the reference's scene, whole and in small chunks: the same arrays
    pth_scene a{}, b{};
    CHECK(pth_load_obj(argv[1], nullptr, &a, err, sizeof err) == 0);
    CHECK(pth_load_obj_ex(argv[1], nullptr, PTH_SMALL_CHUNKS | PTH_QUAD_SHORTER_DIAGONAL, &b, err, sizeof err) == 0);
    CHECK(a.n_tris == 36 && a.n_verts == 108 && b.n_tris == 36);
    pth_free_scene(&b);
    // ---- malformed and ragged texts: an error message or a scene, never a crash or an out-of-bounds access
    const char *texts[] = {
        "", "v", "v 1", "f 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\nf 1 2\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2// 3/4/5 1x\n",
        "v 1e999 -1e999 nan\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", "mtllib\nusemtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 1 2 3 1 2 3 1 2 3\n", "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n",
        "# only a comment", "f 0 0 0\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 4294967297 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 \\\n",
    };
    int n_ok = 0, n_err = 0;
    for (size_t i = 0; i < sizeof texts / sizeof texts[0]; i++) {
        const std::string path = dir + "/t" + std::to_string(i) + ".obj";
        // (also with the text repeated past a few chunk boundaries)
        for (int rep = 0; rep < 2; rep++) {
            std::string text = texts[i];
            if (rep) for (int k = 0; k < 40; k++) text += std::string("\n# pad ") + std::to_string(k) + "\n" + texts[i];
            write_file(path, text);
            for (uint32_t flags : { 0u, (uint32_t)PTH_SMALL_CHUNKS }) {
                pth_scene s{};
                err[0] = 0;
                const int rc = pth_load_obj_ex(path.c_str(), nullptr, flags, &s, err, sizeof err);
                if (rc == 0) { n_ok++; CHECK(s.n_tris > 0 && s.vertices && s.faces); pth_free_scene(&s); }
                else { n_err++; CHECK(err[0] != 0); }
            }
        }
    }
    CHECK(pth_load_obj((dir + "/does_not_exist.obj").c_str(), nullptr, &b, err, sizeof err) != 0);
    // ---- mutation fuzz: the reference's OBJ with random bytes overwritten, spans deleted or duplicated, truncated -- whole and in 256-byte chunks
    {
        std::string base;
        if (FILE *f = std::fopen(argv[1], "rb")) {
            char buf[4096];
            size_t k;
            while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) base.append(buf, k);
            std::fclose(f);
        }
        CHECK(!base.empty());
        uint32_t fz = 2463534242u;
        auto next = [&]() { fz ^= fz << 13; fz ^= fz >> 17; fz ^= fz << 5; return fz; };
        const char alphabet[] = "0123456789-+.e/ \n\tvfxm#usl\r\\";
        int loaded = 0, refused = 0;
        for (int it = 0; it < 400; it++) {
            std::string t = base;
            const int edits = 1 + (int)(next() % 6);
            for (int e = 0; e < edits && !t.empty(); e++) {
                const size_t at = next() % t.size();
                switch (next() % 5) {
                case 0: t[at] = alphabet[next() % (sizeof alphabet - 1)]; break;
                case 1: t.erase(at, next() % 40); break;
                case 2: t.insert(at, t.substr(next() % t.size(), next() % 60)); break;
                case 3: t.resize(at); break;
                default: t[at] = (char)(next() & 0xFF); break;
                }
            }
            const std::string path = dir + "/fuzz.obj";
            write_file(path, t);
            for (uint32_t flags : { 0u, (uint32_t)PTH_SMALL_CHUNKS }) {
                pth_scene s{};
                err[0] = 0;
                // (the MTL next to the original: materials resolve as for the real file)
                const std::string mdir = std::string(argv[1]).substr(0, std::string(argv[1]).find_last_of('/'));
                if (pth_load_obj_ex(path.c_str(), mdir.c_str(), flags, &s, err, sizeof err) == 0) { loaded++; CHECK(s.n_tris > 0); pth_free_scene(&s); }
                else { refused++; CHECK(err[0] != 0); }
            }
        }
        std::printf("san_main: mutation fuzz: %d loads, %d refusals\n", loaded, refused);
    }
    // ---- the oracle: LBVH, both closest-hit modes on a batch of rays, a small frame in every mode
    orc_scene *sc = orc_scene_create(a.vertices, a.n_verts, a.indices, a.n_tris, a.faces);
    CHECK(sc != nullptr);
    orc_bvh_info info{};
    orc_scene_bvh_info(sc, &info);
    CHECK(info.n_tris == 36 && info.n_nodes == 35);
    std::vector<uint64_t> keys(36); std::vector<uint32_t> order(36), nodes(16 * 35);
    orc_scene_bvh_keys(sc, keys.data(), order.data());
    orc_scene_bvh_nodes(sc, nodes.data());
    const uint32_t n_rays = 4096;
    std::vector<float> rays(6 * (size_t)n_rays);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 747796405u + 2891336453u; return (float)((st >> 9) & 0xFFFF) / 65535.0f; };
    for (uint32_t i = 0; i < n_rays; i++) {
        float *r = &rays[6 * (size_t)i];
        r[0] = rnd() * 2.f - 1.f; r[1] = -rnd() * 2.f; r[2] = rnd() * 6.f - 1.f;
        r[3] = rnd() - 0.5f; r[4] = rnd() - 0.5f; r[5] = rnd() - 0.5f;
        if (i % 97 == 0) { r[3] = 0.f; r[4] = 0.f; }          // axis-parallel
        if (i % 211 == 0) { r[3] = r[4] = r[5] = 0.f; }        // a null direction must not fault either
    }
    std::vector<orc_hit> h0(n_rays), h1(n_rays);
    orc_trace_batch(sc, 0, n_rays, rays.data(), 0.001f, 10000.f, h0.data(), nullptr);
    orc_trace_batch(sc, 1, n_rays, rays.data(), 0.001f, 10000.f, h1.data(), nullptr);
    uint32_t differ = 0;
    for (uint32_t i = 0; i < n_rays; i++)
        if (i % 211 != 0 && std::memcmp(&h0[i], &h1[i], sizeof(orc_hit)) != 0) differ++;
    CHECK(differ == 0);
    orc_params p;
    orc_params_default(&p);
    p.width = 37; p.height = 21; p.spp_per_frame = 3; p.max_depth = 5;
    std::vector<float> img(3 * 37 * 21);
    std::vector<orc_hit> first(37 * 21);
    orc_counters cnt{};
    const uint64_t r1 = orc_render_frame(sc, &p, 1, 3, img.data(), first.data(), &cnt);
    const uint64_t r0 = orc_render_frame(sc, &p, 0, 1, img.data(), nullptr, nullptr);
    CHECK(r0 == r1 && r1 >= 37u * 21u * 3u);
    std::vector<float> crop(3 * 5 * 4);
    CHECK(orc_render_rect(sc, &p, 1, 2, 30, 15, 5, 4, crop.data(), nullptr, nullptr) > 0);
    p.nee = 1;
    CHECK(orc_render_frame(sc, &p, 1, 2, img.data(), nullptr, nullptr) > r1);
    p.nee = 0;
    // two-level: a few instances, then back
    const float xf[2][12] = { { 0.5f, 0, 0, -0.6f, 0, 0.5f, 0, -0.5f, 0, 0, 0.5f, 0 }, { 0, -0.4f, 0, 0.6f, 0.4f, 0, 0, -1.2f, 0, 0, 0.4f, 0.2f } };
    CHECK(orc_scene_set_instances(sc, &xf[0][0], 2) == 0);
    CHECK(orc_render_frame(sc, &p, 1, 2, img.data(), nullptr, nullptr) >= 37u * 21u * 3u);
    CHECK(orc_scene_set_instances(sc, nullptr, 0) == 0);
    orc_scene_destroy(sc);
    pth_free_scene(&a);
    std::printf("san_main: %d texts loaded, %d refused with a message, %u rays x 2 modes, %llu rays rendered; %d check(s) failed\n", n_ok, n_err, n_rays,
                (unsigned long long)r1, fails);
    return fails ? 1 : 0;
}
