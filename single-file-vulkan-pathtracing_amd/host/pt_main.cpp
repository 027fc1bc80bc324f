// pt_main.cpp -- headless C++20 host driver: the reference's main() (main.cpp:457-690) without
// Vulkan/GLFW.  Same order of events: load OBJ/MTL (main.cpp:490) -> upload + build the
// acceleration structure (492-538, here pt_scene_create) -> frame loop with the `frame` push
// constant (645-685, here pt_render) -> instead of presenting, write the image to disk.
//
//   pt_main [--obj assets/CornellBox-Original.obj] [--width 1024] [--height 1024]
//           [--frames 1] [--spp 32] [--depth 8] [--device 0] [--batch N]
//           [--ppm out.ppm] [--pfm out.pfm]
// Prints one JSON line with ray count, ms/frame and Mrays/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "../../include/pt_host.h"

namespace {
[[noreturn]] void die(const std::string &msg)
{
    std::fprintf(stderr, "pt_main: %s\n", msg.c_str());
    std::exit(1);
}
}  // namespace

int main(int argc, char **argv)
{
    std::string obj = "assets/CornellBox-Original.obj", ppm, pfm;
    uint32_t width = 1024, height = 1024, frames = 1, spp = 32, depth = 8, batch = 0;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) die("missing value for " + a);
            return argv[++i];
        };
        if (a == "--obj") obj = val();
        else if (a == "--width") width = (uint32_t)std::atoi(val());
        else if (a == "--height") height = (uint32_t)std::atoi(val());
        else if (a == "--frames") frames = (uint32_t)std::atoi(val());
        else if (a == "--spp") spp = (uint32_t)std::atoi(val());
        else if (a == "--depth") depth = (uint32_t)std::atoi(val());
        else if (a == "--device") device = std::atoi(val());
        else if (a == "--batch") batch = (uint32_t)std::atoi(val());
        else if (a == "--ppm") ppm = val();
        else if (a == "--pfm") pfm = val();
        else die("unknown option " + a);
    }

    char err[512] = { 0 };
    pth_scene hs{};
    const auto t0 = std::chrono::steady_clock::now();
    if (pth_load_obj(obj.c_str(), nullptr, &hs, err, sizeof(err)) != 0) die(err);
    const auto t1 = std::chrono::steady_clock::now();

    pt_ctx *ctx = nullptr;
    if (pt_ctx_create(device, nullptr, &ctx) != PT_OK) die(pt_last_error(nullptr));
    pt_scene *scene = nullptr;
    if (pt_scene_create(ctx, hs.vertices, hs.n_verts, hs.indices, hs.n_tris, hs.faces, &scene) != PT_OK) die(pt_last_error(ctx));
    pt_scene_info info{};
    pt_scene_get_info(scene, &info);
    pt_film *film = nullptr;
    if (pt_film_create(ctx, width, height, &film) != PT_OK) die(pt_last_error(ctx));

    pt_params p;
    pt_params_default(&p);
    p.width = width; p.height = height; p.spp_per_frame = spp; p.max_depth = depth;
    p.frames_in_flight = batch;
    // the reference dispatches one frame per loop iteration (main.cpp:647-685); frames are
    // independent until the blend, so they are handed over in one call and batched on the device
    p.frame = 0; p.frame_count = frames;
    if (pt_render(scene, film, &p) != PT_OK) die(pt_last_error(ctx));
    pt_stats st{};
    pt_get_stats(ctx, &st);

    if (!ppm.empty()) {
        std::vector<uint8_t> img(4 * (size_t)width * height);
        if (pt_film_read_bgra8(film, img.data()) != PT_OK) die(pt_last_error(ctx));
        if (pth_write_ppm_bgra8(ppm.c_str(), img.data(), width, height) != 0) die("cannot write " + ppm);
    }
    if (!pfm.empty()) {
        std::vector<float> img(3 * (size_t)width * height);
        if (pt_film_read_f32(film, img.data()) != PT_OK) die(pt_last_error(ctx));
        if (pth_write_pfm(pfm.c_str(), img.data(), width, height) != 0) die("cannot write " + pfm);
    }
    const double load_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    std::printf("{\"obj\": \"%s\", \"triangles\": %u, \"bvh_nodes\": %u, \"bvh_height\": %u, \"load_ms\": %.3f, "
                "\"bvh_build_ms\": %.3f, \"width\": %u, \"height\": %u, \"frames\": %u, \"spp_per_frame\": %u, "
                "\"max_depth\": %u, \"rays\": %llu, \"paths\": %llu, \"rounds\": %u, \"ms_total\": %.3f, "
                "\"ms_per_frame\": %.3f, \"mrays_per_s\": %.1f}\n",
                obj.c_str(), info.n_tris, info.n_nodes, info.bvh_height, load_ms, info.build_ms, width, height, frames, spp,
                depth, (unsigned long long)st.rays, (unsigned long long)st.paths, st.rounds, st.ms_total,
                st.ms_total / frames, st.ms_total > 0 ? (double)st.rays / (st.ms_total * 1e3) : 0.0);

    pt_film_destroy(film);
    pt_scene_destroy(scene);
    pt_ctx_destroy(ctx);
    pth_free_scene(&hs);
    return 0;
}
