// pt_main.cpp -- headless C++20 host driver: the reference's main() (main.cpp:457-690) without
// Vulkan/GLFW.  Same order of events: load OBJ/MTL (main.cpp:490) -> upload + build the
// acceleration structure (492-538, here pt_scene_create) -> frame loop with the `frame` push
// constant (645-685, here pt_render) -> instead of presenting, write the image to disk.
//
//   pt_main [--obj assets/CornellBox-Original.obj] [--width 1024] [--height 1024]
//           [--frames 1] [--spp 32] [--depth 8] [--device 0] [--batch N]
//           [--ppm out.ppm] [--pfm out.pfm] [--pipeline auto|wavefront|fused|nee]
//           [--ranks N [--devices 0,1,...] [--selftest]]
// --ranks N renders with N GPUs: one host thread and one context per GPU, the 8x8 pixel tiles interleaved over the
// ranks (pt_params.rank/world), and ONE RCCL gather of the packed tiles to rank 0 per presented image
// (pt_film_present); the image written is the presented one.  --selftest: before rendering, every rank presents a film whose own tiles
// carry its colour through the same collective and rank 0 checks that every tile arrived from its owner (and that RCCL connected N ranks).
// Prints one JSON line with ray count, ms/frame and Mrays/s.
#include <algorithm>
#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_api.h"
#include "../../include/pt_host.h"

namespace {
[[noreturn]] void die(const std::string &msg)
{
    std::fprintf(stderr, "pt_main: %s\n", msg.c_str());
    std::exit(1);
}

struct Options {
    std::string obj = "assets/CornellBox-Original.obj", ppm, pfm;
    uint32_t width = 1024, height = 1024, frames = 1, spp = 32, depth = 8, batch = 0;
    int device = 0;
    uint32_t pipeline = PT_PIPELINE_AUTO;
    uint32_t ranks = 1;          // --ranks N: one host thread + one GPU per rank, tiles interleaved, RCCL gather to rank 0
    std::vector<int> devices;    // --devices a,b,...: HIP ordinals of the ranks (default 0..N-1)
    bool selftest = false;       // --selftest (with --ranks N): the presentation collective on a rank-coloured film before the render
};

// --ranks N: the ranks agree on success before every collective (ncclCommInitRank, the gather inside pt_film_present): a rank
// that failed on its own -- bad ordinal, out of memory, a failed render -- would otherwise leave its peers waiting inside the
// collective for ever.  Every rank arrives at the barrier whether it failed or not; after it, all of them see the flag.
struct Agreement {
    std::barrier<> sync;
    std::atomic<bool> failed{ false };
    explicit Agreement(uint32_t n) : sync((std::ptrdiff_t)n) {}
    bool all_ok(bool mine_ok)
    {
        if (!mine_ok) failed.store(true);
        sync.arrive_and_wait();
        return !failed.load();
    }
};

struct RankResult {
    pt_stats st{};
    pt_scene_info info{};
    uint32_t rccl_ranks = 0;
    long long selftest_wrong = -1;   // rank 0 after --selftest: pixels of the presented test film that did not carry their owner's colour
    double render_ms = 0.0, present_ms = 0.0;
    std::string error;
};

// What one rank does: the reference's main() from the scene upload on (main.cpp:492-685), for its share of the
// tiles; then the one collective per presented image.  Rank 0 returns the image in `image` (device -> host).
void run_rank(const Options &o, const pth_scene &hs, uint32_t rank, const pt_unique_id *id, RankResult &res,
              std::vector<float> *image_f32, std::vector<uint8_t> *image_bgra8, Agreement *agree)
{
    pt_ctx *ctx = nullptr;
    pt_scene *scene = nullptr;
    pt_film *film = nullptr;
    pt_comm *comm = nullptr;
    float *d_image = nullptr;
    auto fail = [&](const char *what) {
        res.error = std::string(what) + ": " + (ctx ? pt_last_error(ctx) : pt_last_error(nullptr));
    };
    const int device = o.ranks > 1 ? o.devices[rank] : o.device;
    // peers_ok(ok): single rank -> ok; else every rank's verdict (a rank that already failed keeps arriving at the barriers)
    auto peers_ok = [&](bool ok) { return agree ? agree->all_ok(ok) : ok; };
    const char *peer_msg = "stopped: another rank failed";
    do {
        bool ok = true;
        if (pt_ctx_create(device, nullptr, &ctx) != PT_OK) { fail("pt_ctx_create"); ok = false; }
        else if (pt_scene_create(ctx, hs.vertices, hs.n_verts, hs.indices, hs.n_tris, hs.faces, &scene) != PT_OK) { fail("pt_scene_create"); ok = false; }
        else if (pt_film_create(ctx, o.width, o.height, &film) != PT_OK) { fail("pt_film_create"); ok = false; }
        if (ok) pt_scene_get_info(scene, &res.info);
        if (!peers_ok(ok)) { if (ok) res.error = peer_msg; break; }  // (the flag is sticky: every rank leaves here together)
        if (o.ranks > 1) {
            // collective: all ranks are here.  RCCL refuses two ranks on one device (--devices 0,0) on every rank alike
            if (pt_comm_create(ctx, id, o.ranks, rank, &comm) != PT_OK) { fail("pt_comm_create"); ok = false; }
            else pt_comm_ranks(comm, &res.rccl_ranks);
        }
        if (!peers_ok(ok)) { if (ok) res.error = peer_msg; break; }
        if (o.selftest && o.ranks > 1) {
            // the film of distributed.py selftest_film: own tiles (rank + 1, 100 + rank, tile x + 1000 tile y), zero elsewhere -- presented
            // through the communicator the run is about to use; rank 0 holds every pixel against the colour of the rank that owns its tile
            const size_t np = (size_t)o.width * o.height;
            std::vector<float> host(3 * np, 0.f);
            for (uint32_t y = 0; y < o.height; y++)
                for (uint32_t x = 0; x < o.width; x++)
                    if ((x / 8 + y / 8) % o.ranks == rank) {
                        float *c = &host[3 * ((size_t)y * o.width + x)];
                        c[0] = (float)(rank + 1); c[1] = (float)(100 + rank); c[2] = (float)(x / 8 + 1000 * (y / 8));
                    }
            void *d_test = nullptr, *d_out = nullptr;
            pt_film *tf = nullptr;
            if (pt_device_alloc(ctx, sizeof(float) * 3 * np, &d_test) != PT_OK || pt_device_write(ctx, d_test, host.data(), sizeof(float) * 3 * np) != PT_OK ||
                pt_film_create_external(ctx, o.width, o.height, d_test, &tf) != PT_OK ||
                (rank == 0 && pt_device_alloc(ctx, sizeof(float) * 3 * np, &d_out) != PT_OK)) { fail("selftest set-up"); ok = false; }
            if (peers_ok(ok)) {
                if (pt_film_present(tf, comm, 0, (float *)d_out) != PT_OK) { fail("selftest pt_film_present"); ok = false; }
                else if (rank == 0) {
                    if (pt_device_read(ctx, d_out, host.data(), sizeof(float) * 3 * np) != PT_OK) { fail("selftest pt_device_read"); ok = false; }
                    else {
                        long long wrong = 0;
                        for (uint32_t y = 0; y < o.height; y++)
                            for (uint32_t x = 0; x < o.width; x++) {
                                const uint32_t owner = (x / 8 + y / 8) % o.ranks;
                                const float *c = &host[3 * ((size_t)y * o.width + x)];
                                wrong += !(c[0] == (float)(owner + 1) && c[1] == (float)(100 + owner) && c[2] == (float)(x / 8 + 1000 * (y / 8)));
                            }
                        res.selftest_wrong = wrong;
                        if (wrong != 0 || res.rccl_ranks != o.ranks) { res.error = "selftest: " + std::to_string(wrong) + " pixels without their owner's colour, RCCL connected " + std::to_string(res.rccl_ranks) + " ranks"; ok = false; }
                    }
                }
            } else if (ok) { res.error = peer_msg; ok = false; }
            pt_film_destroy(tf);
            if (d_test) pt_device_free(ctx, d_test);
            if (d_out) pt_device_free(ctx, d_out);
            if (!peers_ok(ok)) { if (ok) res.error = peer_msg; break; }
        }
        pt_params p;
        pt_params_default(&p);
        p.width = o.width; p.height = o.height; p.spp_per_frame = o.spp; p.max_depth = o.depth;
        p.frames_in_flight = o.batch;
        p.pipeline = o.pipeline;
        p.rank = rank; p.world = o.ranks;
        // the reference dispatches one frame per loop iteration (main.cpp:647-685); frames are
        // independent until the blend, so they are handed over in one call and batched on the device
        p.frame = 0; p.frame_count = o.frames;
        const auto t0 = std::chrono::steady_clock::now();
        if (pt_render(scene, film, &p) != PT_OK) { fail("pt_render"); ok = false; }
        const auto t1 = std::chrono::steady_clock::now();
        res.render_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (ok) pt_get_stats(ctx, &res.st);
        // the presented image lives in its own device buffer on rank 0 (main.cpp:661-667 copies the storage image)
        if (ok && o.ranks > 1 && rank == 0 && pt_device_alloc(ctx, sizeof(float) * 3 * (size_t)o.width * o.height, (void **)&d_image) != PT_OK) { fail("pt_device_alloc"); ok = false; }
        if (!peers_ok(ok)) { if (ok) res.error = peer_msg; break; }
        if (o.ranks > 1) {
            if (pt_film_present(film, comm, 0, d_image) != PT_OK) { fail("pt_film_present"); break; }
            res.present_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            if (rank == 0 && image_f32) {
                image_f32->resize(3 * (size_t)o.width * o.height);
                if (pt_device_read(ctx, d_image, image_f32->data(), sizeof(float) * image_f32->size()) != PT_OK) { fail("pt_device_read"); break; }
            }
        } else {
            if (image_f32) {
                image_f32->resize(3 * (size_t)o.width * o.height);
                if (pt_film_read_f32(film, image_f32->data()) != PT_OK) { fail("pt_film_read_f32"); break; }
            }
            if (image_bgra8) {
                image_bgra8->resize(4 * (size_t)o.width * o.height);
                if (pt_film_read_bgra8(film, image_bgra8->data()) != PT_OK) { fail("pt_film_read_bgra8"); break; }
            }
        }
    } while (false);
    if (d_image) pt_device_free(ctx, d_image);
    pt_comm_destroy(comm);
    pt_film_destroy(film);
    pt_scene_destroy(scene);
    pt_ctx_destroy(ctx);
}
}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) die("missing value for " + a);
            return argv[++i];
        };
        if (a == "--obj") o.obj = val();
        else if (a == "--width") o.width = (uint32_t)std::atoi(val());
        else if (a == "--height") o.height = (uint32_t)std::atoi(val());
        else if (a == "--frames") o.frames = (uint32_t)std::atoi(val());
        else if (a == "--spp") o.spp = (uint32_t)std::atoi(val());
        else if (a == "--depth") o.depth = (uint32_t)std::atoi(val());
        else if (a == "--device") o.device = std::atoi(val());
        else if (a == "--batch") o.batch = (uint32_t)std::atoi(val());
        else if (a == "--ranks") o.ranks = (uint32_t)std::max(1, std::atoi(val()));
        else if (a == "--selftest") o.selftest = true;
        else if (a == "--devices") {
            std::string v = val();
            for (size_t b = 0; b <= v.size();) {
                const size_t e = std::min(v.find(',', b), v.size());
                o.devices.push_back(std::atoi(v.substr(b, e - b).c_str()));
                b = e + 1;
            }
        }
        else if (a == "--pipeline") {  // same image from wavefront and fused (fused: scenes that fit LDS); nee is another estimator
            const std::string v = val();
            if (v == "auto") o.pipeline = PT_PIPELINE_AUTO;  // the library's choice: fused where the scene lives in LDS, else wavefront
            else if (v == "wavefront") o.pipeline = PT_PIPELINE_WAVEFRONT;
            else if (v == "fused") o.pipeline = PT_PIPELINE_FUSED;
            else if (v == "nee") o.pipeline = PT_PIPELINE_WAVEFRONT_NEE;
            else die("unknown pipeline " + v + " (auto, wavefront, fused, nee)");
        }
        else if (a == "--ppm") o.ppm = val();
        else if (a == "--pfm") o.pfm = val();
        else die("unknown option " + a);
    }
    if (o.ranks > 1) {
        if (o.devices.empty()) for (uint32_t r = 0; r < o.ranks; r++) o.devices.push_back((int)r);
        if (o.devices.size() != o.ranks) die("--devices needs one ordinal per rank");
    }

    char err[512] = { 0 };
    pth_scene hs{};
    const auto t0 = std::chrono::steady_clock::now();
    if (pth_load_obj(o.obj.c_str(), nullptr, &hs, err, sizeof(err)) != 0) die(err);
    const auto t1 = std::chrono::steady_clock::now();

    std::vector<RankResult> res(o.ranks);
    std::vector<float> image_f32;
    std::vector<uint8_t> image_bgra8;
    const bool want_f32 = !o.pfm.empty() || (o.ranks > 1 && !o.ppm.empty());
    const auto t2 = std::chrono::steady_clock::now();
    if (o.ranks == 1) {
        run_rank(o, hs, 0, nullptr, res[0], want_f32 ? &image_f32 : nullptr, !o.ppm.empty() ? &image_bgra8 : nullptr, nullptr);
    } else {
        // one host thread per GPU (north star: host code stays C++; the reference has one device, main.cpp:105)
        pt_unique_id id{};
        if (pt_comm_unique_id(&id) != PT_OK) die("RCCL is not available (pt_comm_unique_id)");
        std::vector<std::thread> th;
        Agreement agree(o.ranks);
        for (uint32_t r = 0; r < o.ranks; r++)
            th.emplace_back([&, r] { run_rank(o, hs, r, &id, res[r], r == 0 && want_f32 ? &image_f32 : nullptr, nullptr, &agree); });
        for (auto &t : th) t.join();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    for (int pass = 0; pass < 2; pass++)   // the rank that failed first, not the peers that stopped because of it
        for (uint32_t r = 0; r < o.ranks; r++)
            if (!res[r].error.empty() && (pass == 1 || res[r].error.rfind("stopped:", 0) != 0)) die("rank " + std::to_string(r) + ": " + res[r].error);

    if (!o.ppm.empty()) {
        if (o.ranks > 1) {  // display transform of the presented float image: clamp + unorm8 (one frame's worth of main.cpp:481-484)
            image_bgra8.resize(4 * (size_t)o.width * o.height);
            for (size_t i = 0; i < (size_t)o.width * o.height; i++) {
                for (int c = 0; c < 3; c++) {
                    float v = image_f32[3 * i + (size_t)c];
                    v = !(v > 0.f) ? 0.f : (v > 1.f ? 1.f : v);
                    image_bgra8[4 * i + (size_t)(2 - c)] = (uint8_t)(v * 255.0f + 0.5f);
                }
                image_bgra8[4 * i + 3] = 255;
            }
        }
        if (pth_write_ppm_bgra8(o.ppm.c_str(), image_bgra8.data(), o.width, o.height) != 0) die("cannot write " + o.ppm);
    }
    if (!o.pfm.empty() && pth_write_pfm(o.pfm.c_str(), image_f32.data(), o.width, o.height) != 0) die("cannot write " + o.pfm);

    unsigned long long rays = 0, paths = 0, rays_min = ~0ull, rays_max = 0, rays_culled = 0;
    double render_ms = 0.0, present_ms = 0.0;
    for (const RankResult &r : res) {
        rays += r.st.rays; paths += r.st.paths; rays_culled += r.st.rays_culled;
        rays_min = std::min<unsigned long long>(rays_min, r.st.rays);
        rays_max = std::max<unsigned long long>(rays_max, r.st.rays);
        render_ms = std::max(render_ms, r.render_ms);
        present_ms = std::max(present_ms, r.present_ms);
    }
    const pt_stats &st = res[0].st;
    const pt_scene_info &info = res[0].info;
    const double load_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    const double ms = o.ranks > 1 ? render_ms + present_ms : (double)st.ms_total;
    std::printf("{\"obj\": \"%s\", \"triangles\": %u, \"bvh_nodes\": %u, \"bvh_height\": %u, \"load_ms\": %.3f, "
                "\"bvh_build_ms\": %.3f, \"width\": %u, \"height\": %u, \"frames\": %u, \"spp_per_frame\": %u, "
                "\"max_depth\": %u, \"ranks\": %u, \"rccl_ranks\": %u, \"rays\": %llu, \"paths\": %llu, "
                "\"rays_per_rank_min\": %llu, \"rays_per_rank_max\": %llu, \"rounds\": %u, \"ms_total\": %.3f, "
                "\"present_ms\": %.3f, \"wall_ms_all_ranks\": %.3f, \"ms_per_frame\": %.3f, \"mrays_per_s\": %.1f, \"mrays_per_s_walked\": %.1f, \"selftest_wrong_pixels\": %lld, "
                "\"pipeline\": %u, \"sample_groups\": %u, \"tail_samples\": %u, \"rays_culled\": %llu}\n",
                o.obj.c_str(), info.n_tris, info.n_nodes, info.bvh_height, load_ms, info.build_ms, o.width, o.height, o.frames, o.spp,
                o.depth, o.ranks, res[0].rccl_ranks, rays, paths, rays_min, rays_max, st.rounds, ms, present_ms, wall_ms,
                ms / o.frames, ms > 0 ? (double)rays / (ms * 1e3) : 0.0, ms > 0 ? (double)(rays - rays_culled) / (ms * 1e3) : 0.0, res[0].selftest_wrong,
                st.pipeline, st.sample_groups, st.tail_samples, rays_culled);  // (rays_culled: camera rays finished without a walk, counted in rays and in mrays_per_s; mrays_per_s_walked prices the same time without them)
    pth_free_scene(&hs);
    return 0;
}
