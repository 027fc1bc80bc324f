// present_rccl.hip -- assembling the presented image of a tile-sharded render on one rank.
//
// The reference renders on physical device 0 only (main.cpp:105) and copies its storage image to the swapchain
// (main.cpp:661-667).  Here N ranks (processes or host threads, one GPU each) render the interleaved 8x8 pixel tiles
// of the SAME image; what stands in for that copy is ONE collective per presented image:
//     pack    every rank gathers the tiles it owns into a dense buffer [tile][64 pixels][rgb]   (k_pack_tiles)
//     gather  ncclSend of that buffer to the root / ncclRecv x (N-1) on the root, one RCCL group: a packed
//             direct-to-root gather moves W*H*12/N bytes per rank (3.1 MB at 1080p, N = 8) over each rank's own
//             xGMI link to the root instead of a 24.9 MB zero-padded ring reduce per rank
//     unpack  the root scatters all ranks' tiles into the presented image                          (k_unpack_tiles)
// The result is written to a separate buffer: every rank's accumulation film stays what it was, so progressive
// rendering can go on and present again (x + 0 = x is not even needed: no arithmetic touches the radiance).
//
// RCCL is loaded with dlopen at the first pt_comm_* call: libpt_amd.so itself does not link it, single-GPU users
// never load it, and a box without RCCL gets PT_ERR_UNSUPPORTED instead of a loader error.
#include "pt_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

constexpr int TBP = 256;
constexpr uint32_t TILE_FLOATS = 64 * 3;

// the few RCCL entry points used, by their published C signatures (rccl.h): ncclResult_t is an int (0 = success),
// ncclComm_t an opaque pointer, ncclUniqueId 128 opaque bytes, ncclFloat32 = 7
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, pt_unique_id id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommCount)(void *comm, int *count) = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load()
{
    std::call_once(g_rccl_once, [] {
        Rccl &r = g_rccl;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("RCCL not found: ") + dlerror(); return; }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return g_rccl.lib && g_rccl.err.empty();
}

// tiles of rank r in the order ensure_work() enumerates them (row major over the tile grid)
std::vector<uint32_t> tiles_of(uint32_t w, uint32_t h, uint32_t rank, uint32_t world)
{
    std::vector<uint32_t> t;
    const uint32_t tiles_x = (w + 7) / 8, tiles_y = (h + 7) / 8;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++)
            if ((tx + ty) % world == rank) t.push_back(tx | (ty << 16));
    return t;
}

// one thread per float of the packed buffer: [tile][pixel 0..63][rgb]
__global__ __launch_bounds__(TBP) void k_pack_tiles(const float *__restrict__ film, uint32_t w, uint32_t h,
                                                    const uint32_t *__restrict__ tiles, uint32_t n_tiles,
                                                    float *__restrict__ packed)
{
    const uint32_t i = blockIdx.x * TBP + threadIdx.x;
    if (i >= n_tiles * TILE_FLOATS) return;
    const uint32_t t = i / TILE_FLOATS, r = i - t * TILE_FLOATS, px = r / 3u, c = r - 3u * px;
    const uint32_t g = tiles[t];
    const uint32_t x = (g & 0xFFFFu) * 8u + (px & 7u), y = (g >> 16) * 8u + (px >> 3);
    packed[i] = (x < w && y < h) ? film[3 * ((size_t)y * w + x) + c] : 0.f;
}

__global__ __launch_bounds__(TBP) void k_unpack_tiles(const float *__restrict__ packed, uint32_t w, uint32_t h,
                                                      const uint32_t *__restrict__ tiles, uint32_t n_tiles,
                                                      float *__restrict__ image)
{
    const uint32_t i = blockIdx.x * TBP + threadIdx.x;
    if (i >= n_tiles * TILE_FLOATS) return;
    const uint32_t t = i / TILE_FLOATS, r = i - t * TILE_FLOATS, px = r / 3u, c = r - 3u * px;
    const uint32_t g = tiles[t];
    const uint32_t x = (g & 0xFFFFu) * 8u + (px & 7u), y = (g >> 16) * 8u + (px >> 3);
    if (x < w && y < h) image[3 * ((size_t)y * w + x) + c] = packed[i];
}

// device copy of a rank's tile list, cached per (film geometry, rank, world) on the communicator / call
struct TileList {
    uint32_t *d = nullptr;
    uint32_t n = 0;
};
pt_status upload_tiles(pt_ctx *ctx, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, TileList &out)
{
    const std::vector<uint32_t> t = tiles_of(w, h, rank, world);
    out.n = (uint32_t)t.size();
    out.d = nullptr;
    PT_HIP(ctx, hipMalloc((void **)&out.d, sizeof(uint32_t) * std::max<size_t>(t.size(), 1)));
    if (!t.empty()) PT_HIP(ctx, hipMemcpy(out.d, t.data(), sizeof(uint32_t) * t.size(), hipMemcpyHostToDevice));
    return PT_OK;
}

}  // namespace

struct pt_comm {
    pt_ctx *ctx = nullptr;
    void *nccl = nullptr;
    uint32_t world = 1, rank = 0;
    // per (film geometry, root): tile lists of every rank, the packed send buffer and the root's receive buffer
    uint32_t w = 0, h = 0, root = 0;
    std::vector<TileList> tiles;   // [world]
    float *d_send = nullptr;
    float *d_recv = nullptr;       // root: concatenation of all ranks' packed tiles
    std::vector<size_t> recv_off;  // [world + 1] in floats
};

static void comm_free_geometry(pt_comm *c)
{
    for (TileList &t : c->tiles) (void)hipFree(t.d);
    c->tiles.clear();
    (void)hipFree(c->d_send);
    (void)hipFree(c->d_recv);
    c->d_send = c->d_recv = nullptr;
    c->recv_off.clear();
    c->w = c->h = 0;
}

extern "C" {

pt_status pt_film_tile_count(const pt_film *f, uint32_t rank, uint32_t world, uint32_t *n_tiles)
{
    if (!f || !n_tiles || world == 0 || rank >= world) return PT_ERR_INVALID_ARG;
    *n_tiles = (uint32_t)tiles_of(f->w, f->h, rank, world).size();
    return PT_OK;
}

pt_status pt_film_pack_tiles(pt_film *f, uint32_t rank, uint32_t world, float *d_packed)
{
    if (!f) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    if (!d_packed || world == 0 || rank >= world) { ctx->err = "pt_film_pack_tiles: bad argument"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    TileList tl;
    pt_status rc = upload_tiles(ctx, f->w, f->h, rank, world, tl);
    if (rc != PT_OK) return rc;
    const uint32_t n = tl.n * TILE_FLOATS;
    if (n) k_pack_tiles<<<(n + TBP - 1) / TBP, TBP, 0, ctx->stream>>>(f->d_rgb, f->w, f->h, tl.d, tl.n, d_packed);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tl.d);
    PT_HIP(ctx, e);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

pt_status pt_film_unpack_tiles(pt_film *f, uint32_t rank, uint32_t world, const float *d_packed, float *d_image)
{
    if (!f) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    if (!d_packed || !d_image || world == 0 || rank >= world) { ctx->err = "pt_film_unpack_tiles: bad argument"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    TileList tl;
    pt_status rc = upload_tiles(ctx, f->w, f->h, rank, world, tl);
    if (rc != PT_OK) return rc;
    const uint32_t n = tl.n * TILE_FLOATS;
    if (n) k_unpack_tiles<<<(n + TBP - 1) / TBP, TBP, 0, ctx->stream>>>(d_packed, f->w, f->h, tl.d, tl.n, d_image);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tl.d);
    PT_HIP(ctx, e);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

pt_status pt_comm_unique_id(pt_unique_id *id)
{
    if (!id) return PT_ERR_INVALID_ARG;
    if (!rccl_load()) return PT_ERR_UNSUPPORTED;
    return g_rccl.GetUniqueId(id) == 0 ? PT_OK : PT_ERR_HIP;
}

pt_status pt_comm_create(pt_ctx *ctx, const pt_unique_id *id, uint32_t world, uint32_t rank, pt_comm **out)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!out || !id || world == 0 || rank >= world) { ctx->err = "pt_comm_create: bad argument"; return PT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!rccl_load()) { ctx->err = g_rccl.err.empty() ? "RCCL unavailable" : g_rccl.err; return PT_ERR_UNSUPPORTED; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_comm *c = new (std::nothrow) pt_comm();
    if (!c) return PT_ERR_OOM;
    c->ctx = ctx; c->world = world; c->rank = rank;
    const int r = g_rccl.CommInitRank(&c->nccl, (int)world, *id, (int)rank);
    if (r != 0) {
        ctx->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
        delete c;
        return PT_ERR_HIP;
    }
    *out = c;
    return PT_OK;
}

pt_status pt_comm_ranks(const pt_comm *c, uint32_t *n)
{
    if (!c || !n) return PT_ERR_INVALID_ARG;
    int cnt = 0;
    if (g_rccl.CommCount(c->nccl, &cnt) != 0) return PT_ERR_HIP;
    *n = (uint32_t)cnt;
    return PT_OK;
}

void pt_comm_destroy(pt_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    comm_free_geometry(c);
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    delete c;
}

// Every rank of the communicator calls this once per presented image, after its pt_render calls, with the film it
// rendered as (rank, world) of the communicator.  d_image: device memory for width*height*3 floats on the root
// (ignored elsewhere).  Blocking, like the copy + present it replaces (main.cpp:661-683).
pt_status pt_film_present(pt_film *f, pt_comm *c, uint32_t root, float *d_image)
{
    if (!f || !c) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    if (ctx != c->ctx) { ctx->err = "pt_film_present: film and communicator belong to different contexts"; return PT_ERR_INVALID_ARG; }
    if (root >= c->world || (c->rank == root && !d_image)) { ctx->err = "pt_film_present: bad root / null image on the root"; return PT_ERR_INVALID_ARG; }
    if (f->work.d_tiles && (f->work.rank != c->rank || f->work.world != c->world)) {
        ctx->err = "pt_film_present: the film was rendered as another (rank, world) than the communicator's";
        return PT_ERR_INVALID_ARG;
    }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (c->w != f->w || c->h != f->h || c->root != root) {  // first presentation of this (geometry, root): tile lists and staging buffers
        comm_free_geometry(c);
        c->tiles.resize(c->world);
        c->recv_off.assign(c->world + 1, 0);
        for (uint32_t r = 0; r < c->world; r++) {
            const bool need = r == c->rank || c->rank == root;  // the root unpacks everyone's tiles
            if (need) {
                const pt_status rc = upload_tiles(ctx, f->w, f->h, r, c->world, c->tiles[r]);
                if (rc != PT_OK) return rc;
            } else {
                c->tiles[r].n = (uint32_t)tiles_of(f->w, f->h, r, c->world).size();
            }
            c->recv_off[r + 1] = c->recv_off[r] + (size_t)c->tiles[r].n * TILE_FLOATS;
        }
        PT_HIP(ctx, hipMalloc((void **)&c->d_send, sizeof(float) * std::max<size_t>((size_t)c->tiles[c->rank].n * TILE_FLOATS, 1)));
        if (c->rank == root) PT_HIP(ctx, hipMalloc((void **)&c->d_recv, sizeof(float) * std::max<size_t>(c->recv_off[c->world], 1)));
        c->w = f->w; c->h = f->h; c->root = root;
    }
    const TileList &mine = c->tiles[c->rank];
    const uint32_t n_mine = mine.n * TILE_FLOATS;
    float *const send = c->rank == root ? c->d_recv + c->recv_off[root] : c->d_send;  // the root packs in place
    if (n_mine) k_pack_tiles<<<(n_mine + TBP - 1) / TBP, TBP, 0, st>>>(f->d_rgb, f->w, f->h, mine.d, mine.n, send);
    if (c->world > 1) {
        int r = g_rccl.GroupStart();
        if (c->rank == root) {
            for (uint32_t p = 0; p < c->world && r == 0; p++)
                if (p != root && c->tiles[p].n)
                    r = g_rccl.Recv(c->d_recv + c->recv_off[p], (size_t)c->tiles[p].n * TILE_FLOATS, 7 /* ncclFloat32 */, (int)p, c->nccl, st);
        } else if (n_mine) {
            r = g_rccl.Send(send, n_mine, 7, (int)root, c->nccl, st);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r != 0 || r2 != 0) {
            ctx->err = std::string("RCCL gather of the packed tiles: ") + g_rccl.GetErrorString(r != 0 ? r : r2);
            return PT_ERR_HIP;
        }
    }
    if (c->rank == root)
        for (uint32_t p = 0; p < c->world; p++) {
            const uint32_t n = c->tiles[p].n * TILE_FLOATS;
            if (n) k_unpack_tiles<<<(n + TBP - 1) / TBP, TBP, 0, st>>>(c->d_recv + c->recv_off[p], f->w, f->h, c->tiles[p].d, c->tiles[p].n, d_image);
        }
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

}  // extern "C"
