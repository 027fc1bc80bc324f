// film_work.hip -- the film's wavefront workspace: how many path slots a render gets (frames in flight x sample groups x pixels,
// RenderShape) and the device buffers behind them (queues, hit records, radiance accumulators or term logs).
// Buffers only ever grow; a grow that does not fit leaves the film usable (PT_ERR_OOM, AUTO shapes are planned again smaller).
#include "wavefront_host.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace {
using ptw::SPILL_POOL_ENTRIES;

// Bytes of workspace per path slot that do not depend on the sample-group shape: two queue sets
// (id 8 + state 16 + rayA 16 + rayB 8), hit 16 + instance 4, term count 4, pool head 4.
// (PT_PIPELINE_FUSED has no queues: 8 B per slot)
constexpr size_t SLOT_BYTES_QUEUES = 2 * (8 + 16 + 16 + 8) + 16 + 4, SLOT_BYTES_META = 4 + 4;
constexpr size_t SLOT_BYTES = SLOT_BYTES_QUEUES + SLOT_BYTES_META;

struct WorkNeed { size_t slots, color, terms, terms_over, total; };
// (tail > 0: n_slots head slots with an accumulator each + n_slots * tail one-sample tail slots with the logs; `slots` then counts the tail slots,
// which are what the per-slot meta arrays and the logs hold)
WorkNeed work_need(uint64_t n_slots, uint32_t groups, uint32_t term_cap, uint32_t term_pcap, bool queues = true, uint32_t tail = 0)
{
    WorkNeed n{};
    const size_t heads = (size_t)std::max<uint64_t>(n_slots, 1);
    n.slots = tail ? heads * tail : heads;
    n.color = groups == 1 ? heads : 0;
    const bool logs = groups > 1 || tail;
    n.terms = logs ? n.slots * (size_t)term_pcap : 0;
    n.terms_over = logs ? n.slots * (size_t)(term_cap - term_pcap) : 0;
    n.total = n.slots * (queues ? SLOT_BYTES : SLOT_BYTES_META) + sizeof(float4) * (n.color + n.terms + n.terms_over) +
              (logs ? sizeof(float4) * (size_t)SPILL_POOL_ENTRIES : 0);
    return n;
}

// One workspace allocation.  Out of memory (the device's, or the context's PT_MEM_BUDGET_MB) is PT_ERR_OOM, and HIP's
// sticky error is cleared so that the context stays usable.
pt_status work_alloc(pt_ctx *ctx, pt_film::Work &w, void **p, size_t bytes, size_t limit)
{
    *p = nullptr;
    if (limit && w.bytes + bytes > limit) {
        ctx->err = "wavefront workspace exceeds the memory budget (" + std::to_string((w.bytes + bytes) >> 20) + " MB wanted, " +
                   std::to_string(limit >> 20) + " MB allowed): fewer frames_in_flight / sample_groups fit";
        return PT_ERR_OOM;
    }
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        ctx->err = std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MB of wavefront workspace: " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP;
    }
    w.bytes += bytes;
    return PT_OK;
}

// Frees the shape-dependent buffers (everything but the tile list and the counters) and zeroes their capacities:
// the state after a failed grow -- the film itself (d_rgb / d_bgra) is untouched and the next render re-allocates.
void free_shape_buffers(pt_film::Work &w)
{
    for (int i = 0; i < 2; i++) {
        (void)hipFree(w.d_qid[i]); (void)hipFree(w.d_qstate[i]); (void)hipFree(w.d_qrayA[i]); (void)hipFree(w.d_qrayB[i]);
        w.d_qid[i] = nullptr; w.d_qstate[i] = w.d_qrayA[i] = nullptr; w.d_qrayB[i] = nullptr;
    }
    (void)hipFree(w.d_hit); (void)hipFree(w.d_hit_inst); (void)hipFree(w.d_nterm); (void)hipFree(w.d_spill_head);
    (void)hipFree(w.d_color); (void)hipFree(w.d_terms); (void)hipFree(w.d_terms_over); (void)hipFree(w.d_spill);
    w.d_hit = nullptr; w.d_hit_inst = nullptr; w.d_nterm = nullptr; w.d_spill_head = nullptr;
    w.d_color = nullptr; w.d_terms = nullptr; w.d_terms_over = nullptr; w.d_spill = nullptr;
    w.cap_slots = w.cap_meta = w.cap_color = w.cap_terms = w.cap_terms_over = 0;
    w.bytes = w.sort_bytes;  // (the ray-sort scratch is not a shape buffer)
}

}  // namespace

// Workspace for (rank, world) tiles, `lanes` frames in flight and `groups` sample groups.  Buffers only
// ever grow: a later call with a smaller shape reuses them (hipMalloc of tens of GB costs 100s of ms).
// A grow that does not fit returns PT_ERR_OOM and leaves the film WITHOUT shape buffers (all freed, capacities 0).
pt_status ptw_ensure_work(pt_film *f, uint32_t rank, uint32_t world, uint32_t lanes, uint32_t groups, uint32_t term_cap,
                          uint32_t term_pcap, bool queues, uint32_t tail)
{
    // queues = false (PT_PIPELINE_FUSED): no path queues and no hit records, only the per-slot radiance arrays
    pt_ctx *ctx = f->ctx;
    pt_film::Work &w = f->work;
    // The order of a rank's tiles is the order slots are numbered in, i.e. the order work is handed out.  The wavefront
    // pipelines take them row by row (queues = image order).  The fused pipeline hands slots out through a counter, and whatever is
    // handed out last runs alone at the end of the launch: its tiles go CENTRE FIRST, BORDER LAST (longest-processing-time-first
    // on the cheapest guess there is -- a camera looks at its subject: in the reference's view the outer ring of the image misses
    // the box after one ray, 32 rays per slot against ~130 inside), which cuts the ~3 ms drain of a launch to the length of a
    // border slot.  Results cannot depend on it (slot -> pixel goes through this table everywhere).
    // (one sample group only: with many short slots per pixel the whole chip reaches the cheap border ring at the same time and
    // every wave wants a new batch of slots every few microseconds -- more than the one counter word takes: one blocking 1080p
    // frame of 16 groups went from 6.8 to 10.9 ms; short slots have no drain worth ordering for anyway)
    const uint32_t order = (!queues && groups == 1u) ? 1u : 0u;
    if (!w.d_tiles || w.rank != rank || w.world != world || w.tile_order != order) {
        (void)hipFree(w.d_tiles);
        w.d_tiles = nullptr;
        const uint32_t tiles_x = (f->w + 7) / 8, tiles_y = (f->h + 7) / 8;
        std::vector<uint32_t> tiles;
        for (uint32_t ty = 0; ty < tiles_y; ty++)
            for (uint32_t tx = 0; tx < tiles_x; tx++)
                if ((tx + ty) % world == rank) tiles.push_back(tx | (ty << 16));
        if (order == 1u) {
            auto ring = [&](uint32_t t) {  // Chebyshev distance from the image centre in units of the half extent, 0 .. 1
                const float cx = 0.5f * (float)tiles_x, cy = 0.5f * (float)tiles_y;
                const float dx = std::fabs(((float)(t & 0xFFFFu) + 0.5f) - cx) / cx, dy = std::fabs(((float)(t >> 16) + 0.5f) - cy) / cy;
                return std::max(dx, dy);
            };
            std::stable_sort(tiles.begin(), tiles.end(), [&](uint32_t a, uint32_t b) { return ring(a) < ring(b); });
        }
        w.tile_order = order;
        w.rank = rank; w.world = world;
        w.n_tiles = (uint32_t)tiles.size();
        w.tile_rect[0] = w.tile_rect[1] = 0; w.tile_rect[2] = w.tile_rect[3] = -1;
        if (order == 1u) w.h_tiles = tiles; else w.h_tiles.clear();
        PT_HIP(ctx, hipMalloc((void **)&w.d_tiles, sizeof(uint32_t) * std::max<size_t>(tiles.size(), 1)));
        if (!tiles.empty())
            PT_HIP(ctx, hipMemcpy(w.d_tiles, tiles.data(), sizeof(uint32_t) * tiles.size(), hipMemcpyHostToDevice));
    }
    const uint64_t n_slots64 = (uint64_t)lanes * groups * w.n_tiles * 64ull;
    if (n_slots64 >= (1ull << 31)) {
        ctx->err = "too many path slots (frames_in_flight x sample_groups x pixels >= 2^31)";
        return PT_ERR_INVALID_ARG;
    }
    if (!w.d_count) PT_HIP(ctx, hipMalloc((void **)&w.d_count, sizeof(uint32_t) * PTW_COUNT_WORDS));  // queue sizes, 2 per pipeline | the fused kernel's slot counters
    if (tail && (n_slots64 * tail >= (1ull << 31) || queues || groups != 1)) {
        ctx->err = "head + tail slots: too many tail slots, or not the fused pipeline's one-group shape";
        return PT_ERR_INVALID_ARG;
    }
    const WorkNeed need = work_need(n_slots64, groups, term_cap, term_pcap, queues, tail);
    const size_t ns = need.slots;
    const size_t limit = ctx->mem_budget;
    pt_status rc = PT_OK;
#define PT_WORK_ALLOC(PTR, BYTES) \
    if (rc == PT_OK) rc = work_alloc(ctx, w, (void **)&(PTR), (BYTES), limit)
    if (queues && ns > w.cap_slots) {
        // the queue set goes as a whole: free first (peak = the new size, not old + new)
        for (int i = 0; i < 2; i++) {
            (void)hipFree(w.d_qid[i]); (void)hipFree(w.d_qstate[i]); (void)hipFree(w.d_qrayA[i]); (void)hipFree(w.d_qrayB[i]);
            w.d_qid[i] = nullptr; w.d_qstate[i] = w.d_qrayA[i] = nullptr; w.d_qrayB[i] = nullptr;
        }
        (void)hipFree(w.d_hit); (void)hipFree(w.d_hit_inst);
        w.d_hit = nullptr; w.d_hit_inst = nullptr;
        w.bytes -= w.cap_slots * SLOT_BYTES_QUEUES;
        w.cap_slots = 0;
        for (int i = 0; i < 2; i++) {
            PT_WORK_ALLOC(w.d_qid[i], sizeof(uint2) * ns);
            PT_WORK_ALLOC(w.d_qstate[i], sizeof(float4) * ns);
            PT_WORK_ALLOC(w.d_qrayA[i], sizeof(float4) * ns);
            PT_WORK_ALLOC(w.d_qrayB[i], sizeof(float2) * ns);
        }
        PT_WORK_ALLOC(w.d_hit, sizeof(float4) * ns);
        PT_WORK_ALLOC(w.d_hit_inst, sizeof(uint32_t) * ns);
        if (rc == PT_OK) w.cap_slots = ns;
    }
    if (rc == PT_OK && ns > w.cap_meta) {
        (void)hipFree(w.d_nterm); (void)hipFree(w.d_spill_head);
        w.d_nterm = nullptr; w.d_spill_head = nullptr;
        w.bytes -= w.cap_meta * SLOT_BYTES_META;
        w.cap_meta = 0;
        PT_WORK_ALLOC(w.d_nterm, sizeof(uint32_t) * ns);
        PT_WORK_ALLOC(w.d_spill_head, sizeof(uint32_t) * ns);
        if (rc == PT_OK) w.cap_meta = ns;
    }
    if (rc == PT_OK && need.color > w.cap_color) {
        (void)hipFree(w.d_color);
        w.bytes -= sizeof(float4) * w.cap_color;
        w.d_color = nullptr; w.cap_color = 0;
        PT_WORK_ALLOC(w.d_color, sizeof(float4) * need.color);
        if (rc == PT_OK) w.cap_color = need.color;
    }
    // primary log: term_pcap entries per slot (dense, what is normally touched); overflow: the rest of the
    // worst case (one entry per ray), allocated but rarely touched
    if (rc == PT_OK && need.terms > w.cap_terms) {
        (void)hipFree(w.d_terms);
        w.bytes -= sizeof(float4) * w.cap_terms;
        w.d_terms = nullptr; w.cap_terms = 0;
        PT_WORK_ALLOC(w.d_terms, sizeof(float4) * need.terms);
        if (rc == PT_OK) w.cap_terms = need.terms;
    }
    if (rc == PT_OK && need.terms_over > w.cap_terms_over) {
        (void)hipFree(w.d_terms_over);
        w.bytes -= sizeof(float4) * w.cap_terms_over;
        w.d_terms_over = nullptr; w.cap_terms_over = 0;
        PT_WORK_ALLOC(w.d_terms_over, sizeof(float4) * need.terms_over);
        if (rc == PT_OK) w.cap_terms_over = need.terms_over;
    }
    if (rc == PT_OK && (groups > 1 || tail) && !w.d_spill) PT_WORK_ALLOC(w.d_spill, sizeof(float4) * (size_t)SPILL_POOL_ENTRIES);
#undef PT_WORK_ALLOC
    if (rc != PT_OK) {
        free_shape_buffers(w);
        w.lanes = w.groups = w.term_cap = 0;
        w.n_slots = 0;
        return rc;
    }
    w.lanes = lanes; w.groups = groups; w.term_cap = term_cap;
    w.n_slots = (uint32_t)n_slots64;
    w.tail = tail;
    return PT_OK;
}


// frames in flight x sample groups: enough live paths (~32M) to fill the chip several times over, and slots that
// do not live longer than they have to
// `shrink`: 0 for the first try; pt_render retries with 1, 2, ... after an out-of-memory workspace grow, each step
// halving the memory the AUTO shape may plan for (explicit frames_in_flight / sample_groups are never overridden).
// `launch_class`: 0 instanced scenes, 1 scenes walked out of L2 / MALL / HBM (no LDS copy), 2 single-level scenes in LDS.  Class 1: the launches take milliseconds per million
// rays and what they gain from being LONG is measured: 1 M-triangle soup, 4 frames of 16 spp -- 4 groups (3.7 M rays per launch)
// 2 617 Mrays/s, 8 groups 2 799, 16 groups (14.8 M) 2 889; 16 frames x 4 groups 2 886, x 8 (29.6 M) 2 926; the 8 M-triangle soup
// at 2 frames +3 % from 8 to 16 groups (profiles/r03au_shapes_c5_c4.log).  So the sample groups of such scenes aim at 128 M live
// paths; the Cornell-class scenes followed in the round's last session (below), instanced scenes aim at 32 M.
RenderShape ptw_choose_shape(const pt_film *f, const pt_params *p, int launch_class, int shrink)
{
    RenderShape sh;
    const uint64_t pixels_local = ((uint64_t)((f->w + 7) / 8) * ((f->h + 7) / 8) * 64ull + p->world - 1) / p->world;
    const uint64_t target = 32ull << 20;  // 128 B of queue state per live path
    uint32_t lanes = p->frames_in_flight;
    if (lanes == 0) {
        // up to 32 frames / 64 M paths in flight, in EQUAL batches: 20 frames run as 1 x 20 (measured 22.2 Grays/s on
        // the Cornell box) rather than 16 + 4 (21.3), 40 as 2 x 20; every batch pays the same ~256 rounds of
        // per-launch fixed cost (~27 us per round and pipeline), so fewer and fuller batches are better
        const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(32, 2 * target / std::max<uint64_t>(pixels_local, 1)));
        const uint64_t batches = ((uint64_t)p->frame_count + cap - 1) / cap;
        lanes = (uint32_t)(((uint64_t)p->frame_count + batches - 1) / batches);
    }
    lanes = std::max(1u, std::min(lanes, p->frame_count));
    // A blocking render can check a batch and redo it; PT_FLAG_ASYNC can not, and keeps the worst-case log.
    const bool can_redo = (p->flags & PT_FLAG_ASYNC) == 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    // what the workspace may occupy: the device's free memory plus what this film already holds, within the
    // context's budget (PT_MEM_BUDGET_MB), 7/8 of it planned for
    uint64_t avail = (uint64_t)free_b + f->work.bytes;
    if (f->ctx->mem_budget) avail = std::min<uint64_t>(avail, f->ctx->mem_budget);
    avail = (avail - avail / 8) >> std::min(shrink, 40);
    const uint64_t have_log = f->work.cap_terms_over * sizeof(float4);  // already ours: counts as free
    if (f->ctx->mem_budget) free_b = (size_t)std::min<uint64_t>(free_b, avail);
    // sample groups: split each pixel's samples over several slots; the term logs keep the sum order exact.
    //  (a) few frames asked for: the frames in flight alone cannot fill the chip;
    //  (b) a slot lives group_size x depth rounds and every round costs ~27 us of launch-bound time per pipeline
    //      whatever its queue holds: 16 frames x 4 groups need 64 rounds instead of 256 (Cornell box, 1080p: +4 %),
    //      as long as the slots (<= 160 M: 21 GB of queues + 25 GB of primary log; 288 M for single-level scenes in LDS) allow it.
    uint32_t groups = p->sample_groups;
    if (groups == 0) {
        groups = 1;
        const uint64_t have = std::max<uint64_t>((uint64_t)lanes * pixels_local, 1);
        // live paths aimed at: 128 M for scenes walked out of HBM (above), 32 M for instanced scenes (C4, K = 8: 4 groups 12.74
        // Grays/s, 8 groups 12.52), and -- round 3, last session; 32 M until then -- 256 M for single-level scenes in LDS: a
        // Cornell-class render is better off with MORE slots and fewer rounds.  K = 2 (config C2 exactly): 8 groups 23.7 Grays/s,
        // 16: 25.4, 32: 26.5; K = 1: 16 groups 23.0, 32: 25.3; K = 4: 4 groups 23.7, 16: 26.3, 32 (266 M slots): 27.1; 16 frames as
        // two batches of 8 with 16 groups: 27.1; K = 16: 4 groups (133 M slots, 51 GB) 23.0 / 26.6 / 26.3 in three
        // processes, 8 groups (266 M, 70 GB) 26.5 / 27.1 / 26.5, 16 groups (109 GB) 26.3 / 27.3 / 27.0
        // (profiles/r03br_*, r03bs_*, r03bt_*, r03bu_*)
        // (instanced scenes, on the round's final kernels: C4 at K = 8 with 2 / 4 / 8 / 16 groups 13.40 / 13.77 / 13.88 / 13.70 Grays/s,
        // profiles/r03cs_c4_shapes_final.log -- so they aim at 128 M now as well; the 32 M of the comment above was measured before)
        double want = (double)((launch_class == 2 ? 8 : 4) * target) / (double)have;
        const uint64_t slot_budget = (launch_class == 2 ? 288ull : 160ull) << 20;
        if (can_redo && have * 4 <= slot_budget) want = std::max(want, 4.0);
        else if (can_redo && have * 2 <= slot_budget) want = std::max(want, 2.0);
        if (want >= 2.0) {
            // even groups only (uneven tails measured 8 % slower): the divisor of spp closest to `want`
            uint32_t g = 1;
            double best = 1e30;
            for (uint32_t d = 1; d <= p->spp_per_frame; d++) {
                if (p->spp_per_frame % d) continue;
                const double r = d > want ? d / want : want / d;
                if (r < best) { best = r; g = d; }
            }
            // without the redo the worst-case log (one 16-B term per ray) is allocated in full, so it has to fit:
            // at most 80 GB of the 288 and half of what is free right now
            const uint64_t log_bytes = (uint64_t)lanes * pixels_local * p->spp_per_frame * p->max_depth * 16ull;
            if (g > 1 && (can_redo || (log_bytes <= (80ull << 30) && log_bytes <= have_log + free_b / 2))) groups = g;
        }
    }
    groups = std::max(1u, std::min(groups, p->spp_per_frame));
    // AUTO shapes have to fit the memory there is (queues + primary log; the overflow log is budgeted below): first
    // fewer sample groups (the next smaller divisor of spp), then fewer frames in flight
    auto planned = [&](uint32_t l, uint32_t g) {
        const uint32_t gs = (p->spp_per_frame + g - 1) / g;
        return work_need((uint64_t)l * g * pixels_local, g, g > 1 ? std::min(gs * p->max_depth, gs + 2u) : 0u,
                         g > 1 ? std::min(gs * p->max_depth, gs + 2u) : 0u, launch_class != 3).total;
    };
    while (planned(lanes, groups) > avail) {
        if (p->sample_groups == 0 && groups > 1) {
            uint32_t g = groups - 1;
            while (g > 1 && p->spp_per_frame % g) g--;
            groups = g;
        } else if (p->frames_in_flight == 0 && lanes > 1) {
            lanes = (lanes + 1) / 2;
        } else {
            break;  // explicit shape (or one frame, one group): ensure_work reports PT_ERR_OOM if it does not fit
        }
    }
    sh.group_size = (p->spp_per_frame + groups - 1) / groups;
    sh.groups = (p->spp_per_frame + sh.group_size - 1) / sh.group_size;  // no empty groups
    const uint32_t worst = sh.groups > 1 ? sh.group_size * p->max_depth : 0u;  // every ray of a slot adds a term
    sh.term_cap = worst;
    sh.term_pcap = std::min(worst, sh.group_size + 2u);  // ~1 term per sample is typical (the miss that ends it)
    if (sh.groups > 1 && can_redo) {
        // overflow log within a budget (2 GB -- 16 GB until round 4: the Cornell frame renders as fast with none at all and never
        // overflows the shared pool, profiles/r04h_overflow_log.txt -- and a quarter of the free memory) instead of the worst case
        // (136 GB for 16 frames x 4 groups at 1080p); a slot that fills it and the pool raises a flag and the batch is redone with groups == 1
        const uint64_t n_slots = (uint64_t)lanes * sh.groups * pixels_local;
        const uint64_t room = avail > planned(lanes, sh.groups) ? avail - planned(lanes, sh.groups) : 0;
        // (the fused pipeline is there to run in a small workspace: 1 GB of overflow log; the shared pool and the redo cover the rest)
        const uint64_t budget = std::min<uint64_t>(std::min<uint64_t>(launch_class == 3 ? 1ull << 30 : 2ull << 30, (have_log + free_b) / 4), room);
        uint64_t ocap = std::min<uint64_t>(worst - sh.term_pcap, budget / std::max<uint64_t>(n_slots * sizeof(float4), 1));
        if (f->ctx->tune.term_ocap >= 0) ocap = std::min<uint64_t>(ocap, (uint64_t)f->ctx->tune.term_ocap);  // tests
        sh.term_cap = sh.term_pcap + (uint32_t)ocap;
        sh.bounded = sh.term_cap < worst;
    }
    sh.lanes = lanes;
    return sh;
}

// The shape of a render and its workspace.  An AUTO shape that does not fit after all (another allocator took the
// memory between hipMemGetInfo and hipMalloc) is planned again for half the memory, down to one frame and one group;
// an explicit shape that does not fit is PT_ERR_OOM.  Either way a failure leaves the film usable.
// The fused pipeline's hand-out order, second key.  Centre first is a guess about where the subject is; the camera and the scene's box say where
// it is NOT: a pixel whose primary rays pass outside the box's projection costs one ray per sample (32 rays a slot against ~170 inside the
// Cornell box), and whatever is handed out last runs alone at the end of the launch.  So the tiles that touch the pixel rectangle `rect`
// {x0, y0, x1, y1} (render.hip: the projection of the scene's box) go first, centre first among themselves, and the others after them: the
// launch ends with the cheapest slots there are.  x1 < x0: no rectangle, the centre-first list as built.  Order only -- slot -> pixel goes through
// the table everywhere, results cannot depend on it.  Costs a pass over the host list and a 130 KB copy when the rectangle changes, else nothing.
pt_status ptw_tiles_subject_first(pt_film *f, const int32_t rect[4], hipStream_t st)
{
    pt_film::Work &w = f->work;
    pt_ctx *ctx = f->ctx;
    if (w.tile_order != 1u || w.h_tiles.empty() || !w.d_tiles) return PT_OK;
    const bool none = rect[2] < rect[0] || rect[3] < rect[1], had_none = w.tile_rect[2] < w.tile_rect[0] || w.tile_rect[3] < w.tile_rect[1];
    if ((none && had_none) || (!none && !had_none && std::equal(rect, rect + 4, w.tile_rect))) return PT_OK;
    std::vector<uint32_t> out;
    out.reserve(w.h_tiles.size());
    if (none) {
        out = w.h_tiles;
    } else {
        auto touches = [&](uint32_t t) {
            const int32_t x = (int32_t)(t & 0xFFFFu) * 8, y = (int32_t)(t >> 16) * 8;
            return x <= rect[2] && x + 7 >= rect[0] && y <= rect[3] && y + 7 >= rect[1];
        };
        for (uint32_t t : w.h_tiles) if (touches(t)) out.push_back(t);
        for (uint32_t t : w.h_tiles) if (!touches(t)) out.push_back(t);
    }
    PT_HIP(ctx, hipStreamSynchronize(st));  // (nothing of this film is in flight when a blocking render begins; kept for callers that come later)
    PT_HIP(ctx, hipMemcpy(w.d_tiles, out.data(), sizeof(uint32_t) * out.size(), hipMemcpyHostToDevice));
    std::copy(rect, rect + 4, w.tile_rect);
    return PT_OK;
}

pt_status ptw_shape_and_work(pt_film *f, const pt_params *p_in, RenderShape &sh, int launch_class, bool queues)
{
    pt_status rc = PT_OK;
    pt_params p_local = *p_in;
    if (p_local.pipeline == PT_PIPELINE_WAVEFRONT_NEE) p_local.sample_groups = 1;  // (up to two radiance terms per hit -- a camera ray's emitter hit, the light sample -- so a sample has more than group_size + 2: the plain accumulator)
    const pt_params *p = &p_local;
    for (int attempt = 0; attempt < 12; attempt++) {
        sh = ptw_choose_shape(f, p, launch_class, attempt);
        rc = ptw_ensure_work(f, p->rank, p->world, sh.lanes, sh.groups, sh.term_cap, sh.term_pcap, queues);
        if (rc != PT_ERR_OOM) return rc;
        const bool can_shrink = (p->frames_in_flight == 0 && sh.lanes > 1) || (p->sample_groups == 0 && sh.groups > 1);
        if (!can_shrink) return rc;
    }
    return rc;
}

ptw::RenderConst ptw_render_const(const pt_params *p, const pt_film::Work &w, const RenderShape &sh)
{
    ptw::RenderConst rc{};
    rc.cam = { p->cam_origin[0], p->cam_origin[1], p->cam_origin[2], p->cam_target[0], p->cam_target[1], p->cam_target[2],
               (float)p->width, (float)p->height,
               // (pt_math.h primary_target: the reciprocals of the launch size by the host's correctly rounded divide, for sizes the three-FMA
               // quotient is proven for)
               p->width <= (1u << 20) ? 1.0f / (float)p->width : 0.0f, p->height <= (1u << 20) ? 1.0f / (float)p->height : 0.0f };
    for (int k = 0; k < 3; k++) rc.env[k] = p->env[k];
    rc.tmin = p->tmin; rc.tmax = p->tmax;
    rc.width = p->width; rc.height = p->height; rc.tiles_x = (p->width + 7) / 8;
    rc.spp = p->spp_per_frame; rc.max_depth = p->max_depth;
    rc.slots_per_lane = w.n_tiles * 64u;
    rc.groups = sh.groups; rc.group_size = sh.group_size; rc.term_cap = sh.term_cap;
    rc.div_spl.init(std::max(rc.slots_per_lane, 1u)); rc.div_groups.init(std::max(sh.groups, 1u));
    rc.term_pcap = sh.term_pcap;
    rc.n_slots = w.n_slots;
    rc.tail = sh.tail;
    rc.head_samples = p->spp_per_frame - std::min(sh.tail, p->spp_per_frame);
    rc.n_head = sh.tail ? w.n_slots : 0u;
    rc.n_tail = sh.tail ? w.n_slots * sh.tail : 0u;
    rc.div_tail.init(std::max(sh.tail, 1u));
    return rc;
}

// pixels of this rank's 8x8 tiles that lie inside the image (samples started = that x spp x frames)
uint64_t ptw_valid_local_pixels(const pt_film *f, const pt_params *p)
{
    uint64_t valid = 0;
    const uint32_t tiles_x = (f->w + 7) / 8, tiles_y = (f->h + 7) / 8;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++)
            if ((tx + ty) % p->world == p->rank)
                valid += (uint64_t)std::min(8u, f->w - tx * 8) * std::min(8u, f->h - ty * 8);
    return valid;
}

// what pt_stats.workspace_bytes reports: everything the film's wavefront workspace holds + the context's stack-spill area
uint64_t ptw_workspace_bytes(const pt_film *f)
{
    const pt_film::Work &w = f->work;
    return (uint64_t)w.bytes + (uint64_t)w.cap_sq * (16 + 8 + 16 + 4 + 4 + 16) + (uint64_t)f->ctx->spill_bytes;
}


void ptw_free_work(pt_film *f)
{
    pt_film::Work &w = f->work;
    (void)hipFree(w.d_tiles);
    (void)hipFree(w.d_color);
    (void)hipFree(w.d_terms);
    (void)hipFree(w.d_terms_over);
    (void)hipFree(w.d_nterm);
    (void)hipFree(w.d_spill_head);
    (void)hipFree(w.d_spill);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(w.d_qid[i]);
        (void)hipFree(w.d_qstate[i]);
        (void)hipFree(w.d_qrayA[i]);
        (void)hipFree(w.d_qrayB[i]);
    }
    (void)hipFree(w.d_hit);
    (void)hipFree(w.d_hit_inst);
    (void)hipFree(w.d_count);
    (void)hipFree(w.d_sort);
    (void)hipFree(w.d_sq_rayA); (void)hipFree(w.d_sq_rayB); (void)hipFree(w.d_sq_contrib); (void)hipFree(w.d_sq_slot);
    (void)hipFree(w.d_sq_tmax); (void)hipFree(w.d_sq_hit); (void)hipFree(w.d_sq_count);
    w = pt_film::Work{};
}

