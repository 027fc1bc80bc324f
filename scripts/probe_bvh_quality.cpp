// Dev probe (CPU, no GPU): how much would a surface-area builder buy on the triangle soup of config C5 over the LBVH the
// device builds?  Builds three binary BVHs over the same soup (pth_make_soup) -- LBVH (30-bit Morton, highest differing
// bit), binned SAH top-down (16 bins), and the LBVH's topology re-rooted by nothing (control) -- and traces the same
// incoherent rays (origins on random triangles, uniform directions: what bounce rays look like) with an ordered
// closest-hit walk, counting inner-node visits and triangle tests per ray.  Lines fetched per ray scale with the visits.
//   g++ -O2 -std=c++20 -Iinclude scripts/probe_bvh_quality.cpp -Lsingle-file-vulkan-pathtracing_amd -lpt_host -Wl,-rpath,$PWD/single-file-vulkan-pathtracing_amd -o /tmp/probe_bvh
//   /tmp/probe_bvh [n_tris=1000000] [n_rays=200000]
#include "pt_host.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

struct V3 { double x, y, z; };
static V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct Box {
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    void grow(const Box &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float *p) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    double area() const { double d[3] = { (double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2] }; return d[0] < 0 ? 0 : 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
};
struct Node { Box box; int left = -1, right = -1, first = 0, count = 0; };  // leaf: count > 0
struct Scene { const float *v; uint32_t n; std::vector<Box> tb; std::vector<float> cen; };

static uint32_t expand10(uint32_t v) { v &= 1023; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v; }

static int build_lbvh(const Scene &s, std::vector<uint32_t> &order, const std::vector<uint32_t> &key, std::vector<Node> &nodes, int lo, int hi, int bit, int leaf)
{
    const int id = (int)nodes.size();
    nodes.emplace_back();
    if (hi - lo <= leaf || bit < 0) {
        if (hi - lo > leaf) {  // identical keys: median split
            const int mid = (lo + hi) / 2;
            const int l = build_lbvh(s, order, key, nodes, lo, mid, -1, leaf), r = build_lbvh(s, order, key, nodes, mid, hi, -1, leaf);
            nodes[id].left = l; nodes[id].right = r; nodes[id].box = nodes[l].box; nodes[id].box.grow(nodes[r].box);
            return id;
        }
        nodes[id].first = lo; nodes[id].count = hi - lo;
        for (int i = lo; i < hi; i++) nodes[id].box.grow(s.tb[order[i]]);
        return id;
    }
    // first element whose `bit` is set
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) / 2; if ((key[order[m]] >> bit) & 1u) b = m; else a = m + 1; }
    if (a == lo || a == hi) { nodes.pop_back(); return build_lbvh(s, order, key, nodes, lo, hi, bit - 1, leaf); }
    const int l = build_lbvh(s, order, key, nodes, lo, a, bit - 1, leaf), r = build_lbvh(s, order, key, nodes, a, hi, bit - 1, leaf);
    nodes[id].left = l; nodes[id].right = r; nodes[id].box = nodes[l].box; nodes[id].box.grow(nodes[r].box);
    return id;
}

static int build_sah(const Scene &s, std::vector<uint32_t> &order, std::vector<Node> &nodes, int lo, int hi, int leaf)
{
    const int id = (int)nodes.size();
    nodes.emplace_back();
    Box b, cb;
    for (int i = lo; i < hi; i++) { b.grow(s.tb[order[i]]); cb.grow(&s.cen[3 * order[i]]); }
    nodes[id].box = b;
    if (hi - lo <= leaf) { nodes[id].first = lo; nodes[id].count = hi - lo; return id; }
    constexpr int NB = 16;
    double best = 1e300; int best_axis = -1, best_bin = 0;
    for (int ax = 0; ax < 3; ax++) {
        const float c0 = cb.lo[ax], c1 = cb.hi[ax];
        if (!(c1 > c0)) continue;
        Box bb[NB]; int cnt[NB] = {};
        const float scale = NB / (c1 - c0);
        for (int i = lo; i < hi; i++) {
            int k = std::min(NB - 1, (int)((s.cen[3 * order[i] + ax] - c0) * scale));
            bb[k].grow(s.tb[order[i]]); cnt[k]++;
        }
        double ra[NB]; Box acc; int n = 0;
        for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k]); n += cnt[k]; ra[k] = n ? acc.area() * n : 0; }
        Box l; int nl = 0;
        for (int k = 0; k < NB - 1; k++) {
            l.grow(bb[k]); nl += cnt[k];
            if (!nl || nl == hi - lo) continue;
            const double c = l.area() * nl + ra[k + 1];
            if (c < best) { best = c; best_axis = ax; best_bin = k; }
        }
    }
    int mid;
    if (best_axis < 0) mid = (lo + hi) / 2;
    else {
        const float c0 = cb.lo[best_axis], scale = NB / (cb.hi[best_axis] - c0);
        mid = (int)(std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t t) { return std::min(NB - 1, (int)((s.cen[3 * t + best_axis] - c0) * scale)) <= best_bin; }) - order.begin());
        if (mid == lo || mid == hi) mid = (lo + hi) / 2;
    }
    const int l = build_sah(s, order, nodes, lo, mid, leaf), r = build_sah(s, order, nodes, mid, hi, leaf);
    nodes[id].left = l; nodes[id].right = r;
    return id;
}

static bool slab(const Box &b, V3 o, V3 inv, double tmax, double &tn)
{
    double t0 = 0, t1 = tmax;
    const double oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
    for (int k = 0; k < 3; k++) {
        double a = (b.lo[k] - oo[k]) * ii[k], c = (b.hi[k] - oo[k]) * ii[k];
        if (a > c) std::swap(a, c);
        t0 = std::max(t0, a); t1 = std::min(t1, c);
    }
    tn = t0;
    return t0 <= t1;
}

struct Counts { double nodes = 0, tris = 0, leaves = 0, hits = 0; };
static void trace(const Scene &s, const std::vector<Node> &nodes, const std::vector<uint32_t> &order, V3 o, V3 d, Counts &c)
{
    const V3 inv = { 1 / d.x, 1 / d.y, 1 / d.z };
    double best = 1e30; int stack[128], sp = 0; double st[128];
    stack[sp] = 0; st[sp++] = 0;
    while (sp) {
        const int id = stack[--sp];
        if (st[sp] >= best) continue;
        const Node &n = nodes[id];
        if (n.count) {
            c.leaves++;
            for (int i = n.first; i < n.first + n.count; i++) {
                c.tris++;
                const float *p = s.v + 9 * (size_t)order[i];
                const V3 v0 = { p[0], p[1], p[2] }, e1 = V3{ p[3], p[4], p[5] } - v0, e2 = V3{ p[6], p[7], p[8] } - v0;
                const V3 pv = cross(d, e2); const double det = dot(e1, pv);
                if (det == 0) continue;
                const V3 tv = o - v0; const double u = dot(tv, pv) / det; if (u < 0 || u > 1) continue;
                const V3 qv = cross(tv, e1); const double v = dot(d, qv) / det; if (v < 0 || u + v > 1) continue;
                const double t = dot(e2, qv) / det;
                if (t > 1e-3 && t < best) best = t;
            }
            continue;
        }
        c.nodes++;  // an inner node = one fetch of both children's boxes
        double tl, tr;
        const bool hl = slab(nodes[n.left].box, o, inv, best, tl), hr = slab(nodes[n.right].box, o, inv, best, tr);
        if (hl && hr) {
            if (tl < tr) { stack[sp] = n.right; st[sp++] = tr; stack[sp] = n.left; st[sp++] = tl; }
            else { stack[sp] = n.left; st[sp++] = tl; stack[sp] = n.right; st[sp++] = tr; }
        } else if (hl) { stack[sp] = n.left; st[sp++] = tl; }
        else if (hr) { stack[sp] = n.right; st[sp++] = tr; }
    }
    if (best < 1e30) c.hits++;
}

static double sah_cost(const std::vector<Node> &nodes)
{
    double c = 0; const double root = nodes[0].box.area();
    for (const Node &n : nodes) c += n.box.area() / root * (n.count ? n.count : 1.0);
    return c;
}

int main(int argc, char **argv)
{
    const uint32_t n_tris = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u, n_rays = argc > 2 ? (uint32_t)atoi(argv[2]) : 200000u;
    pth_scene hs{};
    if (pth_make_soup(n_tris, 1, &hs)) { fprintf(stderr, "soup failed\n"); return 1; }
    Scene s; s.v = hs.vertices; s.n = hs.n_tris; s.tb.resize(s.n); s.cen.resize(3 * (size_t)s.n);
    Box all;
    for (uint32_t t = 0; t < s.n; t++) {
        for (int k = 0; k < 3; k++) s.tb[t].grow(s.v + 9 * (size_t)t + 3 * k);
        for (int k = 0; k < 3; k++) s.cen[3 * (size_t)t + k] = 0.5f * (s.tb[t].lo[k] + s.tb[t].hi[k]);
        all.grow(s.tb[t]);
    }
    std::vector<uint32_t> key(s.n);
    for (uint32_t t = 0; t < s.n; t++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) q[k] = (uint32_t)std::min(1023.0f, std::max(0.0f, (s.cen[3 * (size_t)t + k] - all.lo[k]) / (all.hi[k] - all.lo[k]) * 1024.0f));
        key[t] = expand10(q[0]) << 2 | expand10(q[1]) << 1 | expand10(q[2]);
    }
    // rays: origin on a random triangle's centroid, uniform direction
    std::vector<V3> ro(n_rays), rd(n_rays);
    uint64_t rng = 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
    for (uint32_t i = 0; i < n_rays; i++) {
        const uint32_t t = (uint32_t)(rnd() * s.n) % s.n;
        ro[i] = { s.cen[3 * (size_t)t], s.cen[3 * (size_t)t + 1], s.cen[3 * (size_t)t + 2] };
        const double z = 2 * rnd() - 1, ph = 6.283185307179586 * rnd(), r = std::sqrt(1 - z * z);
        rd[i] = { r * std::cos(ph), r * std::sin(ph), z };
    }
    for (int leaf : { 1, 2, 4 }) {
        for (int kind = 0; kind < 2; kind++) {
            std::vector<uint32_t> order(s.n);
            std::iota(order.begin(), order.end(), 0u);
            std::vector<Node> nodes;
            nodes.reserve(2 * (size_t)s.n);
            if (kind == 0) {
                std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
                build_lbvh(s, order, key, nodes, 0, (int)s.n, 29, leaf);
            } else build_sah(s, order, nodes, 0, (int)s.n, leaf);
            Counts c;
            for (uint32_t i = 0; i < n_rays; i++) trace(s, nodes, order, ro[i], rd[i], c);
            printf("%-10s leaf<=%d: nodes %8zu  SAH cost %8.2f | per ray: inner visits %6.2f  leaf visits %5.2f  triangle tests %6.2f  hit %.3f\n",
                   kind ? "binnedSAH" : "LBVH", leaf, nodes.size(), sah_cost(nodes), c.nodes / n_rays, c.leaves / n_rays, c.tris / n_rays, c.hits / n_rays);
            fflush(stdout);
        }
    }
    return 0;
}
