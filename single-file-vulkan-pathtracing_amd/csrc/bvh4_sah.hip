// bvh4_sah.hip -- "prefer fast trace" BVH4 for small scenes (host code).
//
// The reference builds its acceleration structures with ePreferFastTrace (main.cpp:419): the driver may spend
// build time to make traversal cheap.  The device LBVH (lbvh_build.hip) is the fast BUILD; its Morton
// splits cost ~16 % more traversal work on the Cornell box than a surface-area split (3.9 vs 3.1 BVH4 nodes
// per ray, measured with the device visit counters).  For scenes of a few thousand triangles at most a full
// surface-area sweep is microseconds of host work, so those get their BVH4 from here:
//   * binary tree, top down: all three axes, every split position of the centroid order, cost
//     area(L)*n(L) + area(R)*n(R); a node of <= leaf_max triangles stays a leaf when splitting does not pay
//     (traversal step : triangle test = 1 : 0.6, the instruction ratio of k_extend);
//   * BVH4: a wide node starts from the two children and keeps opening the internal child of largest area
//     until it has four;
//   * same node format and the same box padding as the collapsed LBVH (pt_internal.h), so the traversal
//     kernels do not know the difference.  Hits do not depend on the BVH (closest t, lowest primitive id).
//   * PRIMITIVES are single triangles or FAN PAIRS (pair_with_next: triangle i+1 is (v0, v2, v3) of a quad whose first
//     half (v0, v1, v2) is triangle i -- what a loader makes of a quad): with pair leaves every leaf holds exactly one
//     primitive, the two halves of a quad stay adjacent in the leaf order, and the LDS traversal kernel tests them
//     with shared vertex transforms and edge function (extend_kernel.h: PAIRS).  Every other kernel sees an ordinary
//     1- or 2-triangle leaf.
// Everything is deterministic: double arithmetic, stable sorts, first minimum wins.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "pt_internal.h"
#include "pt_math.h"

namespace {

struct BNode {
    int left = -1, right = -1;  // -1: leaf
    uint32_t first = 0, count = 0;  // range of `ids`
    double lo[3], hi[3];
};

struct Builder {
    const float *tlo, *thi;      // per TRIANGLE
    uint32_t leaf_max;           // primitives per leaf
    std::vector<uint32_t> prim_first;  // primitive -> its first triangle; a primitive is 1 triangle or 2 consecutive ones
    std::vector<uint8_t> prim_tris;
    std::vector<double> plo, phi;      // per primitive box (3 each)
    std::vector<uint32_t> ids;         // primitive ids, permuted by the sweep
    std::vector<BNode> nodes;
    std::vector<double> area_l;

    static double area(const double *lo, const double *hi)
    {
        const double x = std::max(hi[0] - lo[0], 0.0), y = std::max(hi[1] - lo[1], 0.0), z = std::max(hi[2] - lo[2], 0.0);
        return 2.0 * (x * y + y * z + z * x);
    }
    void grow(double *lo, double *hi, uint32_t prim) const
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], plo[3 * (size_t)prim + k]);
            hi[k] = std::max(hi[k], phi[3 * (size_t)prim + k]);
        }
    }
    void init_prims(uint32_t n, const uint8_t *pair_with_next)
    {
        for (uint32_t t = 0; t < n;) {
            const uint32_t cnt = (pair_with_next && t + 1 < n && pair_with_next[t]) ? 2u : 1u;
            prim_first.push_back(t);
            prim_tris.push_back((uint8_t)cnt);
            for (int k = 0; k < 3; k++) {
                double lo = tlo[3 * (size_t)t + k], hi = thi[3 * (size_t)t + k];
                if (cnt == 2) { lo = std::min(lo, (double)tlo[3 * (size_t)(t + 1) + k]); hi = std::max(hi, (double)thi[3 * (size_t)(t + 1) + k]); }
                plo.push_back(lo);
                phi.push_back(hi);
            }
            t += cnt;
        }
    }
    // depth: the sweep has no balance term -- n coincident triangles cost the same at every split position and the
    // lowest one wins, a chain of depth n.  Below SAH_MAX_DEPTH the node is split at the median of the best axis'
    // centroid order instead, which bounds the height (and the traversal stack) by SAH_MAX_DEPTH + log2(n).
    static constexpr int SAH_MAX_DEPTH = 24;
    int build(uint32_t first, uint32_t count, int depth = 0)
    {
        const int me = (int)nodes.size();
        nodes.emplace_back();
        {
            BNode &nd = nodes[me];
            nd.first = first; nd.count = count;
            for (int k = 0; k < 3; k++) { nd.lo[k] = std::numeric_limits<double>::infinity(); nd.hi[k] = -nd.lo[k]; }
            for (uint32_t i = 0; i < count; i++) grow(nd.lo, nd.hi, ids[first + i]);
        }
        if (count <= 1) return me;
        double best = std::numeric_limits<double>::infinity();
        int best_axis = 0;
        uint32_t best_k = 0;
        std::vector<uint32_t> sorted[3];
        for (int ax = 0; ax < 3; ax++) {
            std::vector<uint32_t> &o = sorted[ax];
            o.assign(ids.begin() + first, ids.begin() + first + count);
            std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) {
                const double ca = plo[3 * (size_t)a + ax] + phi[3 * (size_t)a + ax];
                const double cb = plo[3 * (size_t)b + ax] + phi[3 * (size_t)b + ax];
                return ca < cb || (ca == cb && a < b);
            });
            area_l.assign(count, 0.0);
            double lo[3], hi[3];
            for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
            for (uint32_t i = 0; i + 1 < count; i++) { grow(lo, hi, o[i]); area_l[i] = area(lo, hi); }
            for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
            for (uint32_t i = count - 1; i >= 1; i--) {  // split after position i-1
                grow(lo, hi, o[i]);
                const double c = area_l[i - 1] * (double)i + area(lo, hi) * (double)(count - i);
                // sweep runs right to left: '<=' keeps the LOWEST split position among equal costs
                if (c < best || (c == best && ax == best_axis && i - 1 < best_k)) { best = c; best_axis = ax; best_k = i - 1; }
            }
        }
        const double a_node = area(nodes[me].lo, nodes[me].hi);
        if (count <= leaf_max && 0.6 * (double)count * a_node <= 1.0 * a_node + 0.6 * best) return me;  // leaf
        if (depth >= SAH_MAX_DEPTH) best_k = count / 2 - 1;
        std::copy(sorted[best_axis].begin(), sorted[best_axis].end(), ids.begin() + first);
        const int l = build(first, best_k + 1, depth + 1);
        const int r = build(first + best_k + 1, count - best_k - 1, depth + 1);
        nodes[me].left = l; nodes[me].right = r;
        return me;
    }
};

}  // namespace

uint32_t pt_wide_stack_need(const std::vector<uint32_t> &w)
{
    // a node with k children pushes at most k-1 of them before descending:
    // need(node) = k-1 + max over internal children (iterative, children always have larger indices or not -- use DFS)
    struct F { uint32_t node; uint32_t depth; };
    const size_t n = w.size() / 32;
    std::vector<uint32_t> need(n, 0);
    std::vector<int> state(n, 0);
    std::vector<uint32_t> stack{ 0u };
    while (!stack.empty()) {
        const uint32_t nd = stack.back();
        if (nd >= n || stack.size() > 4096) return 1u << 20;  // malformed: forces the spilling variant
        if (state[nd] == 0) {
            state[nd] = 1;
            for (int c = 0; c < 4; c++) {
                const uint32_t word = w[32 * (size_t)nd + 24 + c];
                if (word != 0xFFFFFFFFu && !(word & PT_LEAF)) stack.push_back(word);
            }
        } else {
            stack.pop_back();
            uint32_t k = 0, deepest = 0;
            for (int c = 0; c < 4; c++) {
                const uint32_t word = w[32 * (size_t)nd + 24 + c];
                if (word == 0xFFFFFFFFu) continue;
                k++;
                if (!(word & PT_LEAF) && word < n) deepest = std::max(deepest, need[word]);
            }
            need[nd] = (k ? k - 1 : 0) + deepest;
        }
    }
    return need[0];
}

void pt_sah_build_bvh4(const float *tlo, const float *thi, uint32_t n, const uint8_t *pair_with_next, float pad,
                       uint32_t leaf_max, std::vector<uint32_t> &rows, std::vector<uint32_t> &order)
{
    Builder b;
    b.tlo = tlo; b.thi = thi; b.leaf_max = std::max(leaf_max, 1u);
    b.init_prims(n, pair_with_next);
    const uint32_t np = (uint32_t)b.prim_first.size();
    b.ids.resize(np);
    for (uint32_t i = 0; i < np; i++) b.ids[i] = i;
    b.nodes.reserve(2 * (size_t)np);
    b.build(0, np);
    rows.clear();
    order.clear();
    order.reserve(n);
    const float inf = std::numeric_limits<float>::infinity();
    // emit wide nodes in pre-order (row 0 = root); explicit stack of (binary node, row)
    struct Item { int bnode; uint32_t row; };
    std::vector<Item> todo;
    auto new_row = [&]() -> uint32_t {
        const uint32_t r = (uint32_t)(rows.size() / 32);
        rows.resize(rows.size() + 32, 0u);
        float *f = reinterpret_cast<float *>(&rows[32 * (size_t)r]);
        for (int k = 0; k < 24; k++) f[k] = inf;  // empty slot: lo = hi = +inf
        for (int k = 24; k < 28; k++) rows[32 * (size_t)r + k] = 0xFFFFFFFFu;
        return r;
    };
    todo.push_back({ 0, new_row() });
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        int kids[4];
        int m = 0;
        const BNode &root = b.nodes[it.bnode];
        if (root.left < 0) kids[m++] = it.bnode;  // the whole scene is one leaf
        else { kids[m++] = root.left; kids[m++] = root.right; }
        while (m < 4) {
            int pick = -1;
            double pa = -1.0;
            for (int j = 0; j < m; j++) {
                const BNode &k = b.nodes[kids[j]];
                if (k.left < 0) continue;
                const double a = Builder::area(k.lo, k.hi);
                if (a > pa) { pa = a; pick = j; }
            }
            if (pick < 0) break;
            const BNode &k = b.nodes[kids[pick]];
            for (int j = m; j > pick + 1; j--) kids[j] = kids[j - 1];
            kids[pick] = k.left; kids[pick + 1] = k.right;
            m++;
        }
        // children are numbered in slot order; internal ones get their rows now so that indices are known
        for (int j = 0; j < m; j++) {
            const BNode &k = b.nodes[kids[j]];
            float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
            uint32_t n_tri = 0;
            for (uint32_t t = 0; t < k.count; t++) {
                const uint32_t prim = b.ids[k.first + t];
                for (uint32_t h = 0; h < b.prim_tris[prim]; h++, n_tri++) {
                    const uint32_t tri = b.prim_first[prim] + h;
                    for (int c = 0; c < 3; c++) {
                        lo[c] = std::min(lo[c], tlo[3 * (size_t)tri + c] - pad);  // float, like k_refit
                        hi[c] = std::max(hi[c], thi[3 * (size_t)tri + c] + pad);
                    }
                }
            }
            uint32_t word;
            if (k.left < 0) {
                word = PT_LEAF | ((n_tri - 1u) << 28) | (uint32_t)order.size();  // n_tri <= 8 (leaf_max <= 4 primitives)
                for (uint32_t t = 0; t < k.count; t++) {
                    const uint32_t prim = b.ids[k.first + t];
                    for (uint32_t h = 0; h < b.prim_tris[prim]; h++) order.push_back(b.prim_first[prim] + h);
                }
            } else {
                word = new_row();
                todo.push_back({ kids[j], word });
            }
            float *f = reinterpret_cast<float *>(&rows[32 * (size_t)it.row]);
            for (int c = 0; c < 3; c++) { f[4 * c + j] = lo[c]; f[12 + 4 * c + j] = hi[c]; }
            rows[32 * (size_t)it.row + 24 + j] = word;
        }
    }
}
