// rtx_bvh.cpp — host build of the 16-wide "box list" hierarchy used by the cooperative
// any-hit (shadow) traversal.
//
// This structure has NO counterpart in the reference: it only decides which primitives a shadow
// feeler is tested against.  The reference's any-hit answer is "some primitive intersects the
// feeler inside [EPSILON, dist]" (renderer.rs:209-217 with every transmission() == 0); its octree
// reaches every primitive a ray truly hits (the hit point lies in some leaf cell, the primitive
// is registered in every cell its bbox overlaps, octree.rs:101-128), so any conservative
// structure over the same primitives, followed by the exact f64 entry-bbox and primitive tests,
// gives the same answer.  Scenes with transmissive materials need the reference's candidate
// MULTISET (duplicates included) and stay on the octree (traverse_exact).
//
// Build: binned-SAH binary BVH down to single primitives, then collapsed: a wide node starts
// from a binary node's two children and keeps replacing the internal child of largest surface
// area by its two children until it holds RTX_WIDE children or only primitives.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <atomic>
#include <functional>
#include <thread>
#include <cmath>
#include <cstring>
#include <limits>

#include "rtx_bvh.h"

namespace {

struct BNode {
    double mn[3], mx[3];
    int left = -1, right = -1;   // binary children, or
    int prim = -1;               // leaf: index into the prim list
    double sa = 0.0;             // surface area of the box (the collapse picks by it)
    // collapse DP (see collapse_costs): cost[i-1] = least expected number of wide-node visits below this node when its subtree
    // is handed i child slots of a wide node; how[i-1] = 0: as with i - 1 slots, k > 0: left child gets k slots, right i - k;
    // how[0] is unused (one slot: the node becomes a wide node of its own, or is a primitive)
    // own = the split of the RTX_WIDE slots between the two children when this node is a wide node itself
    float cost[RTX_WIDE];
    uint8_t how[RTX_WIDE], own;
};

inline double area(const double *mn, const double *mx) {
    double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
inline void grow(double *mn, double *mx, const double *bmn, const double *bmx) {
    for (int a = 0; a < 3; ++a) {
        if (bmn[a] < mn[a]) mn[a] = bmn[a];
        if (bmx[a] > mx[a]) mx[a] = bmx[a];
    }
}
inline void reset(double *mn, double *mx) {
    for (int a = 0; a < 3; ++a) {
        mn[a] = std::numeric_limits<double>::infinity();
        mx[a] = -std::numeric_limits<double>::infinity();
    }
}
inline float f32_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float f32_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

constexpr int BINS = 16;
// what one pass over (a part of) a node's prims gathers for the binned SAH split
struct BinSet {
    int cnt[3][BINS];
    double bmn[3][BINS][3], bmx[3][BINS][3];
    void clear() {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < BINS; ++b) { cnt[a][b] = 0; reset(bmn[a][b], bmx[a][b]); }
    }
    void merge(const BinSet &o) {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < BINS; ++b) {
                if (!o.cnt[a][b]) continue;
                cnt[a][b] += o.cnt[a][b];
                grow(bmn[a][b], bmx[a][b], o.bmn[a][b], o.bmx[a][b]);
            }
    }
};

// Binned-SAH binary BVH.  The build is the same whatever the thread count: a node's split depends only on its own prim range.
// Large ranges (the top of the tree) are binned by all threads together; once a range is small its whole subtree is one task.
struct Builder {
    const std::vector<RtxPrimBox> &prims;
    std::vector<int> order;      // permutation of prim indices, partitioned in place
    std::vector<BNode> nodes;
    RtxPool &pool;
    unsigned n_threads;
    static constexpr int PAR_MIN = 8192;   // ranges at least this long are split with parallel passes; shorter ones are subtree tasks

    Builder(const std::vector<RtxPrimBox> &p, RtxPool &pl) : prims(p), order(p.size()), pool(pl), n_threads(pl.size()) {
        for (size_t i = 0; i < p.size(); ++i) order[i] = (int)i;
    }
    static int bin_of(const RtxPrimBox &p, int a, double cmn, double scale) {
        int b = (int)((0.5 * (p.mn[a] + p.mx[a]) - cmn) * scale);
        return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
    }
    void bounds_of(int begin, int end, double *mn, double *mx, double *cmn, double *cmx) const {
        reset(mn, mx);
        reset(cmn, cmx);
        for (int i = begin; i < end; ++i) {
            const RtxPrimBox &p = prims[(size_t)order[(size_t)i]];
            grow(mn, mx, p.mn, p.mx);
            double c[3] = {0.5 * (p.mn[0] + p.mx[0]), 0.5 * (p.mn[1] + p.mx[1]), 0.5 * (p.mn[2] + p.mx[2])};
            grow(cmn, cmx, c, c);
        }
    }
    void bin_range(int begin, int end, const double *cmn, const double *cmx, BinSet &bs) const {
        double scale[3];
        bool use[3];
        for (int a = 0; a < 3; ++a) {
            double ext = cmx[a] - cmn[a];
            use[a] = ext > 0.0;
            scale[a] = use[a] ? (double)BINS / ext : 0.0;
        }
        for (int i = begin; i < end; ++i) {
            const RtxPrimBox &p = prims[(size_t)order[(size_t)i]];
            for (int a = 0; a < 3; ++a) {
                if (!use[a]) continue;
                int b = bin_of(p, a, cmn[a], scale[a]);
                ++bs.cnt[a][b];
                grow(bs.bmn[a][b], bs.bmx[a][b], p.mn, p.mx);
            }
        }
    }
    // splits order[begin, end) (n >= 2) and returns the first index of the right part; nd receives the node's box
    int split(int begin, int end, BNode &nd, bool parallel) {
        const int n = end - begin;
        double cmn[3], cmx[3];
        BinSet bs;
        bs.clear();
        if (!parallel) {
            bounds_of(begin, end, nd.mn, nd.mx, cmn, cmx);
            bin_range(begin, end, cmn, cmx, bs);
        } else {
            const int parts = (int)n_threads * 4, len = (n + parts - 1) / parts;
            std::vector<std::array<double, 12>> pb((size_t)parts);
            pool.run((size_t)parts, [&](size_t t) {
                int b0 = begin + (int)t * len, b1 = std::min(end, b0 + len);
                double *q = pb[t].data();
                bounds_of(b0, std::max(b0, b1), q, q + 3, q + 6, q + 9);
            });
            reset(nd.mn, nd.mx);
            reset(cmn, cmx);
            for (int t = 0; t < parts; ++t) {
                grow(nd.mn, nd.mx, pb[(size_t)t].data(), pb[(size_t)t].data() + 3);
                grow(cmn, cmx, pb[(size_t)t].data() + 6, pb[(size_t)t].data() + 9);
            }
            std::vector<BinSet> part((size_t)parts);
            pool.run((size_t)parts, [&](size_t t) {
                int b0 = begin + (int)t * len, b1 = std::min(end, b0 + len);
                part[t].clear();
                if (b0 < b1) bin_range(b0, b1, cmn, cmx, part[t]);
            });
            for (int t = 0; t < parts; ++t) bs.merge(part[(size_t)t]);
        }
        // binned SAH over the three axes
        int best_axis = -1, best_split = -1;
        double best_cost = std::numeric_limits<double>::infinity();
        for (int a = 0; a < 3; ++a) {
            if (!(cmx[a] - cmn[a] > 0.0)) continue;
            double rarea[BINS];
            int rcnt[BINS];
            double amn[3], amx[3];
            reset(amn, amx);
            int c = 0;
            for (int b = BINS - 1; b > 0; --b) {
                if (bs.cnt[a][b]) grow(amn, amx, bs.bmn[a][b], bs.bmx[a][b]);
                c += bs.cnt[a][b];
                rarea[b] = c ? area(amn, amx) : 0.0;
                rcnt[b] = c;
            }
            reset(amn, amx);
            c = 0;
            for (int b = 0; b < BINS - 1; ++b) {
                if (bs.cnt[a][b]) grow(amn, amx, bs.bmn[a][b], bs.bmx[a][b]);
                c += bs.cnt[a][b];
                if (c == 0 || rcnt[b + 1] == 0) continue;
                double cost = area(amn, amx) * c + rarea[b + 1] * rcnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
            }
        }
        int mid;
        if (best_axis < 0) {
            mid = begin + n / 2;   // all centroids coincide: split the list in half
        } else {
            const int a = best_axis;
            const double scale = (double)BINS / (cmx[a] - cmn[a]), c0 = cmn[a];
            auto it = std::partition(order.begin() + begin, order.begin() + end,
                                     [&](int pi) { return bin_of(prims[(size_t)pi], a, c0, scale) <= best_split; });
            mid = (int)(it - order.begin());
            if (mid == begin || mid == end) mid = begin + n / 2;
        }
        return mid;
    }
    // Collapse DP after Ylitie et al. 2017 (wide BVHs by dynamic programming), for the cost the cooperative walk pays: a visit of a
    // wide node costs the same whatever it holds (16 lanes test its 16 slots at once), and a primitive is queued iff the beam touches
    // its own box, whatever the structure above it; so the collapse minimises the expected number of NODE VISITS, sum of area(node)
    // over wide nodes.  v[] must be in an order where children come after their parents (true for every vector built here).
    static void collapse_costs(std::vector<BNode> &v, size_t first, size_t last) {
        for (size_t idx = last; idx-- > first;) {
            BNode &n = v[idx];
            if (n.prim >= 0) {
                for (int i = 0; i < RTX_WIDE; ++i) { n.cost[i] = 0.0f; n.how[i] = 0; }
                n.own = 0;
                continue;
            }
            const BNode &l = v[(size_t)n.left], &r = v[(size_t)n.right];
            float d[RTX_WIDE + 1];
            uint8_t dk[RTX_WIDE + 1];
            for (int j = 2; j <= RTX_WIDE; ++j) {
                float best = std::numeric_limits<float>::infinity();
                int bk = 1;
                for (int k = 1; k < j; ++k) {
                    float c = l.cost[k - 1] + r.cost[j - k - 1];
                    if (c < best) { best = c; bk = k; }
                }
                d[j] = best;
                dk[j] = (uint8_t)bk;
            }
            static const double infl = std::getenv("RTX_BVH_INFLATE") ? std::atof(std::getenv("RTX_BVH_INFLATE")) : 0.0;
            const double ex = n.mx[0] - n.mn[0] + infl, ey = n.mx[1] - n.mn[1] + infl, ez = n.mx[2] - n.mn[2] + infl;
            n.cost[0] = (float)(2.0 * (ex * ey + ey * ez + ez * ex)) + d[RTX_WIDE];
            n.how[0] = 0;
            n.own = dk[RTX_WIDE];
            for (int i = 2; i <= RTX_WIDE; ++i) {
                if (d[i] < n.cost[i - 2]) { n.cost[i - 1] = d[i]; n.how[i - 1] = dk[i]; }
                else { n.cost[i - 1] = n.cost[i - 2]; n.how[i - 1] = 0; }
            }
        }
    }
    void leaf(int begin, BNode &nd) const {
        const RtxPrimBox &p = prims[(size_t)order[(size_t)begin]];
        for (int a = 0; a < 3; ++a) { nd.mn[a] = p.mn[a]; nd.mx[a] = p.mx[a]; }
        nd.prim = order[(size_t)begin];
    }
    // the subtree over order[begin, end) into `out` (out[0] = its root; children by local index), one thread
    void build_local(std::vector<BNode> &out, int begin, int end) {
        struct Task { int node, begin, end; };
        std::vector<Task> stack;
        out.emplace_back();
        stack.push_back({0, begin, end});
        while (!stack.empty()) {
            Task t = stack.back();
            stack.pop_back();
            BNode nd;
            if (t.end - t.begin == 1) {
                leaf(t.begin, nd);
                out[(size_t)t.node] = nd;
                continue;
            }
            int mid = split(t.begin, t.end, nd, false);
            nd.sa = area(nd.mn, nd.mx);
            nd.left = (int)out.size();
            out.emplace_back();
            nd.right = (int)out.size();
            out.emplace_back();
            out[(size_t)t.node] = nd;
            stack.push_back({nd.left, t.begin, mid});
            stack.push_back({nd.right, mid, t.end});
        }
        collapse_costs(out, 0, out.size());
    }
    // builds the tree over all prims; returns the root's index in `nodes`
    int build() {
        struct Task { int node, begin, end; };
        std::vector<Task> top, subtree;
        nodes.clear();
        nodes.emplace_back();
        top.push_back({0, 0, (int)prims.size()});
        while (!top.empty()) {   // the large ranges, one after the other, each with all threads
            Task t = top.back();
            top.pop_back();
            if (t.end - t.begin < PAR_MIN || n_threads == 1) {
                subtree.push_back(t);
                continue;
            }
            BNode nd;
            int mid = split(t.begin, t.end, nd, true);
            nd.sa = area(nd.mn, nd.mx);
            nd.left = (int)nodes.size();
            nodes.emplace_back();
            nd.right = (int)nodes.size();
            nodes.emplace_back();
            nodes[(size_t)t.node] = nd;
            top.push_back({nd.left, t.begin, mid});
            top.push_back({nd.right, mid, t.end});
        }
        const size_t n_top = nodes.size();
        std::vector<char> is_subtree_root(n_top, 0);
        for (const Task &t : subtree) is_subtree_root[(size_t)t.node] = 1;
        std::vector<std::vector<BNode>> local(subtree.size());
        pool.run(subtree.size(), [&](size_t i) { build_local(local[i], subtree[i].begin, subtree[i].end); });
        for (size_t i = 0; i < subtree.size(); ++i) {   // local node j > 0 becomes base + j - 1, local node 0 is nodes[subtree[i].node]
            const int base = (int)nodes.size() - 1;
            std::vector<BNode> &L = local[i];
            for (BNode &b : L)
                if (b.prim < 0) { b.left += base; b.right += base; }
            nodes[(size_t)subtree[i].node] = L[0];
            nodes.insert(nodes.end(), L.begin() + 1, L.end());
        }
        for (size_t idx = n_top; idx-- > 0;)   // the top nodes (split with parallel passes), children before parents
            if (!is_subtree_root[idx]) collapse_costs(nodes, idx, idx + 1);
        return 0;
    }
};

} // namespace

void rtx_build_wide_bvh(const std::vector<RtxPrimBox> &prims, const double centre[3], std::vector<DWideChild> &out,
                        RtxWideStats *stats, RtxPool &pool) {
    out.clear();
    auto add_node = [&]() {   // appends RTX_WIDE empty children, returns the node index
        DWideChild c;
        for (int a = 0; a < 3; ++a) { c.lo[a] = INFINITY; c.hi[a] = -INFINITY; }
        c.ref = RTX_WREF_EMPTY;
        c.first_node = 0;
        uint32_t idx = (uint32_t)(out.size() / RTX_WIDE);
        out.insert(out.end(), RTX_WIDE, c);
        return idx;
    };
    if (stats) *stats = RtxWideStats();
    if (prims.empty()) {
        add_node();
        if (stats) { stats->nodes = 1; stats->depth = 1; }
        return;
    }
    const bool timing = std::getenv("RTX_BUILD_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    Builder b(prims, pool);
    int broot = b.build();
    if (timing) std::fprintf(stderr, "    SAH binary tree %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    const std::vector<BNode> &bn = b.nodes;
    const bool greedy = std::getenv("RTX_BVH_GREEDY") != nullptr;

    struct Task { uint32_t wide; int bnode; int depth; };
    std::vector<Task> stack;
    std::vector<int> kids;
    kids.reserve(RTX_WIDE + 1);
    out.reserve((prims.size() / 6 + 16) * RTX_WIDE);
    stack.push_back({add_node(), broot, 1});
    int max_depth = 0;
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        kids.clear();
        if (bn[(size_t)t.bnode].prim >= 0) kids.push_back(t.bnode);   // single-prim scene
        else if (!greedy) {
            // the children the collapse DP chose (Builder::collapse_costs): least expected number of node visits
            struct Open { int node, slots; };
            std::vector<Open> open;
            const BNode &w = bn[(size_t)t.bnode];
            open.push_back({w.right, RTX_WIDE - (int)w.own});
            open.push_back({w.left, (int)w.own});
            while (!open.empty()) {
                Open o = open.back();
                open.pop_back();
                const BNode &k = bn[(size_t)o.node];
                while (o.slots > 1 && k.prim < 0 && k.how[o.slots - 1] == 0) --o.slots;   // no better than with one slot fewer
                if (k.prim >= 0 || o.slots == 1) { kids.push_back(o.node); continue; }      // a primitive, or a wide node of its own
                const int kl = (int)k.how[o.slots - 1];
                open.push_back({k.right, o.slots - kl});
                open.push_back({k.left, kl});
            }
        } else { kids.push_back(bn[(size_t)t.bnode].left); kids.push_back(bn[(size_t)t.bnode].right); }
        // (RTX_BVH_GREEDY: up to RTX_WIDE children by opening the largest internal child first)
        for (; greedy;) {
            if ((int)kids.size() >= RTX_WIDE) break;
            int pick = -1;
            double pa = -1.0;
            for (size_t i = 0; i < kids.size(); ++i) {
                const BNode &k = bn[(size_t)kids[i]];
                if (k.prim >= 0) continue;
                if (k.sa > pa) { pa = k.sa; pick = (int)i; }
            }
            if (pick < 0) break;
            int k = kids[(size_t)pick];
            kids[(size_t)pick] = bn[(size_t)k].left;
            kids.push_back(bn[(size_t)k].right);
        }
        max_depth = std::max(max_depth, t.depth);
        // node-kind children get CONSECUTIVE node indices in slot order (see DWideChild)
        uint32_t first_node = 0;
        bool have_first = false;
        for (size_t i = 0; i < kids.size(); ++i) {
            const BNode &k = bn[(size_t)kids[i]];
            uint32_t ref;
            if (k.prim >= 0) {
                ref = prims[(size_t)k.prim].ref;
                if (stats) ++stats->prim_children;
            } else {
                uint32_t idx = add_node();
                if (!have_first) { first_node = idx; have_first = true; }
                ref = RTX_WREF_NODE | idx;
                stack.push_back({idx, kids[i], t.depth + 1});
            }
            DWideChild &w = out[(size_t)t.wide * RTX_WIDE + i];
            for (int a = 0; a < 3; ++a) {
                w.lo[a] = f32_down(k.mn[a] - centre[a]);
                w.hi[a] = f32_up(k.mx[a] - centre[a]);
            }
            w.ref = ref;
        }
        for (int i = 0; i < RTX_WIDE; ++i) out[(size_t)t.wide * RTX_WIDE + (size_t)i].first_node = first_node;
    }
    if (stats) {
        stats->nodes = (int64_t)(out.size() / RTX_WIDE);
        stats->depth = max_depth;
    }
}
