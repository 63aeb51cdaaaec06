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
#include <numeric>

#include "rtx_bvh.h"

namespace {

struct BNode {
    double mn[3], mx[3];
    int left = -1, right = -1;   // binary children, or
    int prim = -1;               // leaf: index into the prim list
    double sa = 0.0;             // surface area of the box (the collapse picks by it)
    // collapse DP (see collapse_costs), once per width: cost[i-1] = least expected number of wide-node visits below this node when its
    // subtree is handed i child slots of a wide node; how[i-1] = 0: as with i - 1 slots, k > 0: left child gets k slots, right i - k;
    // how[0] is unused (one slot: the node becomes a wide node of its own, or is a primitive); own = the split of the W slots
    // between the two children when this node is a wide node itself
    template <int W>
    struct Collapse {
        float cost[W];
        uint8_t how[W], own;
    };
    Collapse<RTX_WIDE> c16;    // the 16-wide box lists of the cooperative shadow walk
    Collapse<RTX_BVH4> c4;     // the 4-wide nodes of the per-lane walks
    template <int W> Collapse<W> &collapse();
    template <int W> const Collapse<W> &collapse() const;
};
template <> inline BNode::Collapse<RTX_WIDE> &BNode::collapse<RTX_WIDE>() { return c16; }
template <> inline BNode::Collapse<RTX_BVH4> &BNode::collapse<RTX_BVH4>() { return c4; }
template <> inline const BNode::Collapse<RTX_WIDE> &BNode::collapse<RTX_WIDE>() const { return c16; }
template <> inline const BNode::Collapse<RTX_BVH4> &BNode::collapse<RTX_BVH4>() const { return c4; }


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
// a float <= x / >= x (x finite): the neighbour of the rounded value where rounding went the wrong way (std::nextafterf, inlined:
// 1.7 M calls per 262 k prims)
inline float f32_step(float f, bool up) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    if ((b & 0x7FFFFFFFu) == 0u) b = up ? 0x00000001u : 0x80000001u;   // +-0 -> the smallest denormal of the wanted sign
    else if (((b >> 31) != 0u) == up) --b;                              // towards zero
    else ++b;                                                           // away from zero
    std::memcpy(&f, &b, 4);
    return f;
}
inline float f32_down(double x) {
    float f = (float)x;
    return (double)f > x ? f32_step(f, false) : f;
}
inline float f32_up(double x) {
    float f = (float)x;
    return (double)f < x ? f32_step(f, true) : f;
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
    std::vector<int> order;      // permutation of prim indices, partitioned in place (stably: the tree does not depend on the thread count)
    std::vector<int> scratch;    // copy of a range while all threads partition it
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
    // A short range (2 .. SWEEP_MAX prims): the exact SAH sweep over the prims sorted by centroid, per axis.  (Binning such a range
    // cost a 4.8 KB BinSet cleared and 45 candidate planes evaluated per node, and most nodes of a tree built down to single prims
    // are this short.)
    static constexpr int SWEEP_MAX = 12;
    int split_sweep(int begin, int end, BNode &nd) {
        const int n = end - begin;
        int idx[3][SWEEP_MAX];
        double cen[3][SWEEP_MAX];
        reset(nd.mn, nd.mx);
        for (int i = 0; i < n; ++i) {
            const RtxPrimBox &p = prims[(size_t)order[(size_t)(begin + i)]];
            grow(nd.mn, nd.mx, p.mn, p.mx);
            for (int a = 0; a < 3; ++a) {   // insertion sort by centroid (stable: equal centroids keep the list's order)
                const double c = 0.5 * (p.mn[a] + p.mx[a]);
                int j = i;
                while (j > 0 && cen[a][j - 1] > c) { cen[a][j] = cen[a][j - 1]; idx[a][j] = idx[a][j - 1]; --j; }
                cen[a][j] = c;
                idx[a][j] = order[(size_t)(begin + i)];
            }
        }
        int best_axis = -1, best_k = 0;
        double best_cost = std::numeric_limits<double>::infinity();
        for (int a = 0; a < 3; ++a) {
            if (!(cen[a][n - 1] > cen[a][0])) continue;
            double rarea[SWEEP_MAX];
            double amn[3], amx[3];
            reset(amn, amx);
            for (int i = n - 1; i > 0; --i) {
                const RtxPrimBox &p = prims[(size_t)idx[a][i]];
                grow(amn, amx, p.mn, p.mx);
                rarea[i] = area(amn, amx);
            }
            reset(amn, amx);
            for (int k = 1; k < n; ++k) {   // left = the first k prims
                const RtxPrimBox &p = prims[(size_t)idx[a][k - 1]];
                grow(amn, amx, p.mn, p.mx);
                const double cost = area(amn, amx) * k + rarea[k] * (n - k);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
            }
        }
        if (best_axis < 0) return begin + n / 2;   // all centroids coincide: split the list in half
        for (int i = 0; i < n; ++i) order[(size_t)(begin + i)] = idx[best_axis][i];
        return begin + best_k;
    }
    // splits order[begin, end) (n >= 2) and returns the first index of the right part; nd receives the node's box
    int split(int begin, int end, BNode &nd, bool parallel) {
        const int n = end - begin;
        if (n <= SWEEP_MAX) return split_sweep(begin, end, nd);
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
            auto is_left = [&](int pi) { return bin_of(prims[(size_t)pi], a, c0, scale) <= best_split; };
            if (!parallel) {
                mid = (int)(std::stable_partition(order.begin() + begin, order.begin() + end, is_left) - order.begin());
            } else {   // the same stable partition by all threads: count per part, then scatter through a copy
                const int parts = (int)n_threads * 4, len = (n + parts - 1) / parts;
                std::vector<int> n_left((size_t)parts + 1, 0);
                scratch.resize(order.size());
                pool.run((size_t)parts, [&](size_t t) {
                    int b0 = begin + (int)t * len, b1 = std::min(end, b0 + len), c = 0;
                    for (int i = b0; i < b1; ++i) {
                        scratch[(size_t)i] = order[(size_t)i];
                        c += is_left(order[(size_t)i]) ? 1 : 0;
                    }
                    n_left[t + 1] = c;
                });
                for (int t = 0; t < parts; ++t) n_left[(size_t)t + 1] += n_left[(size_t)t];
                mid = begin + n_left[(size_t)parts];
                pool.run((size_t)parts, [&](size_t t) {
                    int b0 = begin + (int)t * len, b1 = std::min(end, b0 + len);
                    int l = begin + n_left[t], r = mid + (std::min(b0, end) - begin - n_left[t]);
                    for (int i = b0; i < b1; ++i) {
                        const int pi = scratch[(size_t)i];
                        if (is_left(pi)) order[(size_t)l++] = pi;
                        else order[(size_t)r++] = pi;
                    }
                });
            }
            if (mid == begin || mid == end) mid = begin + n / 2;
        }
        return mid;
    }
    // Collapse DP after Ylitie et al. 2017 (wide BVHs by dynamic programming), for the cost the cooperative walk pays: a visit of a
    // wide node costs the same whatever it holds (16 lanes test its 16 slots at once), and a primitive is queued iff the beam touches
    // its own box, whatever the structure above it; so the collapse minimises the expected number of NODE VISITS, sum of area(node)
    // over wide nodes.  v[] must be in an order where children come after their parents (true for every vector built here).
    template <int W>
    static void collapse_node(BNode &n, const BNode &l, const BNode &r) {
        BNode::Collapse<W> &c = n.collapse<W>();
        const BNode::Collapse<W> &lc = l.collapse<W>(), &rc = r.collapse<W>();
        float d[W + 1];
        uint8_t dk[W + 1];
        for (int j = 2; j <= W; ++j) {
            float best = std::numeric_limits<float>::infinity();
            int bk = 1;
            for (int k = 1; k < j; ++k) {
                float cc = lc.cost[k - 1] + rc.cost[j - k - 1];
                if (cc < best) { best = cc; bk = k; }
            }
            d[j] = best;
            dk[j] = (uint8_t)bk;
        }
        c.cost[0] = (float)n.sa + d[W];
        c.how[0] = 0;
        c.own = dk[W];
        for (int i = 2; i <= W; ++i) {
            if (d[i] < c.cost[i - 2]) { c.cost[i - 1] = d[i]; c.how[i - 1] = dk[i]; }
            else { c.cost[i - 1] = c.cost[i - 2]; c.how[i - 1] = 0; }
        }
    }
    static void collapse_costs(std::vector<BNode> &v, size_t first, size_t last) {
        for (size_t idx = last; idx-- > first;) {
            BNode &n = v[idx];
            if (n.prim >= 0) {
                std::memset(&n.c16, 0, sizeof n.c16);
                std::memset(&n.c4, 0, sizeof n.c4);
                continue;
            }
            collapse_node<RTX_WIDE>(n, v[(size_t)n.left], v[(size_t)n.right]);
            collapse_node<RTX_BVH4>(n, v[(size_t)n.left], v[(size_t)n.right]);
        }
    }
    void leaf(int begin, BNode &nd) const {
        const RtxPrimBox &p = prims[(size_t)order[(size_t)begin]];
        for (int a = 0; a < 3; ++a) { nd.mn[a] = p.mn[a]; nd.mx[a] = p.mx[a]; }
        nd.prim = order[(size_t)begin];
    }
    // the subtree over order[begin, end): its root is nodes[root], its other 2n - 2 nodes are nodes[base ...) in the order they are
    // created (children after their parent), one thread
    void build_local(int root, int base, int begin, int end) {
        struct Task { int node, begin, end; };
        std::vector<Task> stack;
        int next = base;
        stack.push_back({root, begin, end});
        while (!stack.empty()) {
            Task t = stack.back();
            stack.pop_back();
            BNode &nd = nodes[(size_t)t.node];
            if (t.end - t.begin == 1) {
                leaf(t.begin, nd);
                continue;
            }
            int mid = split(t.begin, t.end, nd, false);
            nd.sa = area(nd.mn, nd.mx);
            nd.left = next++;
            nd.right = next++;
            stack.push_back({nd.left, t.begin, mid});
            stack.push_back({nd.right, mid, t.end});
        }
        // collapse costs, children before parents: the subtree's own nodes backwards, then its root
        collapse_costs(nodes, (size_t)base, (size_t)next);
        collapse_costs(nodes, (size_t)root, (size_t)root + 1);
    }
    // builds the tree over all prims; returns the root's index in `nodes`
    int build() {
        struct Task { int node, begin, end; };
        const bool timing = std::getenv("RTX_BUILD_TIMING") != nullptr;
        auto tp = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "      %-24s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
            tp = now;
        };
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
        lap("top levels");
        const size_t n_top = nodes.size();
        std::vector<char> is_subtree_root(n_top, 0);
        std::vector<int> base(subtree.size());
        size_t total = n_top;
        for (size_t i = 0; i < subtree.size(); ++i) {
            is_subtree_root[(size_t)subtree[i].node] = 1;
            base[i] = (int)total;
            total += 2 * (size_t)(subtree[i].end - subtree[i].begin) - 2;
        }
        nodes.resize(total);
        pool.run(subtree.size(), [&](size_t i) { build_local(subtree[i].node, base[i], subtree[i].begin, subtree[i].end); });
        lap("subtrees");
        for (size_t idx = n_top; idx-- > 0;)   // the top nodes (split with parallel passes), children before parents
            if (!is_subtree_root[idx]) collapse_costs(nodes, idx, idx + 1);
        lap("top collapse costs");
        return 0;
    }
};

// The W-wide nodes the collapse DP chose, level by level (breadth first), every level by all threads: pass 1 lists the children of
// each node of the level (Builder::collapse_costs: least expected number of node visits) and counts those that are nodes; a prefix
// sum gives every node the index of its first node-child (node-kind children get CONSECUTIVE indices in slot order; the next level
// starts right behind this one); pass 2 hands each node to `write(node index, kids, n kids, index of its first node-child)`.
// `grow(n)` is called before a level's nodes are written: the output must then hold n nodes.
template <int W, class Grow, class Write>
void emit_levels(const std::vector<BNode> &bn, int root, RtxPool &pool, Grow grow, Write write, int *depth_out, int64_t *prim_children_out) {
    std::vector<int> level{root}, next_level, kids;
    std::vector<uint8_t> n_kids;
    std::vector<uint32_t> first;
    uint32_t level_first = 0;
    int depth = 0;
    int64_t prim_children = 0;
    constexpr size_t CHUNK = 128;
    grow((size_t)1);
    while (!level.empty()) {
        ++depth;
        const size_t n = level.size(), n_chunks = (n + CHUNK - 1) / CHUNK;
        kids.resize(n * W);
        n_kids.resize(n);
        first.assign(n + 1, 0u);
        pool.run(n_chunks, [&](size_t c) {
            struct Open { int node, slots; };
            Open open[2 * W];
            for (size_t i = c * CHUNK; i < std::min(n, (c + 1) * CHUNK); ++i) {
                int *k_out = &kids[i * W];
                int nk = 0, nn = 0, top = 0;
                const BNode &w = bn[(size_t)level[i]];
                if (w.prim >= 0) {
                    k_out[nk++] = level[i];   // single-prim scene
                } else {
                    const int own = (int)w.collapse<W>().own;
                    open[top++] = {w.right, W - own};
                    open[top++] = {w.left, own};
                }
                while (top > 0) {
                    Open o = open[--top];
                    const BNode &k = bn[(size_t)o.node];
                    const uint8_t *how = k.collapse<W>().how;
                    while (o.slots > 1 && k.prim < 0 && how[o.slots - 1] == 0) --o.slots;   // no better than with one slot fewer
                    if (k.prim >= 0 || o.slots == 1) {                                        // a primitive, or a wide node of its own
                        k_out[nk++] = o.node;
                        nn += k.prim < 0 ? 1 : 0;
                        continue;
                    }
                    const int kl = (int)how[o.slots - 1];
                    open[top++] = {k.right, o.slots - kl};
                    open[top++] = {k.left, kl};
                }
                n_kids[i] = (uint8_t)nk;
                first[i + 1] = (uint32_t)nn;
            }
        });
        for (size_t i = 0; i < n; ++i) first[i + 1] += first[i];
        const uint32_t next_first = level_first + (uint32_t)n, n_next = first[n];
        grow((size_t)next_first + n_next);
        next_level.resize(n_next);
        prim_children += (int64_t)std::accumulate(n_kids.begin(), n_kids.end(), (int64_t)0) - (int64_t)n_next;
        pool.run(n_chunks, [&](size_t c) {
            for (size_t i = c * CHUNK; i < std::min(n, (c + 1) * CHUNK); ++i) {
                const int *k_in = &kids[i * W];
                uint32_t nn = first[i];
                for (int j = 0; j < (int)n_kids[i]; ++j)
                    if (bn[(size_t)k_in[j]].prim < 0) next_level[nn++] = k_in[j];
                write(level_first + (uint32_t)i, k_in, (int)n_kids[i], next_first + first[i]);
            }
        });
        level_first = next_first;
        level.swap(next_level);
    }
    *depth_out = depth;
    *prim_children_out = prim_children;
}

} // namespace

void rtx_build_hierarchies(const std::vector<RtxPrimBox> &prims, const double centre[3], std::vector<DWideChild> &out,
                           std::vector<DNode4> &out4, RtxWideStats *stats, RtxPool &pool) {
    out.clear();
    out4.clear();
    DWideChild empty;
    for (int a = 0; a < 3; ++a) { empty.lo[a] = INFINITY; empty.hi[a] = -INFINITY; }
    empty.ref = RTX_WREF_EMPTY;
    empty.first_node = 0;
    DNode4 empty4;
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < RTX_BVH4; ++j) { empty4.lo[a][j] = INFINITY; empty4.hi[a][j] = -INFINITY; }
    for (int j = 0; j < RTX_BVH4; ++j) empty4.ref[j] = RTX_WREF_EMPTY;
    if (stats) *stats = RtxWideStats();
    if (prims.empty()) {
        out.assign(RTX_WIDE, empty);
        out4.assign(1, empty4);
        if (stats) { stats->nodes = 1; stats->depth = 1; stats->nodes4 = 1; stats->depth4 = 1; }
        return;
    }
    const bool timing = std::getenv("RTX_BUILD_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    Builder b(prims, pool);
    int broot = b.build();
    if (timing) std::fprintf(stderr, "    SAH binary tree %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    const std::vector<BNode> &bn = b.nodes;

    int depth = 0, depth4 = 0;
    int64_t prim_children = 0, prim_children4 = 0;
    emit_levels<RTX_WIDE>(
        bn, broot, pool, [&](size_t n_nodes) { out.resize(n_nodes * RTX_WIDE, empty); },
        [&](uint32_t node, const int *kids, int nk, uint32_t first_node) {
            DWideChild *rec = &out[(size_t)node * RTX_WIDE];
            uint32_t nn = first_node;
            for (int j = 0; j < nk; ++j) {
                const BNode &k = bn[(size_t)kids[j]];
                DWideChild &w = rec[j];
                for (int a = 0; a < 3; ++a) {
                    w.lo[a] = f32_down(k.mn[a] - centre[a]);
                    w.hi[a] = f32_up(k.mx[a] - centre[a]);
                }
                w.ref = k.prim >= 0 ? prims[(size_t)k.prim].ref : (RTX_WREF_NODE | nn++);
            }
            for (int j = 0; j < RTX_WIDE; ++j) rec[j].first_node = nn > first_node ? first_node : 0u;
        },
        &depth, &prim_children);
    emit_levels<RTX_BVH4>(
        bn, broot, pool, [&](size_t n_nodes) { out4.resize(n_nodes, empty4); },
        [&](uint32_t node, const int *kids, int nk, uint32_t first_node) {
            DNode4 &rec = out4[(size_t)node];
            uint32_t nn = first_node;
            for (int j = 0; j < nk; ++j) {
                const BNode &k = bn[(size_t)kids[j]];
                for (int a = 0; a < 3; ++a) {
                    rec.lo[a][j] = f32_down(k.mn[a] - centre[a]);
                    rec.hi[a][j] = f32_up(k.mx[a] - centre[a]);
                }
                rec.ref[j] = k.prim >= 0 ? prims[(size_t)k.prim].ref : (RTX_WREF_NODE | nn++);
            }
        },
        &depth4, &prim_children4);
    if (stats) {
        stats->nodes = (int64_t)(out.size() / RTX_WIDE);
        stats->depth = depth;
        stats->prim_children = prim_children;
        stats->nodes4 = (int64_t)out4.size();
        stats->depth4 = depth4;
    }
}
