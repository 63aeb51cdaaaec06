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
#include <cmath>
#include <cstring>
#include <limits>

#include "rtx_bvh.h"

namespace {

struct BNode {
    double mn[3], mx[3];
    int left = -1, right = -1;   // binary children, or
    int prim = -1;               // leaf: index into the prim list
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

struct Builder {
    const std::vector<RtxPrimBox> &prims;
    std::vector<int> order;      // permutation of prim indices, partitioned in place
    std::vector<BNode> nodes;
    static constexpr int BINS = 16;

    explicit Builder(const std::vector<RtxPrimBox> &p) : prims(p), order(p.size()) {
        for (size_t i = 0; i < p.size(); ++i) order[i] = (int)i;
        nodes.reserve(p.size() * 2 + 1);
    }
    // builds the subtree over order[begin, end); returns its node index
    int build(int begin, int end) {
        struct Task { int node, begin, end; };
        std::vector<Task> stack;
        int root = (int)nodes.size();
        nodes.emplace_back();
        stack.push_back({root, begin, end});
        while (!stack.empty()) {
            Task t = stack.back();
            stack.pop_back();
            BNode nd;
            reset(nd.mn, nd.mx);
            double cmn[3], cmx[3];
            reset(cmn, cmx);
            for (int i = t.begin; i < t.end; ++i) {
                const RtxPrimBox &p = prims[(size_t)order[(size_t)i]];
                grow(nd.mn, nd.mx, p.mn, p.mx);
                double c[3] = {0.5 * (p.mn[0] + p.mx[0]), 0.5 * (p.mn[1] + p.mx[1]), 0.5 * (p.mn[2] + p.mx[2])};
                grow(cmn, cmx, c, c);
            }
            int n = t.end - t.begin;
            if (n == 1) {
                nd.prim = order[(size_t)t.begin];
                nodes[(size_t)t.node] = nd;
                continue;
            }
            // binned SAH over the three axes
            int best_axis = -1, best_split = -1;
            double best_cost = std::numeric_limits<double>::infinity();
            for (int a = 0; a < 3; ++a) {
                double ext = cmx[a] - cmn[a];
                if (!(ext > 0.0)) continue;
                int cnt[BINS] = {0};
                double bmn[BINS][3], bmx[BINS][3];
                for (int b = 0; b < BINS; ++b) reset(bmn[b], bmx[b]);
                double scale = (double)BINS / ext;
                for (int i = t.begin; i < t.end; ++i) {
                    const RtxPrimBox &p = prims[(size_t)order[(size_t)i]];
                    int b = (int)((0.5 * (p.mn[a] + p.mx[a]) - cmn[a]) * scale);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    ++cnt[b];
                    grow(bmn[b], bmx[b], p.mn, p.mx);
                }
                double rarea[BINS];
                int rcnt[BINS];
                double amn[3], amx[3];
                reset(amn, amx);
                int c = 0;
                for (int b = BINS - 1; b > 0; --b) {
                    if (cnt[b]) grow(amn, amx, bmn[b], bmx[b]);
                    c += cnt[b];
                    rarea[b] = c ? area(amn, amx) : 0.0;
                    rcnt[b] = c;
                }
                reset(amn, amx);
                c = 0;
                for (int b = 0; b < BINS - 1; ++b) {
                    if (cnt[b]) grow(amn, amx, bmn[b], bmx[b]);
                    c += cnt[b];
                    if (c == 0 || rcnt[b + 1] == 0) continue;
                    double cost = area(amn, amx) * c + rarea[b + 1] * rcnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
                }
            }
            int mid;
            if (best_axis < 0) {
                mid = t.begin + n / 2;   // all centroids coincide: split the list in half
            } else {
                double ext = cmx[best_axis] - cmn[best_axis];
                double scale = (double)BINS / ext;
                int a = best_axis;
                auto it = std::partition(order.begin() + t.begin, order.begin() + t.end, [&](int pi) {
                    const RtxPrimBox &p = prims[(size_t)pi];
                    int b = (int)((0.5 * (p.mn[a] + p.mx[a]) - cmn[a]) * scale);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    return b <= best_split;
                });
                mid = (int)(it - order.begin());
                if (mid == t.begin || mid == t.end) mid = t.begin + n / 2;
            }
            nd.left = (int)nodes.size();
            nodes.emplace_back();
            nd.right = (int)nodes.size();
            nodes.emplace_back();
            nodes[(size_t)t.node] = nd;
            stack.push_back({nd.left, t.begin, mid});
            stack.push_back({nd.right, mid, t.end});
        }
        return root;
    }
};

} // namespace

void rtx_build_wide_bvh(const std::vector<RtxPrimBox> &prims, const double centre[3], std::vector<DWideChild> &out,
                        RtxWideStats *stats) {
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
    Builder b(prims);
    int broot = b.build(0, (int)prims.size());
    const std::vector<BNode> &bn = b.nodes;

    struct Task { uint32_t wide; int bnode; int depth; };
    std::vector<Task> stack;
    stack.push_back({add_node(), broot, 1});
    int max_depth = 0;
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        // collect up to RTX_WIDE children by opening the largest internal child first
        std::vector<int> kids;
        if (bn[(size_t)t.bnode].prim >= 0) kids.push_back(t.bnode);   // single-prim scene
        else { kids.push_back(bn[(size_t)t.bnode].left); kids.push_back(bn[(size_t)t.bnode].right); }
        for (;;) {
            if ((int)kids.size() >= RTX_WIDE) break;
            int pick = -1;
            double pa = -1.0;
            for (size_t i = 0; i < kids.size(); ++i) {
                const BNode &k = bn[(size_t)kids[i]];
                if (k.prim >= 0) continue;
                double ar = area(k.mn, k.mx);
                if (ar > pa) { pa = ar; pick = (int)i; }
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
