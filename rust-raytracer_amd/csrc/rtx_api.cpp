// rtx_api.cpp — the C ABI of include/rtx.h: scene flattening + octree build on the host,
// upload, and the wavefront orchestration of the gfx950 kernels in rtx_kernels.hip.
//
// There is no CPU rendering path in this library: without a HIP device every entry point
// that renders fails with RTX_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <functional>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "rtx_bvh.h"
#include "rtx_build.h"
#include "rtx_device.h"

namespace {

thread_local std::string g_last_error;

rtx_status fail(rtx_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}
#define HIPCHECK(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(e_ == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    } while (0)

// ---------------------------------------------------------------------------- host octree
// Octree::from_iter + OctreeNode::{subdivide,insert} (raytracer/octree.rs:15-43, 64-133),
// with index-linked nodes instead of nested Vecs: the 8 children of a node are allocated
// contiguously at subdivision time, which is exactly the layout the device wants.
struct HBox {
    double mn[3], mx[3];
};
bool overlaps(const HBox &a, const HBox &o) {                      // geometry/bbox.rs:167-173
    bool x = a.mx[0] >= o.mn[0] && a.mn[0] <= o.mx[0];
    bool y = a.mx[1] >= o.mn[1] && a.mn[1] <= o.mx[1];
    bool z = a.mx[2] >= o.mn[2] && a.mn[2] <= o.mx[2];
    return x && y && z;
}
// One node of the finished octree: leaf entries are a range of HostOctree::entries (prim indices, insertion order).
struct FNode {
    HBox bbox;
    int depth;
    int first_child = -1;     // its 8 children are contiguous (octree.rs:75-98), -1: leaf
    int ebegin = 0, ecount = 0;
};
struct HostOctree {
    std::vector<FNode> nodes;     // children always have larger indices than their parent
    std::vector<int> entries;
    std::vector<HBox> prim_box;   // finite prims, insertion order
    int depth = 0;

    static void child_boxes(const HBox &b, HBox out[8]) {          // octree.rs:75-98, child k = x*4 + y*2 + z
        int k = 0;
        for (unsigned x = 0; x < 2; ++x)
            for (unsigned y = 0; y < 2; ++y)
                for (unsigned z = 0; z < 2; ++z, ++k) {
                    double lx = b.mx[0] - b.mn[0], ly = b.mx[1] - b.mn[1], lz = b.mx[2] - b.mn[2];
                    out[k].mn[0] = b.mn[0] + (double)x * lx / 2.0;
                    out[k].mn[1] = b.mn[1] + (double)y * ly / 2.0;
                    out[k].mn[2] = b.mn[2] + (double)z * lz / 2.0;
                    out[k].mx[0] = b.mx[0] - (double)(1 - x) * lx / 2.0;
                    out[k].mx[1] = b.mx[1] - (double)(1 - y) * ly / 2.0;
                    out[k].mx[2] = b.mx[2] - (double)(1 - z) * lz / 2.0;
                }
    }
    HBox root_bounds() const {                                     // BBox::from_union, bbox.rs:85-97 (BBox::zero() when empty)
        HBox bounds{{0, 0, 0}, {0, 0, 0}};
        for (size_t i = 0; i < prim_box.size(); ++i) {
            const HBox &p = prim_box[i];
            if (i == 0) bounds = p;
            else
                for (int a = 0; a < 3; ++a) {
                    bounds.mn[a] = std::fmin(bounds.mn[a], p.mn[a]);
                    bounds.mx[a] = std::fmax(bounds.mx[a], p.mx[a]);
                }
        }
        return bounds;
    }
    int root_depth() const {                                       // octree.rs:29
        double dd = std::round(1.2 * (std::log((double)prim_box.size()) / std::log(8.0)));
        // `as i32` saturates (N = 0 gives -inf -> i32::MIN: the root stays a leaf)
        return dd != dd ? 0 : (dd <= -2147483648.0 ? INT32_MIN : (dd >= 2147483647.0 ? INT32_MAX : (int)dd));
    }

    // ---- Octree::from_iter + OctreeNode::{subdivide, insert} (raytracer/octree.rs:15-43, 64-133), statement for statement, with
    // index-linked nodes instead of nested Vecs.  Kept as the cross-check of the parallel build below (RTX_BUILD_THREADS=0).
    struct SNode {
        HBox bbox;
        int depth;
        int first_child = -1;
        std::vector<int> entries;
    };
    void seq_subdivide(std::vector<SNode> &sn, int ni) const {     // octree.rs:75-98
        int first = (int)sn.size();
        HBox cb[8];
        child_boxes(sn[(size_t)ni].bbox, cb);
        int d = sn[(size_t)ni].depth;
        for (int c = 0; c < 8; ++c) {
            SNode ch;
            ch.bbox = cb[c];
            ch.depth = d - 1;
            sn.push_back(std::move(ch));
        }
        sn[(size_t)ni].first_child = first;
    }
    void seq_insert(std::vector<SNode> &sn, int ni, int index) const {   // octree.rs:101-128
        if (sn[(size_t)ni].depth <= 0) {
            sn[(size_t)ni].entries.push_back(index);
            return;
        }
        bool leaf = sn[(size_t)ni].first_child < 0;
        if (leaf && sn[(size_t)ni].entries.empty()) {
            sn[(size_t)ni].entries.push_back(index);
            return;
        }
        if (leaf && sn[(size_t)ni].entries.size() == 1) {
            seq_subdivide(sn, ni);
            int old = sn[(size_t)ni].entries[0];
            sn[(size_t)ni].entries.clear();
            seq_insert(sn, ni, old);
        }
        int first = sn[(size_t)ni].first_child;
        for (int c = 0; c < 8; ++c)
            if (overlaps(sn[(size_t)(first + c)].bbox, prim_box[(size_t)index])) seq_insert(sn, first + c, index);
    }
    void build() {                                                 // octree.rs:15-43
        depth = root_depth();
        std::vector<SNode> sn;
        SNode root;
        root.bbox = root_bounds();
        root.depth = depth;
        sn.push_back(std::move(root));
        for (size_t i = 0; i < prim_box.size(); ++i) seq_insert(sn, 0, (int)i);
        nodes.resize(sn.size());
        entries.clear();
        for (size_t i = 0; i < sn.size(); ++i) {
            FNode &f = nodes[i];
            f.bbox = sn[i].bbox; f.depth = sn[i].depth; f.first_child = sn[i].first_child;
            f.ebegin = (int)entries.size(); f.ecount = (int)sn[i].entries.size();
            entries.insert(entries.end(), sn[i].entries.begin(), sn[i].entries.end());
        }
    }

    // ---- the same tree, built top-down and in parallel -------------------------------------------------------------------
    // What insert() does to a node depends only on the node's depth and on the ORDERED list of prims that reach it: a node of
    // depth <= 0 keeps them all (octree.rs:103-106); with depth > 0 it keeps a lone prim (:109-112) and, from the second prim on,
    // is an interior node that has handed every prim, the first one included (:114-120), in order, to each child whose bbox
    // overlaps the prim's (:122-127).  So the subtree of a node is a function of (bbox, depth, ordered prim list): interior iff
    // depth > 0 and the list has two or more prims; child c's list = the prims of the list that overlap c's bbox, in order.
    // Sibling subtrees are independent, which the sequential insertion hides: the top two levels are expanded here and the
    // subtrees below them are built by worker threads into local arrays that are appended afterwards (node numbering differs from
    // the sequential build's, the tree does not: rtx_octree_dump / test_octree_matches_oracle_bit_for_bit compare it with the
    // oracle's sequentially inserted one).  Child boxes come from subdivide()'s expressions.
    struct Local {
        std::vector<FNode> nodes;   // [0] = the subtree's root
        std::vector<int> entries;
    };
    // builds the subtree of L.nodes[0] (bbox and depth set) from its ordered prim list
    void build_subtree(Local &L, std::vector<int> list) const {
        struct Item { int node; std::vector<int> list; };
        std::vector<Item> stack;
        stack.push_back(Item{0, std::move(list)});
        while (!stack.empty()) {
            Item it = std::move(stack.back());
            stack.pop_back();
            if (L.nodes[(size_t)it.node].depth <= 0 || it.list.size() < 2) {
                L.nodes[(size_t)it.node].ebegin = (int)L.entries.size();
                L.nodes[(size_t)it.node].ecount = (int)it.list.size();
                L.entries.insert(L.entries.end(), it.list.begin(), it.list.end());
                continue;
            }
            HBox cb[8];
            child_boxes(L.nodes[(size_t)it.node].bbox, cb);
            const int first = (int)L.nodes.size(), d = L.nodes[(size_t)it.node].depth;
            L.nodes[(size_t)it.node].first_child = first;
            for (int c = 0; c < 8; ++c) {
                FNode ch;
                ch.bbox = cb[c];
                ch.depth = d - 1;
                L.nodes.push_back(ch);
            }
            for (int c = 0; c < 8; ++c) {
                std::vector<int> sub;
                for (int i : it.list)
                    if (overlaps(cb[c], prim_box[(size_t)i])) sub.push_back(i);
                stack.push_back(Item{first + c, std::move(sub)});
            }
        }
    }
    void build_parallel(RtxPool &pool) {
        depth = root_depth();
        nodes.clear();
        entries.clear();
        FNode root;
        root.bbox = root_bounds();
        root.depth = depth;
        nodes.push_back(root);
        std::vector<int> all(prim_box.size());
        for (size_t i = 0; i < all.size(); ++i) all[i] = (int)i;
        auto make_leaf = [&](int ni, const std::vector<int> &list) {
            nodes[(size_t)ni].ebegin = (int)entries.size();
            nodes[(size_t)ni].ecount = (int)list.size();
            entries.insert(entries.end(), list.begin(), list.end());
        };
        if (depth <= 0 || all.size() < 2) {
            make_leaf(0, all);
            return;
        }
        // level 1: the root's eight children and their lists
        HBox cb[8];
        child_boxes(nodes[0].bbox, cb);
        nodes[0].first_child = 1;
        for (int c = 0; c < 8; ++c) {
            FNode ch;
            ch.bbox = cb[c];
            ch.depth = depth - 1;
            nodes.push_back(ch);
        }
        std::vector<std::vector<int>> lists(8);
        pool.run(8, [&](size_t c) {
            for (int i : all)
                if (overlaps(cb[c], prim_box[(size_t)i])) lists[c].push_back(i);
        });
        // level 2 where a child is interior: its eight children are the roots of the subtrees the workers build
        struct Sub { int node; int parent_list; };
        std::vector<Sub> subs;
        for (int c = 0; c < 8; ++c) {
            if (nodes[(size_t)(1 + c)].depth <= 0 || lists[(size_t)c].size() < 2) {
                make_leaf(1 + c, lists[(size_t)c]);
                continue;
            }
            HBox gb[8];
            child_boxes(nodes[(size_t)(1 + c)].bbox, gb);
            const int first = (int)nodes.size(), d = nodes[(size_t)(1 + c)].depth;
            nodes[(size_t)(1 + c)].first_child = first;
            for (int g = 0; g < 8; ++g) {
                FNode gn;
                gn.bbox = gb[g];
                gn.depth = d - 1;
                nodes.push_back(gn);
                subs.push_back(Sub{first + g, c});
            }
        }
        std::vector<Local> local(subs.size());
        pool.run(subs.size(), [&](size_t t) {
            const Sub &sb = subs[t];
            std::vector<int> list;
            for (int i : lists[(size_t)sb.parent_list])
                if (overlaps(nodes[(size_t)sb.node].bbox, prim_box[(size_t)i])) list.push_back(i);
            local[t].nodes.push_back(nodes[(size_t)sb.node]);
            build_subtree(local[t], std::move(list));
        });
        // append: local node j > 0 of subtree t becomes node_base[t] + j - 1, local node 0 is nodes[subs[t].node]; offsets first, copies in parallel
        std::vector<size_t> node_base(subs.size()), entry_base(subs.size());
        size_t nn = nodes.size(), ne = entries.size();
        for (size_t t = 0; t < subs.size(); ++t) {
            node_base[t] = nn; entry_base[t] = ne;
            nn += local[t].nodes.size() - 1; ne += local[t].entries.size();
        }
        nodes.resize(nn);
        entries.resize(ne);
        pool.run(subs.size(), [&](size_t t) {
            Local &L = local[t];
            const int shift = (int)node_base[t] - 1;
            for (size_t j = 0; j < L.nodes.size(); ++j) {
                FNode f = L.nodes[j];
                if (f.first_child >= 0) f.first_child += shift;
                f.ebegin += (int)entry_base[t];
                nodes[j == 0 ? (size_t)subs[t].node : node_base[t] + j - 1] = f;
            }
            std::copy(L.entries.begin(), L.entries.end(), entries.begin() + (long)entry_base[t]);
        });
    }
};

// worker threads of the host-side builds (octree subtrees, SAH box lists): the process's share of the box, 16 at most
unsigned build_threads() {
    if (const char *e = std::getenv("RTX_BUILD_THREADS")) return (unsigned)std::max(0, std::atoi(e));
    unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc ? hc : 1u));
}

// Flattened, validated view of an rtx_scene_desc with prims split by finiteness.
struct Flat {
    std::vector<int> fin_kind, fin_index;     // finite prims in insertion order
    std::vector<int> inf_index;               // planes in insertion order
    HostOctree oct;
};

void build_octree(Flat &fl, RtxPool *pool) {
    // pool == NULL (RTX_BUILD_THREADS=0): the sequential insertion of octree.rs, statement for statement (same tree; the cross-check)
    if (!pool) fl.oct.build();
    else fl.oct.build_parallel(*pool);
}

// validation + the prims split by finiteness, with their boxes; the octree is built by build_octree()
rtx_status flatten(const rtx_scene_desc *d, Flat &fl) {
    if (!d) return fail(RTX_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (d->n_spheres < 0 || d->n_planes < 0 || d->n_triangles < 0 || d->n_materials < 0 || d->n_textures < 0 ||
        d->n_images < 0 || d->n_lights < 0)
        return fail(RTX_ERR_INVALID_ARGUMENT, "negative count in rtx_scene_desc");
    long long total = (long long)d->n_spheres + d->n_planes + d->n_triangles;
    if ((d->order_kind == nullptr) != (d->order_index == nullptr))
        return fail(RTX_ERR_INVALID_ARGUMENT, "order_kind and order_index must both be given or both be NULL");
    if (d->order_kind && d->n_prims != total) return fail(RTX_ERR_INVALID_ARGUMENT, "n_prims != n_spheres + n_planes + n_triangles");
    if ((d->n_spheres && (!d->sphere_center || !d->sphere_radius || !d->sphere_material)) ||
        (d->n_planes && (!d->plane_abcd || !d->plane_material)) ||
        (d->n_triangles && (!d->tri_vertices || !d->tri_normals || !d->tri_uv || !d->tri_material)) ||
        (d->n_materials && !d->materials) || (d->n_textures && !d->textures) || (d->n_images && !d->images) ||
        (d->n_lights && !d->lights))
        return fail(RTX_ERR_INVALID_ARGUMENT, "NULL array with a non-zero count");
    if (d->n_lights > RTX_MAX_LIGHTS) return fail(RTX_ERR_LIMIT, "more than 16 lights");
    auto mat_ok = [&](int m) { return m >= 0 && m < d->n_materials; };
    for (int i = 0; i < d->n_spheres; ++i) if (!mat_ok(d->sphere_material[i])) return fail(RTX_ERR_INVALID_ARGUMENT, "sphere material out of range");
    for (int i = 0; i < d->n_planes; ++i) if (!mat_ok(d->plane_material[i])) return fail(RTX_ERR_INVALID_ARGUMENT, "plane material out of range");
    for (int i = 0; i < d->n_triangles; ++i) if (!mat_ok(d->tri_material[i])) return fail(RTX_ERR_INVALID_ARGUMENT, "triangle material out of range");
    for (int i = 0; i < d->n_materials; ++i) {
        const rtx_material &m = d->materials[i];
        if (m.kind < RTX_MAT_COOK_TORRANCE || m.kind > RTX_MAT_FLAT) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown material kind");
        if (m.texture >= d->n_textures) return fail(RTX_ERR_INVALID_ARGUMENT, "material texture out of range");
    }
    for (int i = 0; i < d->n_textures; ++i) {
        const rtx_texture &t = d->textures[i];
        if (t.kind < RTX_TEX_CHECKER || t.kind > RTX_TEX_IMAGE) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown texture kind");
        if (t.kind == RTX_TEX_IMAGE && (t.image < 0 || t.image >= d->n_images)) return fail(RTX_ERR_INVALID_ARGUMENT, "texture image out of range");
    }
    for (int i = 0; i < d->n_images; ++i)
        if (d->images[i].width <= 0 || d->images[i].height <= 0 || !d->images[i].rgba) return fail(RTX_ERR_INVALID_ARGUMENT, "bad image");
    if (d->skybox_image[0] >= 0)
        for (int f = 0; f < 6; ++f)
            if (d->skybox_image[f] < 0 || d->skybox_image[f] >= d->n_images) return fail(RTX_ERR_INVALID_ARGUMENT, "skybox image out of range");

    std::vector<uint8_t> seen_s((size_t)d->n_spheres, 0), seen_p((size_t)d->n_planes, 0), seen_t((size_t)d->n_triangles, 0);
    fl.fin_kind.reserve((size_t)d->n_spheres + (size_t)d->n_triangles);
    fl.fin_index.reserve((size_t)d->n_spheres + (size_t)d->n_triangles);
    fl.oct.prim_box.reserve((size_t)d->n_spheres + (size_t)d->n_triangles);
    for (long long k = 0; k < total; ++k) {
        int kind, idx;
        if (d->order_kind) { kind = d->order_kind[k]; idx = d->order_index[k]; }
        else if (k < d->n_spheres) { kind = RTX_PRIM_SPHERE; idx = (int)k; }
        else if (k < (long long)d->n_spheres + d->n_triangles) { kind = RTX_PRIM_TRIANGLE; idx = (int)(k - d->n_spheres); }
        else { kind = RTX_PRIM_PLANE; idx = (int)(k - d->n_spheres - d->n_triangles); }
        std::vector<uint8_t> *seen = kind == RTX_PRIM_SPHERE ? &seen_s : kind == RTX_PRIM_PLANE ? &seen_p : kind == RTX_PRIM_TRIANGLE ? &seen_t : nullptr;
        if (!seen || idx < 0 || (size_t)idx >= seen->size() || (*seen)[(size_t)idx]) return fail(RTX_ERR_INVALID_ARGUMENT, "order_kind/order_index is not a permutation of the prims");
        (*seen)[(size_t)idx] = 1;
        if (kind == RTX_PRIM_PLANE) {                              // plane.rs:19-23: no bounding box
            fl.inf_index.push_back(idx);
            continue;
        }
        HBox b;
        if (kind == RTX_PRIM_SPHERE) {                             // sphere.rs:18-33
            const double *c = d->sphere_center + (size_t)idx * 3;
            double r = d->sphere_radius[idx];
            for (int a = 0; a < 3; ++a) { b.mn[a] = c[a] - r; b.mx[a] = c[a] + r; }
        } else {                                                   // triangle.rs:102-106
            const double *v = d->tri_vertices + (size_t)idx * 9;
            for (int a = 0; a < 3; ++a) {
                b.mn[a] = std::fmin(std::fmin(v[a], v[3 + a]), v[6 + a]);
                b.mx[a] = std::fmax(std::fmax(v[a], v[3 + a]), v[6 + a]);
            }
        }
        fl.fin_kind.push_back(kind);
        fl.fin_index.push_back(idx);
        fl.oct.prim_box.push_back(b);
    }
    return RTX_OK;
}

// ---------------------------------------------------------------------------- workspace
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;
};

struct Work {
    DPool pool{};
    DShadowQ sq{};
    DShadowQ sq2{};           // second shadow queue: the shadow launches of one step overlap the next step's closest-hit launch
    hipStream_t aux[2] = {nullptr, nullptr};   // the streams the shadow launches run on when they overlap (lights alternate)
    RtxSortBuf sortbuf{};     // the sort of a step's shadow groups by cell (rtx_sort.hip); the queues hold their own cell / order arrays
    uint32_t sort_cap = 0;
    DCounters *d_counters = nullptr;
    DCounters *h_up = nullptr, *h_back = nullptr;   // pinned
    uint32_t *h_step = nullptr, *d_step = nullptr;  // pinned, coherent: (pool tail, overflow, sequence number) of the step, stored by k_group_base, polled by the host
    uint32_t step_seq = 0;
    unsigned long long (*d_ray_slots)[RTX_RAY_SLOT_WORDS] = nullptr;
    unsigned long long (*h_ray_slots)[RTX_RAY_SLOT_WORDS] = nullptr; // pinned
    double *hit_t = nullptr;
    int32_t *hit_ref = nullptr;
    uint32_t *group_base = nullptr;   // workgroup counts and bases of the compact shadow queue (rtxk_trace)
    uint32_t hit_cap = 0;
    double *acc = nullptr;
    size_t acc_cap = 0;       // samples
    uint8_t *out_rgba = nullptr;
    int32_t *out_ids = nullptr;
    size_t out_pixels = 0;
    std::vector<hipEvent_t> events;
};

template <class T>
rtx_status dev_alloc(T **p, size_t n) {
    HIPCHECK(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return RTX_OK;
}

} // namespace

struct rtx_scene {
    int device = 0;
    DScene d{};
    std::vector<void *> allocs;
    rtx_scene_info info{};
    std::vector<rtx_light> lights;
    bool any_reflective = false, any_refractive = false, any_glossy = false;
    int n_materials = 0, n_textures = 0;
    int64_t n_wide_nodes = 0, n_bvh4_nodes = 0;   // sizes of d.wide / d.bvh4 as built (rtx_hierarchy_dump)
    const DWideChild *wide_built = nullptr;       // (d.wide / d.bvh4 are cleared when a walk cannot use them; the dump shows what was built)
    const DNode4 *bvh4_built = nullptr;
    // what the last render call launched (rtx_last_kernel_names)
    RtxTraceNames last_trace;
    const char *last_shadow[RTX_MAX_LIGHTS] = {};
    // k_shadow_coop<false, true> (wide groups handed to the per-lane kernel) earns its keep only in scenes whose frames have hit points
    // OUTSIDE the scene's cube, i.e. on planes; it costs 1.4 % where it finds none (sponza: a ground plane nobody sees beyond the
    // building).  A frame that ran it and handed over next to nothing switches the handle to the plain instance for the next
    // RTX_COOP_PROBE_EVERY - 1 frames; then one frame probes again (a camera that has moved may see the plane).
    uint32_t coop_plain_frames = 0;
    Work w;
};

namespace {

// Uploads on a thread of their own, so that the first host-to-device copies of a process (which pay the runtime's one-off set-up:
// measured 150 ms for the first 100 MB, 8 ms for the same bytes in a second scene) and the copies themselves run beside the host-side
// builds instead of after them.  submit() queues one array (the source must stay alive until finish()); *out is valid after finish().
class Uploader {
public:
    explicit Uploader(int device) : device_(device), th_([this]() { loop(); }) {}
    ~Uploader() { (void)finish(nullptr); }
    template <class T>
    void submit(const T *src, size_t n, const T **out) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(Job{src, n * sizeof(T), (const void **)out});
        }
        cv_.notify_one();
    }
    template <class T>
    void submit(const std::vector<T> &v, const T **out) { submit(v.data(), v.size(), out); }
    // waits for the queue to drain; hands the device allocations to the scene (also on failure, so that the caller's cleanup frees them)
    rtx_status finish(rtx_scene *s) {
        if (th_.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                closing_ = true;
            }
            cv_.notify_one();
            th_.join();
        }
        if (s) {
            for (void *p : allocs_) s->allocs.push_back(p);
            allocs_.clear();
            s->info.device_bytes += bytes_;
            bytes_ = 0;
        } else {
            for (void *p : allocs_) hipFree(p);
            allocs_.clear();
        }
        if (st_ != RTX_OK) return fail(st_, err_);
        return RTX_OK;
    }

private:
    struct Job { const void *src; size_t bytes; const void **out; };
    void loop() {
        hipError_t e = hipSetDevice(device_);
        if (e == hipSuccess) {   // a process's first host-to-device copy pays ~140 ms of runtime set-up whatever its size (tools/exp/h2d_cold.cpp):
            void *w = nullptr;   // pay it here, at once, beside the host-side builds
            char z[64] = {0};
            if (hipMalloc(&w, sizeof(z)) == hipSuccess) {
                (void)hipMemcpy(w, z, sizeof(z), hipMemcpyHostToDevice);
                (void)hipFree(w);
            }
            // set-up for rtx_render_device_noalloc: this device's address of the static scratch, resolved once with the device
            // current (it loads the code object; a failure is kept for that entry point to report, nothing else needs the scratch)
            int herr = 0;
            (void)rtxk_static_scratch(device_, &herr);
        }
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return closing_ || !q_.empty(); });
                if (q_.empty()) return;
                j = q_.front();
                q_.pop_front();
            }
            if (st_ != RTX_OK) continue;   // after a failure the remaining jobs are dropped
            void *p = nullptr;
            const size_t bytes = std::max<size_t>(j.bytes, 1);
            if (e == hipSuccess) e = hipMalloc(&p, bytes);
            if (e == hipSuccess) {
                allocs_.push_back(p);
                bytes_ += (int64_t)bytes;
                e = j.bytes ? hipMemcpy(p, j.src, j.bytes, hipMemcpyHostToDevice) : hipMemset(p, 0, bytes);
            }
            if (e != hipSuccess) {
                st_ = e == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP;
                err_ = std::string("scene upload: ") + hipGetErrorString(e);
                continue;
            }
            *j.out = p;
        }
    }
    int device_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job> q_;
    bool closing_ = false;
    rtx_status st_ = RTX_OK;
    std::string err_;
    std::vector<void *> allocs_;
    int64_t bytes_ = 0;
    std::thread th_;   // last: started when everything above exists
};

void free_work(Work &w) {
    if (w.pool.rec) hipFree(w.pool.rec);
    if (w.pool.thr) hipFree(w.pool.thr);
    if (w.pool.flags) hipFree(w.pool.flags);
    if (w.pool.cell) hipFree(w.pool.cell);
    if (w.sortbuf.ray_order) hipFree(w.sortbuf.ray_order);
    if (w.sq.f) hipFree(w.sq.f);
    if (w.sq.key) hipFree(w.sq.key);
    if (w.sq.sample) hipFree(w.sq.sample);
    if (w.sq.retry) hipFree(w.sq.retry);
    if (w.sq.wide) hipFree(w.sq.wide);
    if (w.sq.count) hipFree(w.sq.count);
    if (w.sq.res) hipFree(w.sq.res);
    if (w.sq.sample0) hipFree(w.sq.sample0);
    if (w.sq.cell) hipFree(w.sq.cell);
    if (w.sq.order) hipFree(w.sq.order);
    if (w.sq2.cell) hipFree(w.sq2.cell);
    if (w.sq2.order) hipFree(w.sq2.order);
    if (w.sortbuf.cell_sorted) hipFree(w.sortbuf.cell_sorted);
    if (w.sortbuf.tmp) hipFree(w.sortbuf.tmp);
    if (w.sq2.f) hipFree(w.sq2.f);
    if (w.sq2.key) hipFree(w.sq2.key);
    if (w.sq2.sample) hipFree(w.sq2.sample);
    if (w.sq2.retry) hipFree(w.sq2.retry);
    if (w.sq2.wide) hipFree(w.sq2.wide);
    if (w.sq2.count) hipFree(w.sq2.count);
    if (w.sq2.res) hipFree(w.sq2.res);
    if (w.sq2.sample0) hipFree(w.sq2.sample0);
    for (hipStream_t a : w.aux)
        if (a) hipStreamDestroy(a);
    if (w.d_counters) hipFree(w.d_counters);
    if (w.d_ray_slots) hipFree(w.d_ray_slots);
    if (w.hit_t) hipFree(w.hit_t);
    if (w.hit_ref) hipFree(w.hit_ref);
    if (w.group_base) hipFree(w.group_base);
    if (w.h_ray_slots) hipHostFree(w.h_ray_slots);
    if (w.h_up) hipHostFree(w.h_up);
    if (w.h_back) hipHostFree(w.h_back);
    if (w.h_step) hipHostFree(w.h_step);
    if (w.acc) hipFree(w.acc);
    if (w.out_rgba) hipFree(w.out_rgba);
    if (w.out_ids) hipFree(w.out_ids);
    for (hipEvent_t e : w.events) hipEventDestroy(e);
    w = Work();
}

rtx_status ensure_work(rtx_scene *s, uint32_t pool_cap, uint32_t batch, size_t n_samples) {
    Work &w = s->w;
    if (w.pool.cap < pool_cap) {
        if (w.pool.rec) { hipFree(w.pool.rec); hipFree(w.pool.thr); hipFree(w.pool.flags); hipFree(w.pool.cell); w.pool = DPool(); }
        rtx_status st;
        if ((st = dev_alloc(&w.pool.rec, (size_t)pool_cap))) return st;
        if ((st = dev_alloc(&w.pool.thr, (size_t)pool_cap * 3))) return st;
        if ((st = dev_alloc(&w.pool.flags, pool_cap))) return st;
        if ((st = dev_alloc(&w.pool.cell, pool_cap))) return st;
        w.pool.cap = pool_cap;
    }
    size_t nl = std::max<size_t>(s->lights.size(), 1);
    if (w.sq.cap < batch) {
        if (w.sq.f) { hipFree(w.sq.f); hipFree(w.sq.key); hipFree(w.sq.sample); hipFree(w.sq.wide); hipFree(w.sq.res); hipFree(w.sq.sample0); hipFree(w.sq.cell); hipFree(w.sq.order); uint32_t *keep = w.sq.retry, *keep2 = w.sq.count; w.sq = DShadowQ(); w.sq.retry = keep; w.sq.count = keep2; }
        w.sq.retry_cap = RTX_RETRY_CAP;
        rtx_status st;
        if ((st = dev_alloc(&w.sq.f, nl * 6 * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq.key, nl * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq.sample, nl * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq.wide, nl * (1u + (size_t)batch)))) return st;
        if ((st = dev_alloc(&w.sq.res, nl * 3 * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq.sample0, (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq.cell, (size_t)batch)) || (st = dev_alloc(&w.sq.order, 1u + (size_t)batch))) return st;
        if (!w.sq.retry && (st = dev_alloc(&w.sq.retry, nl * (2u + 2u * RTX_RETRY_CAP)))) return st;
        if (!w.sq.count && (st = dev_alloc(&w.sq.count, 16))) return st;
        w.sq.cap = batch;
    }
    if (w.sq2.cap < batch) {
        if (w.sq2.f) { hipFree(w.sq2.f); hipFree(w.sq2.key); hipFree(w.sq2.sample); hipFree(w.sq2.wide); hipFree(w.sq2.res); hipFree(w.sq2.sample0); hipFree(w.sq2.cell); hipFree(w.sq2.order); uint32_t *keep = w.sq2.retry, *keep2 = w.sq2.count; w.sq2 = DShadowQ(); w.sq2.retry = keep; w.sq2.count = keep2; }
        w.sq2.retry_cap = RTX_RETRY_CAP;
        rtx_status st;
        if ((st = dev_alloc(&w.sq2.f, nl * 6 * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq2.key, nl * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq2.sample, nl * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq2.wide, nl * (1u + (size_t)batch)))) return st;
        if ((st = dev_alloc(&w.sq2.res, nl * 3 * (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq2.sample0, (size_t)batch))) return st;
        if ((st = dev_alloc(&w.sq2.cell, (size_t)batch)) || (st = dev_alloc(&w.sq2.order, 1u + (size_t)batch))) return st;
        if (!w.sq2.retry && (st = dev_alloc(&w.sq2.retry, nl * (2u + 2u * RTX_RETRY_CAP)))) return st;
        if (!w.sq2.count && (st = dev_alloc(&w.sq2.count, 16))) return st;
        w.sq2.cap = batch;
    }
    if (w.sort_cap < batch) {
        if (w.sortbuf.cell_sorted) { hipFree(w.sortbuf.cell_sorted); hipFree(w.sortbuf.tmp); hipFree(w.sortbuf.ray_order); w.sortbuf = RtxSortBuf(); w.sort_cap = 0; }
        rtx_status st;
        if ((st = dev_alloc(&w.sortbuf.cell_sorted, (size_t)batch)) || (st = dev_alloc(&w.sortbuf.ray_order, (size_t)batch))) return st;
        w.sortbuf.tmp_bytes = rtxk_sort_temp_bytes(batch);
        HIPCHECK(hipMalloc(&w.sortbuf.tmp, std::max<size_t>(w.sortbuf.tmp_bytes, 256)));
        w.sort_cap = batch;
    }
    for (hipStream_t &a : w.aux)
        if (!a) HIPCHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    if (w.hit_cap < batch) {
        if (w.hit_t) { hipFree(w.hit_t); hipFree(w.hit_ref); hipFree(w.group_base); w.hit_t = nullptr; w.hit_ref = nullptr; w.group_base = nullptr; w.hit_cap = 0; }
        rtx_status st;
        if ((st = dev_alloc(&w.hit_t, batch)) || (st = dev_alloc(&w.hit_ref, batch))) return st;
        if ((st = dev_alloc(&w.group_base, 4 * ((size_t)batch / RTX_BLOCK + 2)))) return st;
        w.hit_cap = batch;
    }
    if (!w.d_counters) {
        rtx_status st;
        if ((st = dev_alloc(&w.d_counters, 1))) return st;
        HIPCHECK(hipHostMalloc((void **)&w.h_up, sizeof(DCounters), hipHostMallocDefault));
        HIPCHECK(hipHostMalloc((void **)&w.h_back, sizeof(DCounters), hipHostMallocDefault));
        HIPCHECK(hipHostMalloc((void **)&w.h_step, 4 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(w.h_step, 0, 4 * sizeof(uint32_t));
        HIPCHECK(hipHostGetDevicePointer((void **)&w.d_step, w.h_step, 0));
        HIPCHECK(hipMalloc((void **)&w.d_ray_slots, RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long)));
        HIPCHECK(hipHostMalloc((void **)&w.h_ray_slots, RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    }
    if (w.acc_cap < n_samples) {
        if (w.acc) { hipFree(w.acc); w.acc = nullptr; w.acc_cap = 0; }
        rtx_status st;
        if ((st = dev_alloc(&w.acc, n_samples * 3))) return st;
        w.acc_cap = n_samples;
    }
    return RTX_OK;
}

// Camera::new -> update_eye_vector / update_internal_sizes (scene/camera.rs:26-50, 88-105)
void fill_camera(DFrame &fr, const rtx_camera &c) {
    const double PI = 3.14159265358979323846264338327950288;
    double e[3] = {c.look_at[0] - c.position[0], c.look_at[1] - c.position[1], c.look_at[2] - c.position[2]};
    double l = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    for (int a = 0; a < 3; ++a) {
        fr.pos[a] = c.position[a];
        fr.up[a] = c.up[a];
        fr.eye[a] = e[a] / l;
    }
    fr.right[0] = fr.eye[1] * fr.up[2] - fr.eye[2] * fr.up[1];     // eye.cross(up), NOT normalised
    fr.right[1] = fr.eye[2] * fr.up[0] - fr.eye[0] * fr.up[2];
    fr.right[2] = fr.eye[0] * fr.up[1] - fr.eye[1] * fr.up[0];
    double fov_rad = c.fov_deg * (PI / 180.0);
    double ratio = (double)c.image_height / (double)c.image_width;
    fr.half_width = std::tan(fov_rad);                             // the full angle, camera.rs:97
    fr.half_height = fr.half_width * ratio;
    double camera_width = fr.half_width * 2.0, camera_height = fr.half_height * 2.0;
    fr.pixel_width = camera_width / (double)(c.image_width - 1);
    fr.pixel_height = camera_height / (double)(c.image_height - 1);
}

struct Seg {
    uint32_t begin, end;
    uint32_t gen;   // generation of its rays: 0 = primary, g = children of generation g - 1; a ray of generation `levels` has used up its depth
};

// The device workspace of one render call: the scene handle's own buffers (ensure_work) or a carving of the static device
// scratch (rtx_render_device_noalloc).
struct WorkView {
    DPool pool{};
    DShadowQ sq{}, sq2{};
    DCounters *d_counters = nullptr;
    unsigned long long (*d_ray_slots)[RTX_RAY_SLOT_WORDS] = nullptr;
    double *hit_t = nullptr;
    int32_t *hit_ref = nullptr;
    uint32_t *group_base = nullptr;
    double *acc = nullptr;
};

// Ray pool sizing.  The pool is a stack of segments processed depth-first (deepest generation first, at most `batch` rays per
// launch), so generation g holds at most batch * fanout rays and the live total is bounded by batch * (1 + fanout * levels),
// levels = longest child chain.
struct PoolNeed {
    uint32_t fanout = 0, levels = 0;
    uint64_t per_batch = 1;
};
PoolNeed pool_need(const rtx_scene *s, const rtx_options *opt);

#define RTX_STATIC_MAX_SEGS 64   // segment stack of the allocation-free entry point (levels + 2 are needed)

// Carves the static scratch into the workspace of one call; picks the batch (<= want_batch, a multiple of RTX_BLOCK) that fits.
rtx_status carve_static(uint8_t *base, size_t n_lights, uint32_t n_samples, const PoolNeed &need, uint32_t want_batch,
                        WorkView &v, uint32_t *batch_out, uint32_t *pool_cap_out) {
    const size_t nl = std::max<size_t>(n_lights, 1);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t fixed = up(sizeof(DCounters)) + up(RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long)) +
                         up((size_t)std::max<uint32_t>(n_samples, 1) * 3 * sizeof(double)) + 23 * 256 /* alignment of the arrays below */;
    // per ray of a batch: hit record 12 (+ 16 per workgroup for the queue and pool slots), pool (per_batch rays of 92 B), one shadow queue entry per light 60, one retry pair per light 8,
    // one entry per light in the list of groups left to the per-lane shadow kernel 4
    const size_t pool_per = std::max<uint64_t>(need.per_batch, 1 + (uint64_t)need.fanout);
    const size_t per = 12 + 1 + pool_per * 92 + nl * 60 + nl * 8 + nl * 4 + nl * 24 + 4;   // (+ the lights' deferred results and the clean sample plane)
    if (fixed + per * RTX_BLOCK + nl * 8 + nl * 256 > RTX_STATIC_SCRATCH_BYTES)
        return fail(RTX_ERR_LIMIT, "rtx_render_device_noalloc: this call needs " + std::to_string(fixed + per * RTX_BLOCK) +
                                       " bytes of workspace, the static device scratch holds " + std::to_string(RTX_STATIC_SCRATCH_BYTES) +
                                       " (render fewer samples per call: shard_index / shard_count, or use rtx_render_device)");
    uint64_t batch = (RTX_STATIC_SCRATCH_BYTES - fixed - nl * 8 - nl * 256) / per;
    batch = std::min<uint64_t>(batch, want_batch ? want_batch : (1u << 20));
    batch = std::max<uint64_t>(RTX_BLOCK, batch / RTX_BLOCK * RTX_BLOCK);
    const uint64_t pool_cap = batch * pool_per;
    uint8_t *p = base;
    auto take = [&](size_t bytes) { uint8_t *q = p; p += up(bytes); return q; };
    v.d_counters = (DCounters *)take(sizeof(DCounters));
    v.d_ray_slots = (unsigned long long (*)[RTX_RAY_SLOT_WORDS])take(RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long));
    v.acc = (double *)take((size_t)std::max<uint32_t>(n_samples, 1) * 3 * sizeof(double));
    v.hit_t = (double *)take(batch * sizeof(double));
    v.hit_ref = (int32_t *)take(batch * sizeof(int32_t));
    v.group_base = (uint32_t *)take(4 * (batch / RTX_BLOCK + 2) * sizeof(uint32_t));
    v.pool.rec = (DRayRec *)take(pool_cap * sizeof(DRayRec));
    v.pool.thr = (double *)take(pool_cap * 3 * sizeof(double));
    v.pool.flags = (uint32_t *)take(pool_cap * sizeof(uint32_t));
    v.pool.cap = (uint32_t)pool_cap;
    v.sq.f = (double *)take(nl * 6 * batch * sizeof(double));
    v.sq.key = (uint64_t *)take(nl * batch * sizeof(uint64_t));
    v.sq.sample = (uint32_t *)take(nl * batch * sizeof(uint32_t));
    v.sq.retry_cap = (uint32_t)batch;
    v.sq.retry = (uint32_t *)take(nl * (2 + 2 * batch) * sizeof(uint32_t));
    v.sq.wide = (uint32_t *)take(nl * (1 + batch) * sizeof(uint32_t));
    v.sq.res = (double *)take(nl * 3 * batch * sizeof(double));
    v.sq.sample0 = (uint32_t *)take(batch * sizeof(uint32_t));
    v.sq.count = (uint32_t *)take(sizeof(uint32_t));
    v.sq.cap = (uint32_t)batch;
    v.sq2 = v.sq;   // one launch after the other on the caller's stream: one queue is enough
    if ((size_t)(p - base) > RTX_STATIC_SCRATCH_BYTES) return fail(RTX_ERR_LIMIT, "rtx_render_device_noalloc: workspace layout exceeds the static scratch");
    *batch_out = (uint32_t)batch;
    *pool_cap_out = (uint32_t)pool_cap;
    return RTX_OK;
}

PoolNeed pool_need(const rtx_scene *s, const rtx_options *opt) {
    PoolNeed n;
    if (s->any_reflective) {
        n.fanout += s->any_glossy ? std::max<uint32_t>(1, opt->gloss_samples) : 1;
        n.levels += opt->reflect_depth > 1 ? opt->reflect_depth - 1 : 0;
    }
    if (s->any_refractive) {
        n.fanout += 1;
        n.levels += opt->refract_depth > 1 ? opt->refract_depth - 1 : 0;
    }
    if (n.levels == 0) n.fanout = 0;
    n.per_batch = 1 + (uint64_t)n.fanout * n.levels;
    return n;
}

#define RTX_COOP_PROBE_EVERY 16
constexpr rtx_status RTX_INTERNAL_RETRY_OVERFLOW = -1;   // render_once -> render_impl: render the frame again with exact shadows

// One frame.  `noalloc`: the allocation-free variant (rtx_render_device_noalloc): the workspace is the static device scratch, every
// launch runs on the caller's stream, no events, no pinned host buffers, no host heap.
rtx_status render_once(rtx_scene *s, const rtx_camera *cam, const rtx_options *opt, uint32_t extra_flags, uint8_t *d_rgba,
                       int32_t *d_ids, hipStream_t stream, rtx_stats *stats, bool noalloc) {
    auto wall0 = std::chrono::steady_clock::now();
    uint32_t shard_count = opt->shard_count ? opt->shard_count : 1;

    DFrame fr{};
    fill_camera(fr, *cam);
    fr.W = cam->image_width; fr.H = cam->image_height;
    fr.ps = opt->pixel_samples; fr.spp = opt->pixel_samples * opt->pixel_samples;
    fr.inv_ps = 1.0 / (double)opt->pixel_samples;
    fr.inv_spp = 1.0 / (double)(opt->pixel_samples * opt->pixel_samples);
    fr.seed = opt->seed;
    fr.tiles_x = (fr.W + 7) / 8;
    fr.shard_index = opt->shard_index; fr.shard_count = shard_count;
    uint32_t n_blocks = fr.H / 8;
    fr.n_blocks_local = opt->shard_index < n_blocks ? (n_blocks - opt->shard_index + shard_count - 1) / shard_count : 0;
    fr.reflect_depth = opt->reflect_depth; fr.refract_depth = opt->refract_depth;
    fr.shadow_samples = opt->shadow_samples; fr.gloss_samples = opt->gloss_samples;
    fr.rng_mode = opt->rng_mode; fr.frame = opt->frame;
    fr.flags = (opt->flags & RTX_FLAG_PUBLIC_MASK) | extra_flags | (noalloc ? (uint32_t)RTX_FLAG_SERIAL : 0u);

    uint64_t n_samples64 = (uint64_t)fr.n_blocks_local * fr.tiles_x * 64ull * fr.spp;
    if (n_samples64 >= (1ull << 32)) return fail(RTX_ERR_LIMIT, "more than 2^32 samples in one call");
    uint32_t n_samples = (uint32_t)n_samples64;

    const PoolNeed need = pool_need(s, opt);
    const uint32_t fanout = need.fanout;
    uint32_t batch = opt->batch_rays, pool_cap = 0;
    WorkView v;
    Work &w = s->w;
    rtx_status st;
    if (noalloc) {
        if (need.levels + 2 > RTX_STATIC_MAX_SEGS)
            return fail(RTX_ERR_LIMIT, "rtx_render_device_noalloc: reflect_depth + refract_depth above 62");
        int herr = 0;
        uint8_t *base = (uint8_t *)rtxk_static_scratch(s->device, &herr);
        if (!base) return fail(RTX_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString((hipError_t)herr));
        if ((st = carve_static(base, s->lights.size(), n_samples, need, opt->batch_rays, v, &batch, &pool_cap))) return st;
    } else {
        if (!batch) {   // default: 8 Mi rays per launch (measured: bunny 10.9 -> 9.9 ms against 4 Mi, sponza within 1 %; heptoroid depth 10:
                        // 664 -> 632 ms against 1.8 Mi), shrunk so that the pool (92 B per ray) stays within a quarter of the device memory
                        // that is free now (the pool this handle already holds counts as free) and never below 64 Mi rays (5.6 GB)
            size_t free_b = 0, total_b = 0;
            HIPCHECK(hipMemGetInfo(&free_b, &total_b));
            const uint64_t room = ((uint64_t)free_b + (uint64_t)w.pool.cap * 92u) / 4u / 92u;
            const uint64_t cap_rays = std::min<uint64_t>(std::max<uint64_t>(room, 64ull << 20), (1ull << 32) - 1);
            uint64_t b = cap_rays / need.per_batch;
            batch = (uint32_t)std::min<uint64_t>(8u << 20, std::max<uint64_t>(b / RTX_BLOCK * RTX_BLOCK, 256u << 10));
        }
        batch = std::max<uint32_t>(batch, RTX_BLOCK);
        uint64_t pool_cap64 = opt->pool_rays;
        if (!pool_cap64) {
            // generation 0 holds at most `batch` rays, generation g > 0 at most batch * fanout (the children of one launch), and no
            // generation more than every sample's descendants of that depth: a small frame does not reserve a deep pool
            pool_cap64 = std::min<uint64_t>(batch, n_samples);
            uint64_t gen = n_samples;
            for (uint32_t g = 1; g <= need.levels; ++g) {
                gen = std::min<uint64_t>(gen * fanout, (uint64_t)batch * fanout);
                pool_cap64 += gen;
            }
        }
        pool_cap64 = std::max<uint64_t>(pool_cap64, (uint64_t)std::min<uint64_t>(batch, std::max<uint32_t>(n_samples, 1u)) * (1 + fanout));
        pool_cap64 = std::max<uint64_t>(pool_cap64, RTX_BLOCK);
        if (pool_cap64 >= (1ull << 32)) return fail(RTX_ERR_LIMIT, "ray pool above 2^32 rays");
        pool_cap = (uint32_t)pool_cap64;
        if ((st = ensure_work(s, pool_cap, batch, n_samples))) return st;
        v.pool = w.pool; v.sq = w.sq; v.sq2 = w.sq2; v.d_counters = w.d_counters; v.d_ray_slots = w.d_ray_slots;
        v.hit_t = w.hit_t; v.hit_ref = w.hit_ref; v.acc = w.acc; v.group_base = w.group_base;
    }

    size_t ev_used = 0;
    auto next_event = [&](hipEvent_t *e) -> rtx_status {
        if (ev_used == w.events.size()) {
            hipEvent_t ne;
            HIPCHECK(hipEventCreate(&ne));
            w.events.push_back(ne);
        }
        *e = w.events[ev_used++];
        return RTX_OK;
    };
    struct Span { hipEvent_t a, b; int kind; hipEvent_t mid; };   // kind 0: closest-hit + shading chain (mid: between the two), 1 + light: a shadow launch
    std::vector<Span> spans;
    const bool timing = stats != nullptr && !noalloc;

    // the depth test at the top of Renderer::trace (renderer.rs:106): no rays at all
    bool trace_anything = opt->reflect_depth > 0 && opt->refract_depth > 0 && fr.spp > 0;
    // the any-hit shadow kernels and the two-phase shadow product keep a retry list per launch (rtxk_shadow); its overflow flag is read
    // once, after the last launch
    const bool product_bvh = !s->d.all_opaque && s->d.bvh4 && s->d.prim_leaves && !(fr.flags & (RTX_FLAG_OCTREE_WALK | RTX_FLAG_REFERENCE_ORDER));
    const bool uses_retry = trace_anything && opt->shadow_samples > 0 && !s->lights.empty() && (s->d.all_opaque || product_bvh) &&
                            !(fr.flags & RTX_FLAG_INTERNAL_EXACT_SHADOWS);

    if (n_samples) HIPCHECK(hipMemsetAsync(v.acc, 0, (size_t)n_samples * 3 * sizeof(double), stream));
    if (!trace_anything && d_ids && fr.n_blocks_local) {
        // renderer.rs:106: with a depth counter at 0 trace() returns black before it looks for a hit: no primary ray is traced, every pixel
        // of the call's blocks is a miss (-1).  (k_shade writes the ids otherwise; found by tools/exp/fuzz_campaign2.py: they were left as they were)
        const size_t block_ints = (size_t)8 * fr.W;
        if ((fr.flags & RTX_FLAG_COMPACT_OUTPUT) || shard_count == 1) {
            HIPCHECK(hipMemsetAsync(d_ids, 0xFF, (size_t)fr.n_blocks_local * block_ints * sizeof(int32_t), stream));
        } else {
            for (uint32_t j = 0; j < fr.n_blocks_local; ++j)
                HIPCHECK(hipMemsetAsync(d_ids + (size_t)(opt->shard_index + j * shard_count) * block_ints, 0xFF, block_ints * sizeof(int32_t), stream));
        }
    }
    HIPCHECK(hipMemsetAsync(v.d_ray_slots, 0, RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long), stream));
    if (noalloc) {
        rtxk_init_counters(stream, v.d_counters, v.d_ray_slots);   // no pinned staging buffer in this variant: a one-thread kernel
    } else {
        std::memset(w.h_up, 0, sizeof(DCounters));
        w.h_up->ray_slots = v.d_ray_slots;
        HIPCHECK(hipMemcpyAsync(v.d_counters, w.h_up, sizeof(DCounters), hipMemcpyHostToDevice, stream));
    }
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    if (timing) {
        if ((st = next_event(&ev_first))) return st;
        HIPCHECK(hipEventRecord(ev_first, stream));
    }

    uint32_t launches_closest = 0, launches_shadow = 0;
    uint64_t peak_pool = 0;
    Seg seg_fixed[RTX_STATIC_MAX_SEGS];
    std::vector<Seg> seg_heap;
    size_t n_segs = 0;
    auto seg_push = [&](Seg sg) {
        if (noalloc) seg_fixed[n_segs] = sg;   // bounded by levels + 2 <= RTX_STATIC_MAX_SEGS (checked above)
        else if (n_segs < seg_heap.size()) seg_heap[n_segs] = sg;
        else seg_heap.push_back(sg);
        ++n_segs;
    };
    auto seg_top = [&]() -> Seg & { return noalloc ? seg_fixed[n_segs - 1] : seg_heap[n_segs - 1]; };
    uint32_t s_next = 0;
    // Streams: the closest-hit / shade chain runs on the caller's stream, the shadow launches of a step on
    // w.aux[light % 2], so that they overlap each other and the NEXT step's closest-hit launch (a closest-hit launch ends in a long tail of
    // few waves, a small one is nothing but tail; the shadow kernels only read their queue and add to `acc` with
    // atomics).  Two shadow queues alternate: k_shade of step i waits until the shadow launches of step i - 2 have
    // read theirs.  The caller's stream joins w.aux[] before the resolve.
    const bool overlap = !(fr.flags & RTX_FLAG_SERIAL) && opt->shadow_samples > 0 && !s->lights.empty();
    const bool coop_defer = s->d.n_planes > 0 && s->coop_plain_frames == 0;   // (see rtx_scene::coop_plain_frames)
    // Deferred accumulation (rtx_device.h: DShadowQ.res, k_accumulate) pays where the shadow launches are long and several lights share a ray's
    // sample: measured sponza (2 sphere lights x 16 tries) 184.3 -> 179.6 ms; against it bunny (1 light, 1 try) 4.29 -> 4.52 ms, sibenik
    // (3 lights x 4 tries) 8.83 -> 8.91 ms, heptoroid 440.9 -> 443.4 ms: one more launch per step on a chain that is launch-bound.
    // RTX_DIRECT_ACC=1 / RTX_DEFER_ACC=1 force either (A/B).
    uint32_t n_soft = 0;
    for (const rtx_light &lt : s->lights) n_soft += lt.kind != RTX_LIGHT_POINT ? 1u : 0u;
    const bool direct_acc = std::getenv("RTX_DIRECT_ACC") != nullptr ||
                            (!(n_soft >= 2u && opt->shadow_samples >= 16u) && std::getenv("RTX_DEFER_ACC") == nullptr);
    // Sorted shadow launches (rtx_sort.hip): the steps of the second generation and later whose feelers run in a per-lane kernel (no
    // cooperative walk for that light's try count) and that are large enough to fill the chip many times over.  Measured heptoroid
    // (480x272, depth 10/10): k_shadow<7> 216.7 -> see DESIGN section 11.  RTX_NO_SORT=1 / RTX_SORT_ALL=1 (every step, primaries too): A/B.
    bool per_lane_shadows = false;
    {
        const bool coop_scene = s->d.all_opaque && s->d.wide && !(fr.flags & (RTX_FLAG_NO_COOP | RTX_FLAG_INTERNAL_EXACT_SHADOWS));
        for (const rtx_light &lt : s->lights) {
            const uint32_t tries = lt.kind == RTX_LIGHT_POINT ? 1u : opt->shadow_samples;
            if (!(coop_scene && (tries % 16u == 0u || tries == 4u || tries == 8u))) per_lane_shadows = true;
        }
    }
    const bool sort_all = std::getenv("RTX_SORT_ALL") != nullptr;
    const bool sort_steps = !noalloc && opt->shadow_samples > 0 && per_lane_shadows && !(fr.flags & RTX_FLAG_INTERNAL_EXACT_SHADOWS) &&
                            (!s->d.all_opaque || std::getenv("RTX_SORT_OPAQUE") != nullptr) && std::getenv("RTX_NO_SORT") == nullptr &&
                            std::getenv("RTX_NO_GROUP_SORT") == nullptr;
    // The same for the closest-hit launch of those scenes (DPool.cell, rtxk_sort_rays): RTX_NO_RAY_SORT=1 for A/B.
    const bool sort_rays_frame = !noalloc && (!s->d.all_opaque || std::getenv("RTX_SORT_OPAQUE") != nullptr) &&
                                 std::getenv("RTX_NO_RAY_SORT") == nullptr && std::getenv("RTX_NO_SORT") == nullptr;
    if (!sort_rays_frame) v.pool.cell = nullptr;   // (k_shade then writes no cells)
    uint32_t sort_min = 1u << 16, sort_gen = 1u;
    if (const char *e = std::getenv("RTX_SORT_MIN")) sort_min = (uint32_t)std::atoi(e);
    if (const char *e = std::getenv("RTX_SORT_GEN")) sort_gen = (uint32_t)std::atoi(e);
    bool coop_defer_ran = false;
    // k_shade's small instance (96 VGPRs, spills) for frames whose shadow feelers run in k_shadow_coop<false> beside the next step's closest-hit /
    // shade launches: it paid while every workgroup of k_shade waited for a returning atomic on the pool's tail (sponza 208.6 -> 204.6 ms);
    // without that atomic the general instance is the faster one there too (185.3 against 191.5 ms).  Kept behind RTX_LEAN_SHADE=1 for A/B.
    if (overlap && s->d.all_opaque && s->d.wide && !(fr.flags & RTX_FLAG_NO_COOP) && opt->shadow_samples % 16u == 0u && !(fr.flags & RTX_FLAG_INTERNAL_EXACT_SHADOWS) && std::getenv("RTX_LEAN_SHADE"))
        fr.flags |= RTX_FLAG_INTERNAL_SMALL_SHADE;
    hipEvent_t sq_free[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [queue][aux stream]
    size_t n_aux = std::min<size_t>(2, s->lights.size());
    if (const char *e = std::getenv("RTX_AUX_STREAMS")) n_aux = std::max<size_t>(1, std::min<size_t>(n_aux, (size_t)std::atoi(e)));   // A/B: 1 = the lights' launches one after the other
    uint32_t step = 0;
    // Every failure after the first launch leaves through here: nothing keeps running on the library's own streams (they read
    // the shadow queues and add to `acc`, which the caller may free or render into again as soon as this call has returned).
    auto bail = [&](rtx_status e) {
        if (overlap)
            for (hipStream_t a : w.aux) hipStreamSynchronize(a);
        hipStreamSynchronize(stream);
        return e;
    };
    uint32_t back_tail = 0, back_overflow = 0;
    // The host sizes the next step from this step's children.  k_group_base knows their number BEFORE k_shade runs and stores it (with the
    // overflow flag and the step's sequence number) into pinned host memory; the host polls for the number instead of waiting for a copy
    // and an event behind k_shade, and enqueues the next closest-hit launch while this step still shades: no idle chip between steps
    // (measured: sphere 0.198 -> 0.183 ms, bunny 3.95 -> 3.87 ms, sibenik and sponza unchanged).  RTX_NO_POLL=1: the copy and the event (A/B).  Every 4096 looks the stream is asked whether it
    // has finished or failed without the number having arrived; then the counters are fetched by a copy, as before.
    const bool poll_step = !noalloc && w.d_step != nullptr && std::getenv("RTX_NO_POLL") == nullptr;
    auto wait_step = [&](uint32_t seq, uint32_t *tail, uint32_t *ovf) -> rtx_status {
        uint32_t looks = 0;
        while (__atomic_load_n(&w.h_step[2], __ATOMIC_ACQUIRE) != seq) {
            if ((++looks & 0xFFFu) != 0u) continue;
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return fail(RTX_ERR_HIP, std::string("wavefront step: ") + hipGetErrorString(q));
            if (__atomic_load_n(&w.h_step[2], __ATOMIC_ACQUIRE) == seq) break;
            HIPCHECK(hipMemcpy(w.h_back, v.d_counters, offsetof(DCounters, ray_slots), hipMemcpyDeviceToHost));
            *tail = w.h_back->pool_tail; *ovf = w.h_back->overflow;
            return RTX_OK;
        }
        *tail = w.h_step[0]; *ovf = w.h_step[1];
        return RTX_OK;
    };
    const bool use_generate = std::getenv("RTX_GENERATE") != nullptr;   // A/B: k_generate stores the primaries' records, as until the end of round 3
    uint32_t seg0_s0 = 0;   // first sample of the batch of primaries in the pool
    // one wavefront step; a non-zero status is returned through bail()
    auto one_step = [&]() -> rtx_status {
        if (n_segs == 0) {
            // a batch of primaries: pool slots [0, cnt) stand for the samples s_next .. s_next + cnt; nobody stores their records (the
            // closest-hit and shading kernels compute a primary ray from its sample number: rtx_kernels.hip, primary_ray)
            uint32_t cnt = std::min<uint32_t>(batch, n_samples - s_next);
            if (use_generate) rtxk_generate(stream, fr, v.pool, 0, s_next, cnt);
            seg_push(Seg{0, cnt, 0});
            seg0_s0 = s_next;
            s_next += cnt;
        }
        Seg &top = seg_top();
        uint32_t top_end = top.end;
        uint32_t n = std::min<uint32_t>(batch, top.end - top.begin);
        const uint32_t top_gen = top.gen;
        // room for the children of this launch; the rays of the LAST generation (reflect / refract counters used up) spawn none: without
        // this a frame whose every ray spawns a child at every level (a camera inside a mirror box) fills the pool exactly and found no room
        // for children that cannot exist (tools/exp/fuzz_campaign.py, seeds 5110 and 5118)
        if (fanout > 0 && top_gen < need.levels) {
            uint64_t fit = (uint64_t)(pool_cap - top_end) / fanout;
            if (fit < n) n = (uint32_t)fit;
            if (n == 0) return fail(RTX_ERR_QUEUE_OVERFLOW, "ray pool exhausted; raise rtx_options.pool_rays");
        }
        uint32_t begin = top.end - n;
        // (the pool's tail needs no upload: k_group_base moves it from top_end, which rtxk_trace hands it, past the step's children; the
        // overflow flag was cleared with the frame's counters and ends the frame the first time it is raised)
        Span sp_c{nullptr, nullptr, 0, nullptr};
        if (timing) {
            if ((st = next_event(&sp_c.a)) || (st = next_event(&sp_c.b)) || (st = next_event(&sp_c.mid))) return st;
            HIPCHECK(hipEventRecord(sp_c.a, stream));
        }
        // (the step's outcome comes back through pinned memory as soon as k_group_base knows it: see wait_step)
        uint32_t step_seq = 0;
        if (poll_step) {
            if (++w.step_seq == 0u) ++w.step_seq;
            step_seq = w.step_seq;
        }
        DShadowQ sq = (overlap && (step & 1u)) ? v.sq2 : v.sq;
        if (direct_acc) sq.res = nullptr;   // A/B: every shadow launch adds its results to the samples itself
        const bool sort_step = sort_steps && (sort_all || (top_gen >= sort_gen && n >= sort_min));
        if (!sort_step) sq.cell = sq.order = nullptr;
        rtxk_trace(stream, s->d, fr, v.pool, begin, n, sq, v.d_counters, v.acc, d_ids, v.hit_t, v.hit_ref,
                   overlap ? (void *)sq_free[step & 1u][0] : nullptr, overlap ? (void *)sq_free[step & 1u][1] : nullptr, (void *)sp_c.mid, v.group_base, top_end, &s->last_trace,
                   (sort_step || sort_rays_frame) ? &w.sortbuf : nullptr, sort_rays_frame && (sort_all || (top_gen >= sort_gen && n >= sort_min)),
                   poll_step ? w.d_step : nullptr, step_seq, (top_gen == 0 && !use_generate) ? seg0_s0 : 0xFFFFFFFFu);
        ++launches_closest;
        if (timing) {
            HIPCHECK(hipEventRecord(sp_c.b, stream));
            spans.push_back(sp_c);
        }
        // the counters travel back while the shadow kernels run: the host only needs them (children
        // spawned, overflow) to size the NEXT closest-hit launch
        hipEvent_t ev_back = nullptr;
        if (!noalloc) {
            if (!poll_step) HIPCHECK(hipMemcpyAsync(w.h_back, v.d_counters, offsetof(DCounters, ray_slots), hipMemcpyDeviceToHost, stream));
            if (overlap) {
                if ((st = next_event(&ev_back))) return st;
                HIPCHECK(hipEventRecord(ev_back, stream));
                for (size_t a = 0; a < n_aux; ++a) HIPCHECK(hipStreamWaitEvent(w.aux[a], ev_back, 0));
            }
        }
        // the shadow queue is compact (one group per ray that hit something, in ray order; its count is on the device): the launches
        // get the grid of all n rays of the step and blocks past the count leave at once
        if (opt->shadow_samples > 0) {
            uint32_t deferred = 0u;   // lights whose launch stores its results for k_accumulate (rtx_device.h: rtx_shadow_defers)
            for (size_t li = 0; li < s->lights.size(); ++li) {
                uint32_t tries = s->lights[li].kind == RTX_LIGHT_POINT ? 1u : opt->shadow_samples;   // renderer.rs:194
                if (sq.res && rtx_shadow_defers(tries)) deferred |= 1u << li;
                hipStream_t sstream = overlap ? w.aux[li % n_aux] : stream;
                Span sp_s{nullptr, nullptr, 1 + (int)li, nullptr};
                if (timing) {   // one event pair per launch, so ms_shadow / launches_shadow is a kernel duration
                    if ((st = next_event(&sp_s.a)) || (st = next_event(&sp_s.b))) return st;
                    HIPCHECK(hipEventRecord(sp_s.a, sstream));
                }
                s->last_shadow[li] = rtxk_shadow(sstream, s->d, fr, sq, (uint32_t)li, n, tries, v.d_counters, v.acc, coop_defer);
                coop_defer_ran = coop_defer_ran || std::strcmp(s->last_shadow[li], "k_shadow_coop<false, true>") == 0;
                ++launches_shadow;
                if (timing) {
                    HIPCHECK(hipEventRecord(sp_s.b, sstream));
                    spans.push_back(sp_s);
                }
            }
            // the step's local lighting goes to the samples once every light has answered: on the first shadow stream, after the second
            if (deferred) {
                hipStream_t astream = overlap ? w.aux[0] : stream;
                if (overlap && n_aux > 1 && s->lights.size() > 1) {
                    hipEvent_t ev_l = nullptr;
                    if ((st = next_event(&ev_l))) return st;
                    HIPCHECK(hipEventRecord(ev_l, w.aux[1]));
                    HIPCHECK(hipStreamWaitEvent(astream, ev_l, 0));
                }
                rtxk_accumulate(astream, sq, (uint32_t)s->lights.size(), deferred, n, v.acc);
            }
        }
        if (noalloc) {
            uint32_t two[2] = {0, 0};   // pool_tail, overflow: a blocking 8-byte copy into pageable memory
            HIPCHECK(hipMemcpyAsync(two, v.d_counters, sizeof(two), hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            back_tail = two[0]; back_overflow = two[1];
        } else if (overlap) {
            for (size_t a = 0; a < n_aux; ++a) {
                if (!sq_free[step & 1u][a] && (st = next_event(&sq_free[step & 1u][a]))) return st;
                HIPCHECK(hipEventRecord(sq_free[step & 1u][a], w.aux[a]));
            }
            if (poll_step) {
                if ((st = wait_step(step_seq, &back_tail, &back_overflow))) return st;
            } else {
                HIPCHECK(hipEventSynchronize(ev_back));   // the counters are back; the shadow launches go on
                back_tail = w.h_back->pool_tail; back_overflow = w.h_back->overflow;
            }
        } else if (poll_step) {
            if ((st = wait_step(step_seq, &back_tail, &back_overflow))) return st;
        } else {
            HIPCHECK(hipStreamSynchronize(stream));
            back_tail = w.h_back->pool_tail; back_overflow = w.h_back->overflow;
        }
        ++step;
        if (back_overflow) return fail(RTX_ERR_QUEUE_OVERFLOW, "device queue overflow");
        top.end = begin;
        if (top.begin == top.end) --n_segs;
        uint32_t children = back_tail - top_end;
        if (children) seg_push(Seg{top_end, top_end + children, top_gen + 1});
        peak_pool = std::max<uint64_t>(peak_pool, (uint64_t)top_end + children);
        return RTX_OK;
    };
    while (trace_anything && (n_segs > 0 || s_next < n_samples))
        if ((st = one_step())) return bail(st);

    auto finish = [&]() -> rtx_status {
        if (overlap)
            for (int i = 0; i < 2; ++i)
                for (int a = 0; a < 2; ++a)
                    if (sq_free[i][a]) HIPCHECK(hipStreamWaitEvent(stream, sq_free[i][a], 0));
        rtxk_resolve(stream, fr, v.acc, d_rgba);
        HIPCHECK(hipGetLastError());
        if (!stats && !uses_retry) {
            // nothing to read back: rtx_render_device returns with the frame enqueued; the allocation-free variant returns when the
            // frame is complete (rtx.h: the next call on this GPU clears the shared static array, whatever stream it runs on)
            if (noalloc) HIPCHECK(hipStreamSynchronize(stream));
            return RTX_OK;
        }

        unsigned long long slots_stack[RTX_RAY_SLOTS][RTX_RAY_SLOT_WORDS];   // 64 KB
        unsigned long long (*h_slots)[RTX_RAY_SLOT_WORDS] = noalloc ? slots_stack : w.h_ray_slots;
        DCounters back_stack;
        DCounters *h_back = noalloc ? &back_stack : w.h_back;
        if (timing) {
            if ((st = next_event(&ev_last))) return st;
            HIPCHECK(hipEventRecord(ev_last, stream));
        }
        HIPCHECK(hipMemcpyAsync(h_back, v.d_counters, sizeof(DCounters), hipMemcpyDeviceToHost, stream));
        if (stats)
            HIPCHECK(hipMemcpyAsync(h_slots, v.d_ray_slots, RTX_RAY_SLOTS * RTX_RAY_SLOT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        if (coop_defer_ran) {
            if (h_back->coop_deferred <= n_samples / 4096u) s->coop_plain_frames = RTX_COOP_PROBE_EVERY - 1;
        } else if (s->coop_plain_frames > 0) {
            --s->coop_plain_frames;
        }
        if (uses_retry && h_back->retry_overflow) return RTX_INTERNAL_RETRY_OVERFLOW;
        if (!stats) return RTX_OK;
        std::memset(stats, 0, sizeof(*stats));
        for (int sl = 0; sl < RTX_RAY_SLOTS; ++sl) {
            stats->rays_primary += h_slots[sl][RAY_PRIMARY];
            stats->rays_reflect += h_slots[sl][RAY_REFLECT];
            stats->rays_gloss += h_slots[sl][RAY_GLOSS];
            stats->rays_refract += h_slots[sl][RAY_REFRACT];
            for (size_t li = 0; li < s->lights.size(); ++li) {   // shadow groups emitted per light x tries (renderer.rs:194-198)
                uint32_t tries = s->lights[li].kind == RTX_LIGHT_POINT ? 1u : opt->shadow_samples;
                stats->rays_shadow += h_slots[sl][RTX_RAY_SLOT_GROUPS + li] * tries;
            }
        }
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < 3; ++k) stats->work[c][k] = h_back->work[c][k];
        if (timing) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, ev_first, ev_last));
            stats->ms_device = ms;
            for (const Span &sp : spans) {
                HIPCHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
                (sp.kind == 0 ? stats->ms_closest : stats->ms_shadow) += ms;
                if (sp.kind > 0) stats->ms_shadow_light[sp.kind - 1] += ms;
                if (sp.mid) {
                    HIPCHECK(hipEventElapsedTime(&ms, sp.mid, sp.b));
                    stats->ms_shade += ms;
                }
            }
        }
        stats->launches_closest = launches_closest;
        stats->launches_shadow = launches_shadow;
        stats->peak_pool_rays = peak_pool;
        if (!noalloc) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        return RTX_OK;
    };
    if ((st = finish())) return bail(st);
    return RTX_OK;
}

rtx_status render_impl(rtx_scene *s, const rtx_camera *cam, const rtx_options *opt, uint8_t *d_rgba, int32_t *d_ids,
                       hipStream_t stream, rtx_stats *stats, bool noalloc) {
    if (!s || !cam || !opt || !d_rgba) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (cam->image_width == 0 || cam->image_height == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "empty image");
    if (cam->image_height % 8 != 0)
        return fail(RTX_ERR_INVALID_ARGUMENT, "image_height must be a multiple of 8 (renderer.rs:73 underflows otherwise)");
    if (opt->reflect_depth > RTX_MAX_DEPTH || opt->refract_depth > RTX_MAX_DEPTH)
        return fail(RTX_ERR_LIMIT, "reflect_depth / refract_depth above 4095");
    if (opt->rng_mode > RTX_RNG_PHILOX) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown rng_mode");
    if (opt->flags & ~(uint32_t)RTX_FLAG_PUBLIC_MASK) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown bit in rtx_options.flags");
    if (opt->shard_index >= (opt->shard_count ? opt->shard_count : 1)) return fail(RTX_ERR_INVALID_ARGUMENT, "shard_index >= shard_count");
    HIPCHECK(hipSetDevice(s->device));
    rtx_status st = render_once(s, cam, opt, 0u, d_rgba, d_ids, stream, stats, noalloc);
    // A launch met more degenerate shadow feelers than its retry list holds (a light lying in the plane of a floor: every feeler from
    // the floor has a zero direction component).  No feeler is dropped: the frame is rendered again with every shadow launch on the
    // all-f64 restatement of the reference's walk, which needs no retry list.  Slow, and only for such scenes.
    if (st == RTX_INTERNAL_RETRY_OVERFLOW) {
        st = render_once(s, cam, opt, RTX_FLAG_INTERNAL_EXACT_SHADOWS, d_rgba, d_ids, stream, stats, noalloc);
        if (st == RTX_OK && stats) stats->exact_shadow_rerenders = 1;
    }
    return st;
}

} // namespace

// ================================================================================ C ABI
extern "C" {

rtx_status rtx_scene_create(const rtx_scene_desc *d, int32_t device, rtx_scene **out) {
    if (!out) return fail(RTX_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    const bool timing = std::getenv("RTX_BUILD_TIMING") != nullptr;   // stage times of the build on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rtx_scene_create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    Flat fl;
    rtx_status st = flatten(d, fl);
    if (st) return st;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(RTX_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "device index out of range");
    HIPCHECK(hipSetDevice(device));
    Uploader up(device);   // its thread warms the copy path up while the octree is built
    lap("validate + device");
    const unsigned n_threads_env = build_threads();
    RtxPool pool(std::max(1u, n_threads_env));
    // The SAH hierarchy (16-wide box lists + 4-wide nodes) is built ON THE DEVICE (rtx_build.hip), on a thread of its own beside the
    // host's octree build: it needs the finite prims' boxes and the root centre only.  RTX_BUILD_HIERARCHY=host keeps the host
    // builder (rtx_bvh.cpp: the same tree, the A/B and the cross-check of tests/test_gpu_build.py); scenes below
    // RTX_DEVICE_BUILD_MIN prims take it too unless RTX_BUILD_HIERARCHY=device (a device build is ~60 launches whatever the size).
    const int n_fin0 = (int)fl.fin_kind.size();
    std::vector<RtxPrimBox> pboxes((size_t)n_fin0);
    pool.run(((size_t)n_fin0 + 16383) / 16384, [&](size_t chunk) {
        for (int i = (int)(chunk * 16384); i < std::min(n_fin0, (int)((chunk + 1) * 16384)); ++i) {
            const HBox &pb = fl.oct.prim_box[(size_t)i];
            for (int a = 0; a < 3; ++a) { pboxes[(size_t)i].mn[a] = pb.mn[a]; pboxes[(size_t)i].mx[a] = pb.mx[a]; }
            const uint32_t idx = (uint32_t)fl.fin_index[(size_t)i];
            pboxes[(size_t)i].ref = fl.fin_kind[(size_t)i] == RTX_PRIM_TRIANGLE ? (RTX_WREF_TRI | idx) : (RTX_WREF_SPHERE | idx);
        }
    });
    const HBox root_box = fl.oct.root_bounds();   // = the octree root's box (octree.rs:24-25)
    double centre[3], radius = 0.0;
    for (int a = 0; a < 3; ++a) centre[a] = 0.5 * (root_box.mn[a] + root_box.mx[a]);
    for (int a = 0; a < 3; ++a)
        radius = std::max(radius, std::max(std::fabs(root_box.mn[a] - centre[a]), std::fabs(root_box.mx[a] - centre[a])));
    radius *= 1.0 + 1.0e-12;
    const char *hb_env = std::getenv("RTX_BUILD_HIERARCHY");
    const bool device_build = n_fin0 >= 2 && d->n_triangles < (1 << 30) && d->n_spheres < (1 << 30) &&
                              (hb_env ? std::strcmp(hb_env, "device") == 0 : n_fin0 >= RTX_DEVICE_BUILD_MIN);
    if (hb_env && std::strcmp(hb_env, "device") != 0 && std::strcmp(hb_env, "host") != 0)
        return fail(RTX_ERR_INVALID_ARGUMENT, "RTX_BUILD_HIERARCHY must be `host` or `device`");
    RtxDeviceHierarchy dh;
    int dh_rc = 0;
    std::string dh_err;
    std::thread dh_thread;
    if (device_build)
        dh_thread = std::thread([&]() {
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) { dh_rc = (int)e; dh_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return; }
            dh_rc = rtxk_build_hierarchies_device(pboxes.data(), (int64_t)n_fin0, centre, &dh, timing, &dh_err);
        });
    struct Joiner {   // no path out of this function leaves the thread running (it reads pboxes / centre)
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } dh_joiner{dh_thread};
    lap("prim boxes, device build started");
    build_octree(fl, n_threads_env ? &pool : nullptr);
    lap("octree");
    const HostOctree &oct = fl.oct;
    int depth_for_stack = std::max(oct.depth, 0);
    if (depth_for_stack > RTX_MAX_LEVELS) return fail(RTX_ERR_LIMIT, "octree deeper than 12 levels");
    if (oct.nodes.size() >= (1u << 24)) return fail(RTX_ERR_LIMIT, "octree has 2^24 nodes or more");

    std::unique_ptr<rtx_scene> s(new rtx_scene());
    s->device = device;
    auto cleanup = [&](rtx_status e) {
        for (void *p : s->allocs) hipFree(p);
        return e;
    };

    // prim records; prim id = rank among finite prims, planes after (rtx.h)
    int n_fin = (int)fl.fin_kind.size();
    // (new T[n] leaves plain records uninitialised: their pages are first touched by the parallel fill, not by a serial zero-fill)
    const size_t n_tri = (size_t)d->n_triangles, n_sph = (size_t)d->n_spheres;
    std::unique_ptr<DTri[]> tris(new DTri[std::max<size_t>(n_tri, 1)]);
    std::unique_ptr<DTriShade[]> shade(new DTriShade[std::max<size_t>(n_tri, 1)]);
    std::unique_ptr<DSphere[]> spheres(new DSphere[std::max<size_t>(n_sph, 1)]);
    std::vector<DPlane> planes(fl.inf_index.size());
    std::vector<int32_t> ref_of_prim((size_t)n_fin);
    pool.run(((size_t)n_fin + 16383) / 16384, [&](size_t chunk) {
      for (int i = (int)(chunk * 16384); i < std::min(n_fin, (int)((chunk + 1) * 16384)); ++i) {
        int idx = fl.fin_index[(size_t)i];
        if (fl.fin_kind[(size_t)i] == RTX_PRIM_TRIANGLE) {
            DTri &t = tris[(size_t)idx];
            std::memcpy(t.v, d->tri_vertices + (size_t)idx * 9, 9 * sizeof(double));
            t.prim_id = i;
            t.material = d->tri_material[idx];
            DTriShade &sh = shade[(size_t)idx];
            std::memcpy(sh.n, d->tri_normals + (size_t)idx * 9, 9 * sizeof(double));
            std::memcpy(sh.uv, d->tri_uv + (size_t)idx * 6, 6 * sizeof(double));
            sh.pad = 0.0;
            ref_of_prim[(size_t)i] = idx;
        } else {
            DSphere &sp = spheres[(size_t)idx];
            std::memcpy(sp.c, d->sphere_center + (size_t)idx * 3, 3 * sizeof(double));
            sp.r = d->sphere_radius[idx];
            sp.prim_id = i;
            sp.material = d->sphere_material[idx];
            sp.pad = 0.0;
            ref_of_prim[(size_t)i] = ~idx;
        }
      }
    });
    for (size_t j = 0; j < fl.inf_index.size(); ++j) {
        int idx = fl.inf_index[j];
        DPlane &pl = planes[j];
        std::memcpy(pl.abcd, d->plane_abcd + (size_t)idx * 4, 4 * sizeof(double));
        pl.prim_id = n_fin + (int)j;
        pl.material = d->plane_material[idx];
        pl.pad = 0.0;
    }
    lap("prim records");
    // images -> one RGBA8 atlas
    std::vector<DImage> images((size_t)d->n_images);
    std::vector<uint8_t> atlas;
    for (int i = 0; i < d->n_images; ++i) {
        images[(size_t)i].w = d->images[i].width;
        images[(size_t)i].h = d->images[i].height;
        images[(size_t)i].offset = (int64_t)atlas.size();
        size_t nb = (size_t)d->images[i].width * (size_t)d->images[i].height * 4;
        atlas.insert(atlas.end(), d->images[i].rgba, d->images[i].rgba + nb);
        while (atlas.size() % 16) atlas.push_back(0);
    }
    std::vector<rtx_material> materials(d->materials, d->materials + d->n_materials);
    std::vector<rtx_texture> textures(d->textures, d->textures + d->n_textures);
    s->lights.assign(d->lights, d->lights + d->n_lights);
    bool all_opaque = true;
    for (const rtx_material &m : materials) {
        if (m.kind == RTX_MAT_FLAT) continue;
        if (m.transmission[0] != 0.0 || m.transmission[1] != 0.0 || m.transmission[2] != 0.0) all_opaque = false;
        if (m.k_sg > 0.0) s->any_reflective = true;
        if (m.k_tg > 0.0) s->any_refractive = true;
        if (m.glossiness > 2.220446049250313e-16) s->any_glossy = true;
    }

    // what a ray that hits a prim spawns (DScene.ref_class): one byte per prim from its material, the three kinds one after the other
    std::vector<uint8_t> ref_class(n_tri + n_sph + planes.size() + 1, 0);
    {
        std::vector<uint8_t> mat_class(materials.size(), 0);
        for (size_t i = 0; i < materials.size(); ++i) {
            const rtx_material &m = materials[i];
            if (m.kind == RTX_MAT_FLAT) continue;                                        // flatmaterial.rs:15-21
            mat_class[i] = (uint8_t)((m.k_sg > 0.0 ? RTX_CLS_REFLECT : 0u) | (m.k_tg > 0.0 ? RTX_CLS_REFRACT : 0u) |
                                     (m.glossiness > 2.220446049250313e-16 ? RTX_CLS_GLOSSY : 0u));
        }
        for (size_t i = 0; i < n_tri; ++i) ref_class[i] = mat_class[(size_t)d->tri_material[i]];
        for (size_t i = 0; i < n_sph; ++i) ref_class[n_tri + i] = mat_class[(size_t)d->sphere_material[i]];
        for (size_t j = 0; j < planes.size(); ++j) ref_class[n_tri + n_sph + j] = mat_class[(size_t)planes[j].material];
    }

    // everything that is ready goes to the device now, on the uploader's thread, while this thread builds the rest
    DScene &ds = s->d;
    up.submit(ref_class, &ds.ref_class);
    ds.cls_spheres = (int32_t)n_tri;
    ds.cls_planes = (int32_t)(n_tri + n_sph);
    up.submit(tris.get(), n_tri, &ds.tris);
    up.submit(shade.get(), n_tri, &ds.tri_shade);
    up.submit(spheres.get(), n_sph, &ds.spheres);
    up.submit(planes, &ds.planes);
    up.submit(materials, &ds.materials);
    up.submit(textures, &ds.textures);
    up.submit(images, &ds.images);
    up.submit(atlas, &ds.atlas);
    up.submit(s->lights, &ds.lights);
    // nodes + leaf entries (leaf_refs in the order of HostOctree::entries)
    const size_t n_nodes = oct.nodes.size();
    std::vector<DNode> nodes(n_nodes);
    std::vector<int32_t> leaf_refs(oct.entries.size());
    std::vector<uint8_t> nonempty(n_nodes, 0);
    for (size_t i = n_nodes; i-- > 0;) {   // children have larger indices than their parent: one backward pass
        const FNode &hn = oct.nodes[i];
        bool ne = hn.ecount > 0;
        if (hn.first_child >= 0)
            for (int c = 0; c < 8; ++c) ne = ne || nonempty[(size_t)(hn.first_child + c)];
        nonempty[i] = ne ? 1 : 0;
    }
    int n_interior = 0, max_entries = 0;
    for (size_t i = 0; i < n_nodes; ++i) {
        if (oct.nodes[i].first_child >= 0) ++n_interior;
        max_entries = std::max(max_entries, oct.nodes[i].ecount);
    }
    {
        const size_t chunk = 16384, n_chunks = (n_nodes + chunk - 1) / chunk;
        pool.run(n_chunks, [&](size_t t) {
            for (size_t i = t * chunk; i < std::min(n_nodes, (t + 1) * chunk); ++i) {
                const FNode &hn = oct.nodes[i];
                DNode &dn = nodes[i];
                for (int a = 0; a < 3; ++a) { dn.bmin[a] = hn.bbox.mn[a]; dn.bmax[a] = hn.bbox.mx[a]; }
                dn.first_child = hn.first_child;
                dn.entry_begin = hn.ebegin;
                dn.entry_count = hn.ecount;
                dn.child_mask = 0;
                if (hn.first_child >= 0)
                    for (int c = 0; c < 8; ++c)
                        if (nonempty[(size_t)(hn.first_child + c)]) dn.child_mask |= 1u << c;
            }
        });
        const size_t n_e = leaf_refs.size(), e_chunks = (n_e + chunk - 1) / chunk;
        pool.run(e_chunks, [&](size_t t) {
            for (size_t e = t * chunk; e < std::min(n_e, (t + 1) * chunk); ++e) leaf_refs[e] = ref_of_prim[(size_t)oct.entries[e]];
        });
    }
    up.submit(nodes, &ds.nodes);
    up.submit(leaf_refs, &ds.leaf_refs);
    lap("linearise nodes");
    // The conservative f32 twins (DNode32 / DEntry32: boxes relative to the root centre, rounded outward) are derived ON THE DEVICE
    // from the uploaded f64 records (rtxk_build_twins below): nothing of them is built or copied by the host.
    // Every box lies inside the root's (the union of the prim boxes; child boxes can leave it by an ulp of rounding).
    // 16-wide box lists for the cooperative any-hit traversal, 4-wide nodes for the per-lane walks
    std::vector<DWideChild> wide;
    std::vector<DNode4> bvh4;
    RtxWideStats wstats;
    if (device_build) {
        dh_thread.join();
        if (dh_rc != 0) return cleanup(fail(dh_rc == (int)hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP, dh_err));
        s->allocs.push_back(dh.wide);
        s->allocs.push_back(dh.bvh4);
        wstats = dh.stats;
        if (timing)
            std::fprintf(stderr, "    device build: upload %.2f, %d binned levels %.2f, small subtrees %.2f, collapse DP %.2f, emission %.2f ms\n",
                         dh.ms_upload, dh.levels, dh.ms_tree, dh.ms_small, dh.ms_dp, dh.ms_emit);
    } else {
        rtx_build_hierarchies(pboxes, centre, wide, bvh4, &wstats, pool);
    }
    // k_shadow_coop's level stack (24 levels), queue entry bits, 23-bit prim index of a waiting pair
    // (and its 32-bit byte offsets into the box lists, 512 B per node, and into the triangle records, 80 B each)
    const bool wide_ok = 2 * wstats.depth + 2 <= 24 && wstats.nodes < (1 << 23) && d->n_triangles < (1 << 23) && d->n_spheres < (1 << 23);
    // traverse_bvh's stack entries: node index | quantised entry distance (at least 6 bits of it)
    uint32_t bvh4_bits = 1;
    while ((1ll << bvh4_bits) < wstats.nodes4) ++bvh4_bits;
    // (traverse_bvh packs a primitive's index into 24 bits of a pair word)
    const bool bvh4_ok = bvh4_bits <= 25 /* (112-byte nodes at 32-bit byte offsets) */ && d->n_triangles < (1 << 24) && d->n_spheres < (1 << 24) && std::getenv("RTX_NO_BVH4") == nullptr;
    lap(device_build ? "hierarchies (joined device build)" : "hierarchies (SAH, host)");
    if (device_build) {
        ds.wide = dh.wide;
        ds.bvh4 = dh.bvh4;
        s->info.device_bytes += (int64_t)(wstats.nodes * RTX_WIDE * (int64_t)sizeof(DWideChild) + wstats.nodes4 * (int64_t)sizeof(DNode4));
    } else {
        up.submit(wide, &ds.wide);
        up.submit(bvh4, &ds.bvh4);
    }
    // the inverse of the leaf lists (which leaves list a prim), for the two-phase shadow product of scenes with transmissive materials
    std::vector<int32_t> prim_leaf_begin, prim_leaves;
    if (!all_opaque && bvh4_ok) {
        prim_leaf_begin.assign((size_t)n_fin + 1, 0);
        for (int32_t e : oct.entries) ++prim_leaf_begin[(size_t)e + 1];
        for (int i = 0; i < n_fin; ++i) prim_leaf_begin[(size_t)i + 1] += prim_leaf_begin[(size_t)i];
        prim_leaves.resize(oct.entries.size());
        std::vector<int32_t> fill(prim_leaf_begin.begin(), prim_leaf_begin.end() - 1);
        for (size_t i = 0; i < n_nodes; ++i) {
            const FNode &hn = oct.nodes[i];
            if (hn.first_child >= 0) continue;
            for (int e = hn.ebegin; e < hn.ebegin + hn.ecount; ++e) prim_leaves[(size_t)fill[(size_t)oct.entries[(size_t)e]]++] = (int32_t)i;
        }
        up.submit(prim_leaf_begin, &ds.prim_leaf_begin);
        up.submit(prim_leaves, &ds.prim_leaves);
        lap("inverse leaf lists");
    }
    if ((st = up.finish(s.get()))) return cleanup(st);
    DNode32 *d_nodes32 = nullptr;
    DEntry32 *d_entries32 = nullptr;
    if ((st = dev_alloc(&d_nodes32, n_nodes))) return cleanup(st);
    s->allocs.push_back(d_nodes32);
    if ((st = dev_alloc(&d_entries32, leaf_refs.size()))) return cleanup(st);
    s->allocs.push_back(d_entries32);
    s->info.device_bytes += (int64_t)(std::max<size_t>(n_nodes, 1) * sizeof(DNode32) + std::max<size_t>(leaf_refs.size(), 1) * sizeof(DEntry32));
    ds.nodes32 = d_nodes32;
    ds.entries32 = d_entries32;
    lap("rest of the uploads");
    rtxk_build_twins(nullptr, ds.nodes, d_nodes32, (int64_t)n_nodes, ds.leaf_refs, ds.tris, ds.spheres, d_entries32, (int64_t)leaf_refs.size(), centre);
    if (ds.prim_leaves) {
        DNode32 *d_leaf_boxes = nullptr;
        if ((st = dev_alloc(&d_leaf_boxes, prim_leaves.size()))) return cleanup(st);
        s->allocs.push_back(d_leaf_boxes);
        s->info.device_bytes += (int64_t)(std::max<size_t>(prim_leaves.size(), 1) * sizeof(DNode32));
        rtxk_build_leaf_boxes(nullptr, d_nodes32, ds.prim_leaves, d_leaf_boxes, (int64_t)prim_leaves.size());
        ds.prim_leaf_boxes = d_leaf_boxes;
    }
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipGetLastError());
    lap("f32 twins (device)");
    s->wide_built = ds.wide;
    s->bvh4_built = ds.bvh4;
    s->n_wide_nodes = wstats.nodes;
    s->n_bvh4_nodes = wstats.nodes4;
    if (!wide_ok) ds.wide = nullptr;   // too deep for the LDS stack: shadows fall back to the per-lane walk
    if (!bvh4_ok) ds.bvh4 = nullptr;   // the per-lane walks fall back to the octree
    ds.bvh4_node_bits = bvh4_bits;
    ds.shade_features = s->any_glossy ? RTX_SHADE_GLOSSY : 0u;
    for (const rtx_material &m : materials)
        if (m.kind == RTX_MAT_PHONG) ds.shade_features |= RTX_SHADE_PHONG;
    s->n_materials = d->n_materials;
    s->n_textures = d->n_textures;
    ds.n_planes = (int32_t)planes.size();
    ds.n_lights = d->n_lights;
    ds.has_skybox = d->skybox_image[0] >= 0 ? 1 : 0;
    ds.all_opaque = all_opaque ? 1 : 0;
    for (int f = 0; f < 6; ++f) ds.skybox[f] = d->skybox_image[f];
    for (int a = 0; a < 3; ++a) ds.background[a] = d->background[a];
    for (int a = 0; a < 3; ++a) ds.centre[a] = centre[a];
    ds.radius = radius;

    rtx_scene_info &info = s->info;
    info.n_finite = n_fin;
    info.n_infinite = (int32_t)planes.size();
    info.octree_depth = depth_for_stack;
    info.n_nodes = (int32_t)nodes.size();
    info.n_interior = n_interior;
    info.n_leaf_entries = (int64_t)leaf_refs.size();
    info.max_leaf_entries = max_entries;
    info.all_opaque = ds.all_opaque;
    info.wide_nodes = ds.wide ? (int32_t)wstats.nodes : 0;
    info.wide_depth = wstats.depth;
    info.bvh4_nodes = ds.bvh4 ? (int32_t)wstats.nodes4 : 0;
    info.hierarchy_on_device = device_build ? 1 : 0;
    for (int a = 0; a < 3; ++a) { info.root_min[a] = nodes[0].bmin[a]; info.root_max[a] = nodes[0].bmax[a]; }
    *out = s.release();
    return RTX_OK;
}

rtx_status rtx_scene_destroy(rtx_scene *s) {
    if (!s) return RTX_OK;
    hipSetDevice(s->device);
    free_work(s->w);
    for (void *p : s->allocs) hipFree(p);
    delete s;
    return RTX_OK;
}

rtx_status rtx_scene_get_info(const rtx_scene *s, rtx_scene_info *info) {
    if (!s || !info) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
    *info = s->info;
    return RTX_OK;
}

rtx_status rtx_last_kernel_names(const rtx_scene *s, char *out, int64_t cap) {
    if (!s || !out || cap <= 0) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_last_kernel_names: bad argument");
    std::string t = std::string("closest=") + s->last_trace.closest + ";shade=" + s->last_trace.shade;
    for (size_t li = 0; li < s->lights.size(); ++li)
        t += ";shadow" + std::to_string(li) + "=" + (s->last_shadow[li] ? s->last_shadow[li] : "");
    if ((int64_t)t.size() + 1 > cap) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_last_kernel_names: buffer too small");
    std::memcpy(out, t.c_str(), t.size() + 1);
    return RTX_OK;
}

rtx_status rtx_hierarchy_dump(rtx_scene *s, int32_t which, void *out, int64_t cap_bytes, int64_t *n_bytes) {
    if (!s || !n_bytes || which < 0 || which > 1) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_hierarchy_dump: bad argument");
    const void *src = which == 0 ? (const void *)s->wide_built : (const void *)s->bvh4_built;
    const int64_t bytes = which == 0 ? s->n_wide_nodes * RTX_WIDE * (int64_t)sizeof(DWideChild) : s->n_bvh4_nodes * (int64_t)sizeof(DNode4);
    *n_bytes = src ? bytes : 0;
    if (!out || !src) return RTX_OK;
    if (cap_bytes < bytes) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_hierarchy_dump: buffer too small");
    HIPCHECK(hipSetDevice(s->device));
    HIPCHECK(hipMemcpy(out, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return RTX_OK;
}

rtx_status rtx_render_device(rtx_scene *s, const rtx_camera *cam, const rtx_options *opt, void *rgba_out_dev,
                             void *primary_prim_id_out_dev, void *stream, rtx_stats *stats) {
    return render_impl(s, cam, opt, (uint8_t *)rgba_out_dev, (int32_t *)primary_prim_id_out_dev, (hipStream_t)stream, stats, false);
}

rtx_status rtx_render_device_noalloc(rtx_scene *s, const rtx_camera *cam, const rtx_options *opt, void *rgba_out_dev,
                                     void *primary_prim_id_out_dev, void *stream, rtx_stats *stats) {
    return render_impl(s, cam, opt, (uint8_t *)rgba_out_dev, (int32_t *)primary_prim_id_out_dev, (hipStream_t)stream, stats, true);
}

rtx_status rtx_render(rtx_scene *s, const rtx_camera *cam, const rtx_options *opt, uint8_t *rgba_out,
                      int32_t *primary_prim_id_out, rtx_stats *stats) {
    if (!s || !cam || !opt || !rgba_out) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (opt->flags & RTX_FLAG_COMPACT_OUTPUT) return fail(RTX_ERR_INVALID_ARGUMENT, "RTX_FLAG_COMPACT_OUTPUT is for rtx_render_device (device buffers) only");
    HIPCHECK(hipSetDevice(s->device));
    size_t n_pix = (size_t)cam->image_width * cam->image_height;
    Work &w = s->w;
    if (w.out_pixels < n_pix) {
        if (w.out_rgba) hipFree(w.out_rgba);
        if (w.out_ids) hipFree(w.out_ids);
        w.out_rgba = nullptr; w.out_ids = nullptr; w.out_pixels = 0;
        rtx_status st;
        if ((st = dev_alloc(&w.out_rgba, n_pix * 4)) || (st = dev_alloc(&w.out_ids, n_pix))) return st;
        w.out_pixels = n_pix;
    }
    rtx_status st = render_impl(s, cam, opt, w.out_rgba, primary_prim_id_out ? w.out_ids : nullptr, nullptr, stats, false);
    if (st) return st;
    HIPCHECK(hipDeviceSynchronize());
    uint32_t shard_count = opt->shard_count ? opt->shard_count : 1;
    size_t W = cam->image_width;
    for (uint32_t b = opt->shard_index; b < cam->image_height / 8; b += shard_count) {   // only the rows this shard owns
        size_t off = (size_t)b * 8 * W;
        HIPCHECK(hipMemcpy(rgba_out + off * 4, w.out_rgba + off * 4, 8 * W * 4, hipMemcpyDeviceToHost));
        if (primary_prim_id_out)
            HIPCHECK(hipMemcpy(primary_prim_id_out + off, w.out_ids + off, 8 * W * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return RTX_OK;
}

rtx_status rtx_nearest_hits(rtx_scene *s, int64_t n, const double *origin, const double *direction, int32_t *prim_id_out,
                            double *t_out) {
    return rtx_nearest_hits_ex(s, n, origin, direction, prim_id_out, t_out, 0);
}

rtx_status rtx_nearest_hits_ex(rtx_scene *s, int64_t n, const double *origin, const double *direction, int32_t *prim_id_out,
                               double *t_out, uint32_t flags) {
    if (!s || n < 0 || (n && (!origin || !direction || !prim_id_out || !t_out))) return fail(RTX_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(s->device));
    double *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int32_t *d_id = nullptr;
    rtx_status st = RTX_OK;
    auto done = [&](rtx_status e) {
        hipFree(d_o); hipFree(d_d); hipFree(d_t); hipFree(d_id);
        return e;
    };
    if ((st = dev_alloc(&d_o, (size_t)n * 3)) || (st = dev_alloc(&d_d, (size_t)n * 3)) || (st = dev_alloc(&d_t, (size_t)n)) ||
        (st = dev_alloc(&d_id, (size_t)n)))
        return done(st);
    if (hipMemcpy(d_o, origin, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_d, direction, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(RTX_ERR_HIP, "upload failed"));
    rtxk_nearest(nullptr, s->d, n, d_o, d_d, d_id, d_t, flags);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return done(fail(RTX_ERR_HIP, "k_nearest failed"));
    if (hipMemcpy(prim_id_out, d_id, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(t_out, d_t, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(RTX_ERR_HIP, "download failed"));
    return done(RTX_OK);
}

rtx_status rtx_shadow_products(rtx_scene *s, int64_t n, const double *origin, const double *target, double *shadow_out) {
    return rtx_shadow_products_ex(s, n, origin, target, shadow_out, 0);
}

rtx_status rtx_shadow_products_ex(rtx_scene *s, int64_t n, const double *origin, const double *target, double *shadow_out,
                                  uint32_t flags) {
    if (!s || n < 0 || (n && (!origin || !target || !shadow_out))) return fail(RTX_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(s->device));
    double *d_o = nullptr, *d_s = nullptr, *d_out = nullptr;
    rtx_status st = RTX_OK;
    auto done = [&](rtx_status e) {
        hipFree(d_o); hipFree(d_s); hipFree(d_out);
        return e;
    };
    if ((st = dev_alloc(&d_o, (size_t)n * 3)) || (st = dev_alloc(&d_s, (size_t)n * 3)) || (st = dev_alloc(&d_out, (size_t)n * 3)))
        return done(st);
    if (hipMemcpy(d_o, origin, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_s, target, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(RTX_ERR_HIP, "upload failed"));
    rtxk_shadow_products(nullptr, s->d, n, d_o, d_s, d_out, flags);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return done(fail(RTX_ERR_HIP, "k_shadow_products failed"));
    if (hipMemcpy(shadow_out, d_out, (size_t)n * 24, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(RTX_ERR_HIP, "download failed"));
    return done(RTX_OK);
}

rtx_status rtx_shade_probe(rtx_scene *s, int32_t which, int64_t n, const double *in, double *out) {
    if (!s || n < 0 || (n && (!in || !out))) return fail(RTX_ERR_INVALID_ARGUMENT, "bad argument");
    if (which < RTX_PROBE_FRESNEL || which > RTX_PROBE_BBOX) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown probe");
    if (n == 0) return RTX_OK;
    if (which == RTX_PROBE_SKYBOX && !s->d.has_skybox) return fail(RTX_ERR_INVALID_ARGUMENT, "the scene has no skybox");
    for (int64_t i = 0; i < n; ++i) {   // table rows are indices: validate them here, the kernel does not
        double row = in[i * 16];
        if (which == RTX_PROBE_MATERIAL && !(row >= 0 && row < (double)s->n_materials && row == (double)(int)row))
            return fail(RTX_ERR_INVALID_ARGUMENT, "material row out of range");
        if (which == RTX_PROBE_TEXTURE && !(row >= 0 && row < (double)s->n_textures && row == (double)(int)row))
            return fail(RTX_ERR_INVALID_ARGUMENT, "texture row out of range");
    }
    HIPCHECK(hipSetDevice(s->device));
    double *d_in = nullptr, *d_out = nullptr;
    auto done = [&](rtx_status e) {
        hipFree(d_in); hipFree(d_out);
        return e;
    };
    rtx_status st;
    if ((st = dev_alloc(&d_in, (size_t)n * 16)) || (st = dev_alloc(&d_out, (size_t)n * 4))) return done(st);
    if (hipMemcpy(d_in, in, (size_t)n * 16 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return done(fail(RTX_ERR_HIP, "upload failed"));
    rtxk_shade_probe(nullptr, s->d, which, n, d_in, d_out);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return done(fail(RTX_ERR_HIP, "k_shade_probe failed"));
    if (hipMemcpy(out, d_out, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return done(fail(RTX_ERR_HIP, "download failed"));
    return done(RTX_OK);
}

// Diagnostic, no device needed: the octree rtx_scene_create would build, in DFS pre-order
// (children 0..7): per node 6 doubles (bbox min, max), 2 ints (n_children, n_entries); the
// entries are finite-prim ids.  Pass NULL arrays to get the sizes only.
rtx_status rtx_octree_dump(const rtx_scene_desc *d, int32_t *depth, int64_t *n_nodes, int64_t *n_entries, double *bbox_out,
                           int32_t *shape_out, int32_t *entries_out) {
    const unsigned n_threads_env = build_threads();
    RtxPool pool(std::max(1u, n_threads_env));
    Flat fl;
    rtx_status st = flatten(d, fl);
    if (st) return st;
    build_octree(fl, n_threads_env ? &pool : nullptr);
    const HostOctree &oct = fl.oct;
    int64_t ne = (int64_t)oct.entries.size();
    if (depth) *depth = oct.depth;
    if (n_nodes) *n_nodes = (int64_t)oct.nodes.size();
    if (n_entries) *n_entries = ne;
    if (!bbox_out || !shape_out || !entries_out) return RTX_OK;
    std::vector<int> stack{0};
    while (!stack.empty()) {
        int ni = stack.back();
        stack.pop_back();
        const FNode &n = oct.nodes[(size_t)ni];
        for (int a = 0; a < 3; ++a) *bbox_out++ = n.bbox.mn[a];
        for (int a = 0; a < 3; ++a) *bbox_out++ = n.bbox.mx[a];
        *shape_out++ = n.first_child >= 0 ? 8 : 0;
        *shape_out++ = (int32_t)n.ecount;
        for (int e = 0; e < n.ecount; ++e) *entries_out++ = oct.entries[(size_t)(n.ebegin + e)];
        if (n.first_child >= 0)
            for (int c = 7; c >= 0; --c) stack.push_back(n.first_child + c);
    }
    return RTX_OK;
}

const char *rtx_strerror(rtx_status st) {
    switch (st) {
    case RTX_OK: return "ok";
    case RTX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RTX_ERR_NO_DEVICE: return "no HIP device";
    case RTX_ERR_HIP: return "HIP runtime error";
    case RTX_ERR_OUT_OF_MEMORY: return "out of device memory";
    case RTX_ERR_LIMIT: return "scene or options exceed a documented limit";
    case RTX_ERR_QUEUE_OVERFLOW: return "ray pool overflow";
    default: return "unknown status";
    }
}
const char *rtx_last_error(void) { return g_last_error.c_str(); }
const char *rtx_version(void) { return "rtx 0.1 gfx950 f64 wavefront"; }

} // extern "C"
