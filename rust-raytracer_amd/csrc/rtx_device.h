// rtx_device.h — device-side data layout of the flattened scene and the wavefront queues.
// Shared by rtx_api.cpp (host: build, upload, orchestration) and rtx_kernels.hip (gfx950 kernels).
#pragma once
#include <stdint.h>

#include "../../include/rtx.h"

#define RTX_MAX_LEVELS 12     // level stack entries per lane (octree depth <= 12)
#define RTX_MAX_LIGHTS 16
#define RTX_MAX_DEPTH 4095    // reflect/refract depth fits 12 bits of the ray flags
#define RTX_BLOCK 256         // threads per workgroup = 4 wave64
#define RTX_RAY_SLOTS 256
#define RTX_RAY_SLOT_WORDS 32   // u64 per slot: [0..4] rays by type, [8..23] shadow groups emitted per light
#define RTX_RAY_SLOT_GROUPS 8

// One octree node (octree.rs:51-56) in 64 bytes = one half cache line: the node's own bbox
// (children's boxes are re-derived in registers with the reference's expressions,
// octree.rs:81-92), the index of its 8 contiguous children, its leaf entries.
struct alignas(64) DNode {
    double bmin[3];
    double bmax[3];
    int32_t first_child;   // -1: leaf
    int32_t entry_begin;   // into leaf_refs
    int32_t entry_count;
    uint32_t child_mask;   // bit k (k = x*4+y*2+z, octree.rs:76-78): child k's subtree holds >= 1 entry
};

// Conservative single-precision twins used by the fast traversal (closest-hit with near-to-far
// culling and any-hit shadows): boxes are relative to the root centre and rounded OUTWARD to
// f32; the kernel widens them by a per-ray delta that covers every f32 rounding error, so the
// f32 test never rejects a box the reference's f64 test (bbox.rs:99-165) accepts.  Survivors
// still go through the exact f64 entry-bbox and primitive tests, so results do not change.
struct alignas(32) DNode32 {
    float bmin[3];
    float bmax[3];
    uint32_t link;   // interior: (first_child << 8) | child_mask;  leaf: entry_begin
    uint32_t aux;    // interior: 0xFFFFFFFF;                        leaf: entry_count
};
struct alignas(32) DEntry32 {
    float bmin[3];
    float bmax[3];
    int32_t ref;     // >= 0: triangle index, < 0: ~sphere index
    int32_t pad;
};

// A node of the 16-wide box-list hierarchy walked cooperatively by 16 lanes (one child per lane):
// boxes are relative to DScene.centre and rounded outward to f32 like DNode32's.
#define RTX_WIDE 16
#define RTX_WREF_NODE   0x00000000u
#define RTX_WREF_TRI    0x40000000u
#define RTX_WREF_SPHERE 0x80000000u
#define RTX_WREF_EMPTY  0xC0000000u
#define RTX_WREF_KIND   0xC0000000u
#define RTX_WREF_INDEX  0x3FFFFFFFu
// One child of a wide node, 32 B; a node is RTX_WIDE consecutive children (512 B), lane l of a group
// reads child l with two 16-byte loads.  Children that are nodes have consecutive node indices in
// slot order: child s is node  first_node + popcount(node-kind slots below s), so every lane can
// compute the next node from ballots alone, with no cross-lane data movement.
struct alignas(32) DWideChild {
    float lo[3];
    float hi[3];
    uint32_t ref;          // RTX_WREF_* kind | index (index unused for RTX_WREF_NODE children)
    uint32_t first_node;   // same value in all RTX_WIDE children of a node
};

// A node of the 4-wide hierarchy walked per lane (closest hit, any-hit feelers of scenes without transmissive materials): the same
// binary SAH tree as the box lists, cut into 4-wide nodes; SoA so that one 16-byte load fetches a bound of all four children, and
// a lane picks the near / far planes of its ray's direction signs by ADDRESS (lo or hi row), not by per-component selects.
// 112 B.  ref: RTX_WREF_* kind | index (a node's children that are nodes need not be consecutive here).
#define RTX_BVH4 4
struct alignas(16) DNode4 {
    float lo[3][RTX_BVH4];
    float hi[3][RTX_BVH4];
    uint32_t ref[RTX_BVH4];
};

// Intersection record of a triangle: 3 vertices (triangle.rs:90) + ids, 80 B, 16-B aligned.
struct alignas(16) DTri {
    double v[9];
    int32_t prim_id;
    int32_t material;
};
// Shading record (fetched once per closest hit): vertex normals and UVs (triangle.rs:94-97).
struct alignas(16) DTriShade {
    double n[9];
    double uv[6];
    double pad;
};
struct alignas(16) DSphere {
    double c[3];
    double r;
    int32_t prim_id;
    int32_t material;
    double pad;
};
struct alignas(16) DPlane {
    double abcd[4];
    int32_t prim_id;
    int32_t material;
    double pad;
};
struct DImage {
    int32_t w, h;
    int64_t offset; // byte offset of the RGBA8 pixels in the atlas
};

#define RTX_CLS_REFLECT 1u   // is_reflective(): k_sg > 0 (not for FlatMaterial)
#define RTX_CLS_REFRACT 2u   // is_refractive(): k_tg > 0
#define RTX_CLS_GLOSSY 4u    // is_glossy(): glossiness > f64::EPSILON
#define RTX_SHADE_GLOSSY 1u   // some material has glossiness > f64::EPSILON (renderer.rs:155: Ray::perturb)
#define RTX_SHADE_PHONG 2u    // some material is a PhongMaterial (phongmaterial.rs:33: powf(shininess))
struct DScene {
    const DNode *nodes;
    const DNode32 *nodes32;
    const DEntry32 *entries32;  // same order as leaf_refs
    const int32_t *leaf_refs;   // >= 0: triangle index, < 0: ~sphere index
    const DWideChild *wide;     // 16-wide hierarchy over the finite prims: node n = wide[16 n .. 16 n + 15], node 0 = root
    const DNode4 *bvh4;         // 4-wide hierarchy over the same prims, node 0 = root; NULL: the per-lane walks use the octree
    // the inverse of the octree's leaf lists, per finite prim id: the leaves (node indices) that list it = prim_leaves[prim_leaf_begin[id] ..
    // prim_leaf_begin[id + 1]); built for scenes with transmissive materials only (the two-phase shadow product), else NULL
    const int32_t *prim_leaf_begin;
    const int32_t *prim_leaves;
    const DNode32 *prim_leaf_boxes;   // [e] = nodes32[prim_leaves[e]]: a prim's leaf boxes as one run of 32-byte records (built on the device)
    uint32_t bvh4_node_bits;    // bits of a stack entry that hold the node index (the rest: quantised entry distance)
    uint32_t shade_features;    // RTX_SHADE_*: what the scene's materials need of k_shade (0: the lean instance)
    const DTri *tris;
    const DTriShade *tri_shade;
    const DSphere *spheres;
    const DPlane *planes;
    // what a ray that hits the prim spawns, from the prim's material (RTX_CLS_*): triangles [0, n), spheres from cls_spheres, planes from
    // cls_planes; read by the pass that counts a step's children and by k_shade (rtx_kernels.hip: spawn_counts)
    const uint8_t *ref_class;
    int32_t cls_spheres, cls_planes;
    const rtx_material *materials;
    const rtx_texture *textures;
    const DImage *images;
    const uint8_t *atlas;
    const rtx_light *lights;
    int32_t n_planes;
    int32_t n_lights;
    int32_t has_skybox;
    int32_t all_opaque;
    int32_t skybox[6];
    double background[3];
    double centre[3];           // root bbox centre: origin of the f32 coordinates
    double radius;              // largest |coordinate - centre| of any box
};

// Camera (camera.rs:26-50,88-105) + RenderOptions (renderer.rs:14-21) + sharding, by value.
struct DFrame {
    double pos[3], eye[3], right[3], up[3];
    double half_width, half_height, pixel_width, pixel_height;
    double inv_ps;        // 1.0 / pixel_samples          (renderer.rs:78)
    double inv_spp;       // 1.0 / (pixel_samples^2)      (renderer.rs:95)
    uint64_t seed;
    uint32_t W, H, ps, spp;
    uint32_t tiles_x;     // ceil(W / 8)
    uint32_t shard_index, shard_count, n_blocks_local;
    uint32_t reflect_depth, refract_depth, shadow_samples, gloss_samples;
    uint32_t rng_mode, frame, flags, pad;
};

// Ray pool: SoA, `cap` rays. f holds 9 component planes (origin, direction, throughput).
// A ray of the pool: what the closest-hit walk and the shading need of it in ONE 64-byte line (four 16-byte loads of a lane; a launch
// that takes its rays through an index touches one line per ray instead of one per plane), the throughput apart (k_shade alone reads it),
// and the flags once more as a plane of their own for the passes that read nothing else of a ray (block_counts).
struct DRayRec {
    double o[3], d[3];
    uint64_t key;       // RNG path key of the trace() call this ray stands for
    uint32_t sample;    // primary sample the ray contributes to
    uint32_t flags;
};
static_assert(sizeof(DRayRec) == 64, "one ray record per 64-byte line");
struct DPool {
    DRayRec *rec;       // [cap]
    double *thr;        // [3][cap]
    uint32_t *flags;    // [cap]
    uint32_t *cell;     // [cap] sort key of a child ray: the octant of its direction (3 bits) over the cell of its origin (21-bit Morton code in
                        // the scene's cube); written by k_shade, read by the sort that orders a later generation's closest-hit launch. NULL: unused
    uint32_t cap;
};
#define RAYF_REFLECT_SHIFT 0
#define RAYF_REFRACT_SHIFT 12
#define RAYF_DEPTH_MASK 0xFFFu
#define RAYF_INSIDE (1u << 24)
#define RAYF_TYPE_SHIFT 25
#define RAYF_DEAD (1u << 31)
enum { RAY_PRIMARY = 0, RAY_REFLECT = 1, RAY_GLOSS = 2, RAY_REFRACT = 3, RAY_SHADOW = 4 };

// Shadow groups: one entry per (closest hit, light); the shadow kernel gives each of the
// light's `tries` feelers its own lane.  f holds 6 planes per light: hit position, weight.
struct DShadowQ {
    // COMPACT: k_shade writes a group per ray that hit something, in ray order without gaps (k_closest_retry / k_group_base), so the
    // shadow launches run full waves whatever share of the rays left the scene (bunny: half of them); the grid is sized for every
    // ray of the step and blocks past `*count` leave at once
    double *f;          // [n_lights][6][cap]
    uint64_t *key;      // [n_lights][cap]
    uint32_t *sample;   // [n_lights][cap]
    uint32_t *retry;    // [n_lights][2 + 2 * retry_cap]: count, pad, then (feeler index, primary sample) pairs: the degenerate feelers
                        // of the light's launch, answered by k_shadow_retry with the all-f64 walk
    double *res;        // [n_lights][3][cap]: what the light's shadow launch found for the group (shadow_intensity's return value), stored for
                        // k_accumulate; NULL in the copy a launch gets: that launch adds weight * shadow to the sample itself (atomics)
    uint32_t *sample0;  // [cap]: the group's primary sample once more (k_shadow_coop<MULTI> marks the per-light copies it is done with)
    uint32_t *count;    // one word: the groups of the step (the same for every light: a ray that hits emits one per light)
    uint32_t *wide;     // [n_lights][1 + cap]: count, then the queue groups k_shadow_coop<MULTI> leaves to the per-lane kernel that follows it
    // Sorted launches (per-lane shadow kernels on rays of the second generation and later, whose hit points lie all over the scene):
    // k_shade writes the cell of the group's hit point (24-bit Morton code in the scene's cube), rtxk_sort_groups orders the groups by
    // it, and the light's launch takes its groups through `order`: neighbouring lanes walk neighbouring feelers.  NULL: queue order.
    uint32_t *cell;     // [cap]
    uint32_t *order;    // [1 + cap]: count, then the queue's groups by cell
    uint32_t cap;
    uint32_t retry_cap; // pairs per light (RTX_RETRY_CAP in the library's own workspace, smaller in the static one)
};
#ifndef RTX_RETRY_CAP
#define RTX_RETRY_CAP (1u << 20)   // (2^14 in the forced-failure test build librtx_smallstack.so: a 1.5 M-sample frame overflows it)
#endif
// internal bit of DFrame.flags (never accepted from rtx_options.flags): every shadow launch runs the all-f64 restatement of the
// reference's walk (k_shadow<MODE_PRODUCT>), which is valid for every feeler of every scene.  Set for the re-render of a frame
// whose retry list overflowed (more degenerate feelers in one launch than the list holds).
#define RTX_FLAG_INTERNAL_EXACT_SHADOWS 0x80000000u
// internal bit: k_shade's small instance (see the kernel): set by the host for frames whose shadow feelers run in the cooperative kernel
// beside the closest-hit / shade chain
#define RTX_FLAG_INTERNAL_SMALL_SHADE 0x40000000u
#define RTX_FLAG_PUBLIC_MASK 0x1Fu

struct DCounters {
    uint32_t pool_tail;
    uint32_t overflow;
    uint32_t reserved_[RTX_MAX_LIGHTS];
    // closest-hit rays by type, counted one atomic per wave into RTX_RAY_SLOTS slots (one 128-B line
    // each, slot = block index % RTX_RAY_SLOTS): tens of millions of atomics on ONE address would
    // serialise at ~12 ns each and dominate the frame.  Shadow feelers are counted on the host
    // (groups x tries per launch).
    unsigned long long (*ray_slots)[RTX_RAY_SLOT_WORDS];
    // builds with -DRTX_COUNT_WORK only (librtx_count.so): [0] closest-hit rays, [1] shadow feelers;
    // nodes visited, leaf entries fetched, prim tests run
    unsigned long long work[2][3];
    // set by the any-hit shadow kernels when a launch's retry list is full.  It lies behind `ray_slots`, so the per-step upload
    // of (pool_tail, overflow) never clears it while shadow launches of an earlier step are still running; the host reads it
    // once, after the frame's last shadow launch.
    uint32_t retry_overflow;
    // queue groups k_shadow_coop<false, true> has handed to the per-lane kernel in this frame (wide beams: hit points outside the scene's
    // cube); behind `ray_slots` like retry_overflow, cleared with it at the start of a frame, read once after it
    uint32_t coop_deferred;
};

// ---- launch wrappers implemented in rtx_kernels.hip (all enqueue on `stream`) -----------
struct hipStreamOpaque;
typedef void *rtx_stream_t;

void rtxk_generate(rtx_stream_t stream, const DFrame &fr, DPool pool, uint32_t pool_base, uint32_t s0, uint32_t count);
// (names: what was launched, as rocprofv3 prints it, e.g. "k_closest<5>": rtx_last_kernel_names, for the measurement harness)
struct RtxTraceNames { const char *closest = ""; const char *shade = ""; };
// what the sort of a step's shadow groups needs beside the queue's own arrays (shared by the two queues: the sorts run one after the
// other on the main stream)
struct RtxSortBuf { uint32_t *cell_sorted = nullptr; void *tmp = nullptr; size_t tmp_bytes = 0; uint32_t *ray_order = nullptr; };
size_t rtxk_sort_temp_bytes(uint32_t n);
// orders the groups [0, n) of the queue by sq.cell (slots past *sq.count hold ~0 and come last) into sq.order; rtx_sort.hip
void rtxk_sort_groups(rtx_stream_t stream, const DShadowQ &sq, const RtxSortBuf &sb, uint32_t n);
// orders the rays [begin, begin + n) of the pool by pool.cell into sb.ray_order (indices relative to begin)
void rtxk_sort_rays(rtx_stream_t stream, const DPool &pool, const RtxSortBuf &sb, uint32_t begin, uint32_t n);
void rtxk_trace(rtx_stream_t stream, const DScene &sc, const DFrame &fr, DPool pool, uint32_t begin, uint32_t count,
                DShadowQ sq, DCounters *counters, double *acc, int32_t *prim_id_out, double *hit_t, int32_t *hit_ref,
                void *wait_before_shade, void *wait_before_shade2 /* hipEvent_t or NULL: k_shade (it writes `sq`) waits for them, k_closest does not */,
                void *record_after_closest /* hipEvent_t or NULL: recorded between the closest-hit launches and the shading launches (timing) */,
                uint32_t *group_base /* [4 * ceil(count / RTX_BLOCK)]: rays that hit something per workgroup, their exclusive scan; children per workgroup, their scan */,
                uint32_t child_begin /* the pool's tail before the step: where its children go */,
                RtxTraceNames *names = nullptr, const RtxSortBuf *sortbuf = nullptr /* with sq.cell and sq.order: the step's groups are sorted */,
                bool sort_rays = false /* with sortbuf and pool.cell: the closest-hit launch takes the step's rays by cell */,
                uint32_t *host_step = nullptr /* pinned host memory, 3 words: k_group_base stores the pool's tail after the step, the overflow flag and */,
                uint32_t host_seq = 0 /* ... this number (last, with release semantics): the host polls for it */,
                uint32_t prim_s0 = 0xFFFFFFFFu /* a step of primaries whose records nobody stored: pool slot i stands for sample prim_s0 + i (the kernels compute the ray) */);
// returns the name of the kernel that does the launch's work ("" when nothing was launched)
const char *rtxk_shadow(rtx_stream_t stream, const DScene &sc, const DFrame &fr, DShadowQ sq, uint32_t light, uint32_t n_groups,
                 uint32_t tries, DCounters *counters, double *acc,
                        bool coop_defer /* k_shadow_coop<false, true> where 16 tries or a multiple share a hit point: see the kernel */);
// a launch with this many tries per hit point can store its groups' results for k_accumulate (a group then lies within one wave)
inline bool rtx_shadow_defers(uint32_t tries) { return tries <= 64u && (tries & (tries - 1u)) == 0u; }
// after the last light's shadow launch of a step: adds the deferred results of the lights in the mask to the samples (k_accumulate)
void rtxk_accumulate(rtx_stream_t stream, DShadowQ sq, uint32_t n_lights, uint32_t lights, uint32_t n_groups, double *acc);
void rtxk_resolve(rtx_stream_t stream, const DFrame &fr, const double *acc, uint8_t *rgba_out);
void rtxk_nearest(rtx_stream_t stream, const DScene &sc, int64_t n, const double *origin, const double *direction,
                  int32_t *prim_id, double *t_out, uint32_t flags);
void rtxk_shadow_products(rtx_stream_t stream, const DScene &sc, int64_t n, const double *origin, const double *target,
                          double *out, uint32_t flags);
// scene build, on the device: the conservative f32 twins of the octree (DNode32 per node, DEntry32 per leaf entry: the entry's prim box
// recomputed from the prim record) from the uploaded f64 records; boxes relative to `centre`, rounded outward
void rtxk_build_leaf_boxes(rtx_stream_t stream, const DNode32 *nodes32, const int32_t *prim_leaves, DNode32 *out, int64_t n);
void rtxk_build_twins(rtx_stream_t stream, const DNode *nodes, DNode32 *nodes32, int64_t n_nodes, const int32_t *leaf_refs,
                      const DTri *tris, const DSphere *spheres, DEntry32 *entries32, int64_t n_entries, const double centre[3]);
void rtxk_shade_probe(rtx_stream_t stream, const DScene &sc, int which, int64_t n, const double *in, double *out);
// sets counters->pool_tail, clears counters->overflow (a one-thread kernel: the allocation-free entry point has no pinned
// host buffer to copy them from)
// zeroes the whole record and sets its ray_slots pointer (same reason)
void rtxk_init_counters(rtx_stream_t stream, DCounters *counters, unsigned long long (*ray_slots)[RTX_RAY_SLOT_WORDS]);
// The static device scratch of rtx_render_device_noalloc: a file-scope __device__ array in rtx_kernels.hip.  Resolves (once per
// device, with that device current) and returns the array's address ON `device`; NULL + a hipError_t code on failure.
#define RTX_STATIC_SCRATCH_BYTES (32u << 20)
#define RTX_MAX_DEVICES 64
void *rtxk_static_scratch(int device, int *hip_error);
