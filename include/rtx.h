/*
 * rtx.h — C ABI of the MI355X-native per-pixel trace loop.
 *
 * This is the drop-in boundary for ONE path of gyng/rust-raytracer: the body of
 *     Renderer::render(&self, camera: Camera, shared_scene: Arc<Scene>) -> Surface
 *                                                   (src/raytracer/renderer.rs:30-65)
 * i.e. render_tile (renderer.rs:67-103) + the recursive Renderer::trace
 * (renderer.rs:105-247) with everything below it (ray.rs, octree.rs, geometry/,
 * material/, light/).  The reference has no FFI of its own; its
 * Scene is a graph of Box<dyn Trait> objects that cannot cross a language
 * boundary, so the Rust side flattens its concrete types into the plain arrays
 * below (INTEGRATION.md shows the `extern "C"` stub) and this library does the
 * rest on the GPU.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer it passes in,
 *     the library copies what it needs during rtx_scene_create;
 *   - every entry point returns an rtx_status, nothing aborts or throws across
 *     the ABI (the reference panics / process::exit(1): main.rs:66-111);
 *   - there is NO CPU fallback: without a HIP device every call that needs one
 *     fails with RTX_ERR_NO_DEVICE / RTX_ERR_HIP;
 *   - a scene handle may be rendered from one host thread at a time;
 *     independent handles are independent;
 *   - all arithmetic is IEEE f64 with the reference's operation order.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rtx_status;
enum {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARGUMENT = 1,
    RTX_ERR_NO_DEVICE = 2,
    RTX_ERR_HIP = 3,
    RTX_ERR_OUT_OF_MEMORY = 4,
    RTX_ERR_LIMIT = 5,          /* scene exceeds a documented limit (octree levels, node count) */
    RTX_ERR_QUEUE_OVERFLOW = 6  /* ray pool exhausted (raise rtx_options.pool_rays) */
};

/* ---- materials: src/material/materials/ (all files) --------------------------------- */
enum { RTX_MAT_COOK_TORRANCE = 0, RTX_MAT_PHONG = 1, RTX_MAT_FLAT = 2 };

/* One row per DISTINCT material (the reference clones one Box<Material> per
 * triangle, util/import.rs:104; the flattening dedupes them).
 * CookTorranceMaterial (cooktorrancematerial.rs:8-23): all fields; shine = roughness.
 * PhongMaterial        (phongmaterial.rs:7-21): shine = shininess; gauss_constant unused.
 * FlatMaterial         (flatmaterial.rs:6-8): diffuse = color; everything else ignored. */
typedef struct rtx_material {
    int32_t kind;
    int32_t texture;         /* diffuse_texture: index into rtx_scene_desc.textures, -1 = None */
    double k_a, k_d, k_s, k_sg, k_tg;
    double ambient[3];
    double diffuse[3];
    double specular[3];
    double transmission[3];
    double shine;
    double gauss_constant;
    double glossiness;
    double ior;
} rtx_material;

/* ---- textures: src/material/textures/ (all files) ------------------------------------ */
enum { RTX_TEX_CHECKER = 0, RTX_TEX_UV = 1, RTX_TEX_IMAGE = 2 };

typedef struct rtx_texture {
    int32_t kind;
    int32_t image;           /* RTX_TEX_IMAGE: index into rtx_scene_desc.images */
    double color1[3];        /* CheckerTexture.color1 (checkertexture.rs:7) r,g,b in [0,1] */
    double color2[3];
    double scale;
} rtx_texture;

/* Surface as ImageTexture holds it (imagetexture.rs:7-9): RGBA8, row-major, row 0 first. */
typedef struct rtx_image {
    int32_t width, height;
    const uint8_t *rgba;
} rtx_image;

/* ---- lights: src/light/lights/ (all files) ------------------------------------------- */
enum { RTX_LIGHT_POINT = 0, RTX_LIGHT_SPHERE = 1 };

typedef struct rtx_light {
    int32_t kind;
    int32_t reserved;
    double position[3];
    double color[3];
    double radius;           /* SphereLight only (spherelight.rs:6-10) */
} rtx_light;

/* ---- primitives: src/geometry/prims/ (all files) -------------------------------------- */
enum { RTX_PRIM_SPHERE = 0, RTX_PRIM_PLANE = 1, RTX_PRIM_TRIANGLE = 2 };

/*
 * The flattened Scene (src/scene/scene.rs:7-12).
 *
 * Primitive order matters: the reference's octree keeps finite prims in
 * insertion order (octree.rs:19-20,38) and that order defines the prim id this
 * library reports: prim_id = rank among finite prims in insertion order; planes
 * (infinite, plane.rs:19-23) follow as n_finite + j; a miss is -1.
 * order_kind/order_index give the insertion order over all prims (n_prims =
 * n_spheres + n_planes + n_triangles entries).  If both are NULL the order is:
 * all spheres, all triangles, all planes.
 */
typedef struct rtx_scene_desc {
    int32_t n_spheres;
    const double *sphere_center;    /* [n_spheres][3]  Sphere.center  (sphere.rs:13) */
    const double *sphere_radius;    /* [n_spheres]     Sphere.radius  (sphere.rs:14) */
    const int32_t *sphere_material; /* [n_spheres]     row of materials */

    int32_t n_planes;
    const double *plane_abcd;       /* [n_planes][4]   Plane.a,b,c,d  (plane.rs:12-15) */
    const int32_t *plane_material;

    int32_t n_triangles;
    const double *tri_vertices;     /* [n_triangles][3][3]  Triangle.vertices (triangle.rs:90) */
    const double *tri_normals;      /* [n_triangles][3][3]  Triangle.normals  (triangle.rs:94)  */
    const double *tri_uv;           /* [n_triangles][3][2]  Triangle.texinfo  (triangle.rs:97)  */
    const int32_t *tri_material;

    int32_t n_prims;
    const int32_t *order_kind;      /* [n_prims] RTX_PRIM_*  (or NULL) */
    const int32_t *order_index;     /* [n_prims] index into the per-kind arrays (or NULL) */

    int32_t n_materials;
    const rtx_material *materials;
    int32_t n_textures;
    const rtx_texture *textures;
    int32_t n_images;
    const rtx_image *images;
    int32_t n_lights;
    const rtx_light *lights;        /* Scene.lights order (renderer.rs:114 folds in this order) */

    double background[3];           /* Scene.background */
    int32_t skybox_image[6];        /* CubeMap.faces[0..5] = left,right,down,up,front,back
                                       (cubemap.rs:12-19); skybox_image[0] < 0 = no skybox */
} rtx_scene_desc;

/* Camera::new arguments (src/scene/camera.rs:26-28). */
typedef struct rtx_camera {
    double position[3];
    double look_at[3];
    double up[3];
    double fov_deg;
    uint32_t image_width;
    uint32_t image_height;
} rtx_camera;

enum { RTX_RNG_CENTER = 0, RTX_RNG_PHILOX = 1 };

/* RenderOptions (renderer.rs:14-21) + what the reference leaves to thread_rng()
 * and num_cpus (main.rs:138). Zero-initialise, then set the five knobs. */
typedef struct rtx_options {
    uint32_t reflect_depth;
    uint32_t refract_depth;
    uint32_t shadow_samples;
    uint32_t gloss_samples;
    uint32_t pixel_samples;   /* SIDE of the sample grid: samples per pixel = pixel_samples^2 */
    uint32_t rng_mode;        /* RTX_RNG_CENTER: every uniform draw is 0.5 (deterministic parity
                                 mode); RTX_RNG_PHILOX: counter-based Philox4x32-10 keyed on
                                 (seed, frame, pixel, subsample, path node, light, try) */
    uint64_t seed;
    uint32_t frame;           /* animation frame number, part of the RNG key */
    /* Tile-row sharding for multi-GPU stills: image rows are cut into blocks
     * of 8 (the reference's tile height, renderer.rs:42); this call renders the blocks b with
     * b % shard_count == shard_index and leaves all other rows of the output untouched.
     * shard_count == 0 means 1. */
    uint32_t shard_index;
    uint32_t shard_count;
    uint32_t batch_rays;      /* rays traced per wavefront step (0 = default: 8 Mi, fewer when the pool of a deep recursion would exceed a quarter of the free device memory) */
    uint32_t pool_rays;       /* capacity of the ray pool in rays (0 = default: what the frame and its recursion depth can need at this batch size) */
    uint32_t flags;           /* RTX_FLAG_* */
} rtx_options;

enum {
    RTX_FLAG_COMPACT_OUTPUT = 16, /* rtx_render_device* only: rgba_out_dev / primary_prim_id_out_dev hold ONLY the call's own 8-row blocks
                                    (shard_index, shard_index + shard_count, ...), packed in that order: ceil((H/8 - shard_index) /
                                    shard_count) * 8 rows of image_width pixels, instead of a full frame with the other ranks' rows
                                    untouched.  The multi-GPU still gather (`Surface::merge`, surface.rs:83-94) sends this
                                    buffer as it is. */
    RTX_FLAG_OCTREE_WALK = 8,    /* the per-lane closest-hit and any-hit walks run over the reference's octree (conservative f32
                                    twins of its nodes and leaf entries) instead of the 4-wide SAH hierarchy built beside it (A/B
                                    measurements and tests: both reach a superset of the prims a ray truly hits and decide each
                                    with the same exact f64 tests, so the results are the same but for exact-t ties) */
    RTX_FLAG_SERIAL = 4,         /* everything on the caller's stream, one launch after the other: by default the
                                    shadow launches of a wavefront step run on a stream of the library's own and
                                    overlap the next step's closest-hit launch (A/B measurements, clean per-kernel
                                    timings; the image does not depend on it beyond the order of f64 atomic adds) */
    RTX_FLAG_NO_COOP = 2,        /* shadow feelers walk one ray per lane instead of the
                                    cooperative 16-lane box-list traversal (A/B measurements; the
                                    cooperative kernel is used for 4, 8 or a multiple of 16 tries per
                                    hit point in scenes without transmissive materials) */
    RTX_FLAG_REFERENCE_ORDER = 1 /* the all-f64 restatement of the reference's walks: closest-hit ties
                                    resolve exactly as the reference's unordered DFS would (first
                                    visited wins, ray.rs:52-60; no near-to-far culling), and in scenes
                                    with transmissive materials the shadow product walks the octree
                                    in f64 instead of classifying boxes in f32 first. Slow; for
                                    parity experiments. */
};

typedef struct rtx_stats {
    uint64_t rays_primary;    /* get_nearest_hit calls from render_tile's trace */
    uint64_t rays_reflect;    /* mirror reflection rays (renderer.rs:164) */
    uint64_t rays_gloss;      /* glossy reflection rays (renderer.rs:158-161) */
    uint64_t rays_refract;    /* transmission rays incl. total internal reflection (renderer.rs:183) */
    uint64_t rays_shadow;     /* shadow tries (renderer.rs:198) */
    double ms_total;          /* wall time of the call, host clock */
    double ms_device;         /* first kernel start -> last kernel end, HIP events */
    double ms_closest;        /* sum of the durations of the closest-hit + shading chains (one per wavefront step), HIP events */
    double ms_shadow;         /* sum of trace_shadow kernel durations, HIP events */
    uint32_t launches_closest;
    uint32_t launches_shadow;
    uint64_t peak_pool_rays;
    /* traversal work actually done by the kernels, all 0 unless the library was built with
     * -DRTX_COUNT_WORK (librtx_count.so, profiling only): [0] closest-hit rays, [1] shadow
     * feelers; nodes visited, leaf entries fetched, primitive tests run */
    uint64_t work[2][3];
    uint64_t exact_shadow_rerenders;   /* 1 if a shadow launch met more degenerate feelers than its retry list holds and the frame
                                          was therefore rendered again with all-f64 shadow walks ("Degenerate rays") */
    double ms_shade;          /* the part of ms_closest spent in the shading launches (queue-slot scan + k_shade): ms_closest - ms_shade
                                 = the closest-hit kernels (HIP events on the launches' own stream) */
    double ms_shadow_light[16];   /* ms_shadow split by light: lights of different kinds / try counts run different kernels */
} rtx_stats;

typedef struct rtx_scene_info {
    int32_t n_finite;         /* prims in the octree */
    int32_t n_infinite;       /* planes */
    int32_t octree_depth;     /* round(1.2 ln N / ln 8), octree.rs:29 (clamped to >= 0 for reporting) */
    int32_t n_nodes;
    int32_t n_interior;
    int64_t n_leaf_entries;   /* sum of leaf_data lengths = N x duplication factor */
    int32_t max_leaf_entries;
    int32_t all_opaque;       /* 1 if every material's transmission() is exactly (0,0,0) */
    int64_t device_bytes;     /* bytes uploaded for the scene */
    double root_min[3], root_max[3];
    int32_t wide_nodes;       /* 16-wide box lists of the cooperative shadow traversal (0: not used) */
    int32_t wide_depth;
    int32_t bvh4_nodes;       /* 4-wide nodes of the per-lane walks (0: not used) */
    int32_t hierarchy_on_device;   /* 1: both cuts of the SAH tree were built by the device-side builder, 0: by the host's */
} rtx_scene_info;

typedef struct rtx_scene rtx_scene;

/* Replaces the scene half of the seam: what `Arc<Scene>` is to Renderer::render
 * (renderer.rs:30, main.rs:117).  Builds the octree exactly as Octree::from_iter does
 * (octree.rs:15-43, 64-133), linearises it and uploads everything to HIP device `device`. */
rtx_status rtx_scene_create(const rtx_scene_desc *desc, int32_t device, rtx_scene **out);

/* Drop for Arc<Scene>. NULL is a no-op. */
rtx_status rtx_scene_destroy(rtx_scene *scene);

rtx_status rtx_scene_get_info(const rtx_scene *scene, rtx_scene_info *info);

/* Replaces Renderer::render (renderer.rs:30-65): renders the frame seen by `camera` and
 * writes image_width*image_height RGBA8 pixels, row-major, row 0 = top of the image
 * (Surface.buffer, surface.rs:18-25; abs_y = H-1-row, renderer.rs:73), alpha = 255
 * (ColorRGBA::new_rgb, colorrgba.rs:88-90).  rgba_out / primary_prim_id_out are HOST
 * buffers; primary_prim_id_out (int32 per pixel: prim id hit by sub-sample (0,0), -1 = miss)
 * and stats may be NULL.  image_height must be a multiple of 8 (the reference underflows
 * otherwise, renderer.rs:73). */
rtx_status rtx_render(rtx_scene *scene, const rtx_camera *camera, const rtx_options *options,
                      uint8_t *rgba_out, int32_t *primary_prim_id_out, rtx_stats *stats);

/* Same, but the outputs are DEVICE pointers on the scene's device and the work is enqueued on
 * `stream` (a hipStream_t passed as void*, NULL = default stream).  The call returns after the
 * last kernel has been enqueued and the pool bookkeeping it needs has completed; the caller
 * synchronises the stream before reading the outputs.  This is the entry point bench.py and
 * the multi-GPU gather use: nothing crosses PCIe inside it.  Unless RTX_FLAG_SERIAL is set the
 * shadow launches run on two non-blocking streams that belong to the scene handle; they are ordered
 * against `stream` with events in both directions (they start after the step's shading on `stream`,
 * and `stream` waits for them before the resolve), so the caller sees ordinary stream semantics. */
rtx_status rtx_render_device(rtx_scene *scene, const rtx_camera *camera, const rtx_options *options,
                             void *rgba_out_dev, void *primary_prim_id_out_dev, void *stream,
                             rtx_stats *stats);

/* NO COUNTERPART IN THE REFERENCE.  rtx_render_device for callers that cannot allocate: tight serving loops, harnesses that
 * forbid hipMalloc, code that must not touch the allocator between two of its own captures.  (NOT capturable itself: the call
 * synchronises `stream` once per wavefront step, because the host sizes each launch from the step's counters.)  Same arguments, same frame (to the order of the f64 atomic adds), but the call
 * allocates nothing: no hipMalloc / hipFree / hipHostMalloc, no stream or event creation, no host heap.  Its workspace (ray pool,
 * shadow queue, hit records, counters, sample accumulators) is carved out of a static 32 MiB device array that every GPU the
 * library runs on holds until the process ends; the batch shrinks until the call fits, and a call that cannot fit (the
 * accumulators alone take 24 bytes per primary sample: about 1.3 M samples at most; reflect_depth + refract_depth above 62) fails
 * with RTX_ERR_LIMIT.  Larger frames go through in pieces with shard_index / shard_count (8-row blocks).  Every launch runs on
 * `stream` (as with RTX_FLAG_SERIAL), the call returns when the frame is complete (it synchronises `stream` after the resolve, so
 * the next call's clearing of the shared array cannot overtake it), rtx_options.pool_rays is ignored, and the timing fields of
 * `stats` (ms_*) stay 0.  All calls on one GPU share that GPU's array: calls that overlap on one GPU race (one call at
 * a time per GPU, whatever the scene handle); the array keeps the last call's contents and each call clears what it needs. */
rtx_status rtx_render_device_noalloc(rtx_scene *scene, const rtx_camera *camera, const rtx_options *options,
                                     void *rgba_out_dev, void *primary_prim_id_out_dev, void *stream, rtx_stats *stats);

/* Replaces Ray::get_nearest_hit (src/raytracer/ray.rs:44-66) for a batch of n rays given as
 * host arrays origin[n][3], direction[n][3] (directions as the caller has them; the reference
 * passes unit vectors).  Writes prim_id_out[n] (-1 = None) and t_out[n] (+inf for a miss).
 * Used by the parity tests to compare hit ids / t bit for bit with the oracle. */
rtx_status rtx_nearest_hits(rtx_scene *scene, int64_t n, const double *origin, const double *direction,
                            int32_t *prim_id_out, double *t_out);
/* Same query (ray.rs:44-66).  flags = 0: what the renderer does per ray (what rtx_nearest_hits calls): the conservative
 * f32 walk with exact f64 tests, and for rays with a direction component below 1e-9 the all-f64 walk (they can lie in a face
 * plane of an octree cell, where the reference's own walk loses subtrees: DESIGN.md, "Degenerate rays");
 * RTX_FLAG_REFERENCE_ORDER: the all-f64 restatement in the reference's visitation order for every ray. */
rtx_status rtx_nearest_hits_ex(rtx_scene *scene, int64_t n, const double *origin, const double *direction,
                               int32_t *prim_id_out, double *t_out, uint32_t flags);

/* Replaces the candidate fold in Renderer::shadow_intensity (renderer.rs:203-217) for a batch
 * of n shadow feelers from origin[i] towards target[i] (host arrays [n][3]): shadow_out[n][3]
 * is the product of transmission() over all occluders, duplicates included. */
rtx_status rtx_shadow_products(rtx_scene *scene, int64_t n, const double *origin, const double *target,
                               double *shadow_out);
/* Same fold (renderer.rs:203-217 over the candidates of octree.rs:145-189).  flags = 0: the walk the renderer runs per
 * feeler in scenes with transmissive materials (f32-classified boxes, exact decision per leaf, mailbox of known
 * misses; what rtx_shadow_products calls); RTX_FLAG_REFERENCE_ORDER: the all-f64 restatement of the reference's walk.
 * The parity tests compare the two with each other and with the oracle, ray by ray. */
rtx_status rtx_shadow_products_ex(rtx_scene *scene, int64_t n, const double *origin, const double *target,
                                  double *shadow_out, uint32_t flags);

/* Parity probe for the shading functions the reference holds no test for: evaluates, on the device and with the very
 * functions the shading kernel calls, one of
 *   RTX_PROBE_FRESNEL   Renderer::fresnel_reflect (renderer.rs:226-247)     in: ior, i[3], n[3], inside          out: F
 *   RTX_PROBE_REFRACT   Vec3::refract (vec3.rs:77-92)                       in: v[3], n[3], ior, inside          out: t[3], ok
 *   RTX_PROBE_REFLECT   Vec3::reflect (vec3.rs:70-72)                       in: v[3], n[3]                       out: r[3]
 *   RTX_PROBE_MATERIAL  Material::sample (cooktorrancematerial.rs:26-58, phongmaterial.rs:24-36, flatmaterial.rs:11-13)
 *                                                                           in: material row, n[3], i[3], l[3], u, v   out: rgb
 *   RTX_PROBE_TEXTURE   Texture::color (checkertexture.rs:13-23, uvtexture.rs:10-14, imagetexture.rs:26-47)
 *                                                                           in: texture row, u, v                out: rgb
 *   RTX_PROBE_SKYBOX    CubeMap::color (cubemap.rs:45-84)                   in: direction[3]                     out: rgb
 *   RTX_PROBE_BBOX      BBox::intersects (bbox.rs:99-165)                   in: min[3], max[3], origin[3], direction[3]   out: 0 / 1
 * for n argument rows: in[n][16] doubles (unused columns ignored, booleans as 0 / 1), out[n][4] doubles; host arrays. */
enum { RTX_PROBE_FRESNEL = 0, RTX_PROBE_REFRACT = 1, RTX_PROBE_REFLECT = 2, RTX_PROBE_MATERIAL = 3, RTX_PROBE_TEXTURE = 4,
       RTX_PROBE_SKYBOX = 5, RTX_PROBE_BBOX = 6 };
rtx_status rtx_shade_probe(rtx_scene *scene, int32_t which, int64_t n, const double *in, double *out);

/* Diagnostic, needs no device: the octree rtx_scene_create builds for `desc` (Octree::from_iter,
 * octree.rs:15-43), dumped in DFS pre-order with children 0..7: per node 6 doubles (bbox min,
 * max) and 2 ints (number of children: 0 or 8, number of leaf entries); entries_out receives
 * the leaf entries' prim ids in the same order.  depth is octree.rs:29's value (i32::MIN for an
 * empty scene).  Pass NULL for the three arrays to query the sizes only. */
rtx_status rtx_octree_dump(const rtx_scene_desc *desc, int32_t *depth, int64_t *n_nodes, int64_t *n_entries,
                           double *bbox_out, int32_t *shape_out, int32_t *entries_out);

/* Diagnostic: the SAH hierarchy rtx_scene_create built for the scene, copied back from the device as it lies there.  No
 * counterpart in the reference (octree.rs:15-43 builds the octree; this is the structure the walks use beside it).
 * which = 0: the 16-wide box lists (32-byte children: float lo[3], hi[3], uint32 ref, uint32 first_node; 16 per node),
 * which = 1: the 4-wide nodes (112 bytes: float lo[3][4], hi[3][4], uint32 ref[4]).  *n_bytes receives the array's size; the copy is
 * made when out != NULL and cap_bytes >= that size.  tests/test_gpu_build.py compares the device-side builder's arrays with the
 * host builder's (RTX_BUILD_HIERARCHY=host|device). */
rtx_status rtx_hierarchy_dump(rtx_scene *scene, int32_t which, void *out, int64_t cap_bytes, int64_t *n_bytes);

/* Measurement aid, no counterpart in the reference: the kernels the scene's LAST render call launched, by the names rocprofv3 prints,
 * as "closest=k_closest<5>;shade=k_shade<true>;shadow0=k_shadow_coop<false>;shadow1=..." (one shadowN per light; empty before the
 * first render).  bench.py uses it to attribute counter passes and launch times to the kernel that actually ran. */
rtx_status rtx_last_kernel_names(const rtx_scene *scene, char *out, int64_t cap);

const char *rtx_strerror(rtx_status status);
/* Last error text recorded by this thread (HIP error strings, limit details). Never NULL. */
const char *rtx_last_error(void);
/* "rtx <version> gfx950 ..." */
const char *rtx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
