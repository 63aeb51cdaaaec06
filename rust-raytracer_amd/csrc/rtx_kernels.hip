// rtx_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the per-pixel trace loop.
//
// The reference's recursive Renderer::trace (src/raytracer/renderer.rs:105-247) is unrolled
// into a wavefront of ray queues: one ray per lane, an iterative octree traversal whose
// short per-lane level stack lives in LDS, shadow feelers fanned out to their own lanes,
// children (reflection / gloss / transmission) appended to a ray pool with wave-aggregated
// atomics, per-sample f64 accumulators fed by hardware f64 atomic adds.  No MFMA: nothing
// here is a dense contraction.  All arithmetic is IEEE f64 in the reference's operation
// order; the file is compiled with -ffp-contract=off so no a*b+c is ever fused (Rust never
// fuses), which is what keeps hit ids and the EPSILON-level shadow "acne" bit-identical.
//
// Each device function cites the reference file:line it follows (paths relative to src/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rtx_device.h"

namespace {

#define DEV __device__ __forceinline__

constexpr double F64_EPSILON = 2.220446049250313e-16;  // std::f64::EPSILON
constexpr double EPSILON = F64_EPSILON * 10000.0;       // raytracer/renderer.rs:12
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double T_MIN = 0.000001;                      // raytracer/ray.rs:45
DEV double inf() { return __longlong_as_double(0x7FF0000000000000LL); }

// ------------------------------------------------------------------------------ vec3.rs
struct V3 {
    double x, y, z;
};
DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 scale(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }                   // vec3.rs:56
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                // vec3.rs:34
DEV V3 cross(V3 a, V3 b) {                                                               // vec3.rs:38
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEV double len(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }                 // vec3.rs:30
DEV V3 unit(V3 a) {                                                                      // vec3.rs:46
    double l = len(a);
    return v3(a.x / l, a.y / l, a.z / l);
}
DEV V3 reflect(V3 v, V3 n) { return scale(n, 2.0 * dot(n, v)) - v; }                     // vec3.rs:70
DEV bool refract(V3 v, V3 n, double ior, bool inside, V3 *out) {                         // vec3.rs:77-92
    double n1, n2, n_dot_v;
    V3 nn;
    if (!inside) {
        n1 = 1.0; n2 = ior; n_dot_v = dot(n, v); nn = n;
    } else {
        n1 = ior; n2 = 1.0; n_dot_v = -dot(n, v); nn = -n;
    }
    double ratio = n1 / n2;
    double disc = 1.0 - ((ratio * ratio) * (1.0 - n_dot_v * n_dot_v));
    if (disc < 0.0) return false;
    *out = scale(v, -ratio) + scale(nn, ratio * n_dot_v - sqrt(disc));
    return true;
}
// f64::max/min ignore a NaN operand exactly like fmax/fmin (vec3.rs:102-108)
DEV double clamp01(double x) { return fmin(fmax(x, 0.0), 1.0); }

// ------------------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon et al. 2011), the build's own counter-based generator: the
// reference's thread_rng() is OS-seeded and unreproducible (renderer.rs:69, spherelight.rs:14,
// vec3.rs:113).  Spec in DESIGN.md §RNG; the oracle implements the same spec independently.
// the full 32 x 32 -> 64-bit product in ONE instruction (the compiler emits v_mul_hi_u32 + v_mul_lo_u32, two quarter-rate instructions:
// sponza shadows 168.9 -> 167.5 ms with this form; -DRTX_PHILOX_MULHILO keeps the other for A/B)
DEV uint64_t mul_wide(uint32_t a, uint32_t b) {
#ifdef RTX_PHILOX_MULHILO
    return ((uint64_t)__umulhi(a, b) << 32) | (uint64_t)(a * b);
#else
    uint64_t p, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(a), "v"(b));
    return p;
#endif
}
DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = mul_wide(0xD2511F53u, c0), p1 = mul_wide(0xCD9E8D57u, c2);
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
constexpr uint32_t TAG_ROOT = 0x52545831u, TAG_CHILD = 0x4348494Cu, TAG_DRAW = 0x44524157u;
enum { STREAM_JITTER = 0, STREAM_LIGHT = 1, STREAM_GLOSS = 2 };
enum { CODE_MIRROR = 1, CODE_TRANSMIT = 2, CODE_GLOSS0 = 16 };
DEV uint64_t root_key(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sub) {
    uint32_t o[4];
    philox4x32_10(pixel, sub, frame, TAG_ROOT, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return (uint64_t)o[0] | ((uint64_t)o[1] << 32);
}
DEV uint64_t child_key(uint64_t parent, uint32_t code) {
    uint32_t o[4];
    philox4x32_10(code, 0, 0, TAG_CHILD, (uint32_t)parent, (uint32_t)(parent >> 32), o);
    return (uint64_t)o[0] | ((uint64_t)o[1] << 32);
}
DEV void draw2(uint32_t mode, uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *a, double *b) {
    if (mode == RTX_RNG_CENTER) { *a = 0.5; *b = 0.5; return; }
    uint32_t o[4];
    philox4x32_10(stream, i, j, TAG_DRAW, (uint32_t)key, (uint32_t)(key >> 32), o);
    uint64_t ua = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
    uint64_t ub = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
    *a = (double)(ua >> 11) * (1.0 / 9007199254740992.0);
    *b = (double)(ub >> 11) * (1.0 / 9007199254740992.0);
}
// three draws from ONE Philox block, (word + 0.5) / 2^32 each: the light and gloss samplers
DEV void draw3(uint32_t mode, uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *a, double *b, double *c) {
    if (mode == RTX_RNG_CENTER) { *a = 0.5; *b = 0.5; *c = 0.5; return; }
    uint32_t o[4];
    philox4x32_10(stream, i, j, TAG_DRAW, (uint32_t)key, (uint32_t)(key >> 32), o);
    *a = ((double)o[0] + 0.5) * (1.0 / 4294967296.0);
    *b = ((double)o[1] + 0.5) * (1.0 / 4294967296.0);
    *c = ((double)o[2] + 0.5) * (1.0 / 4294967296.0);
}

// ------------------------------------------------------------------------------ raytracer/ray.rs:15-42
struct Ray {
    double ox, oy, oz, dx, dy, dz, ix, iy, iz;
    bool sx, sy, sz;
    uint32_t flip; // child k = j ^ flip visits the octants near-to-far for j = 0..7
};
DEV Ray make_ray(V3 o, V3 d) {
    Ray r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = 1.0 / d.x; r.iy = 1.0 / d.y; r.iz = 1.0 / d.z;
    r.sx = r.ix > 0.0; r.sy = r.iy > 0.0; r.sz = r.iz > 0.0;
    r.flip = (r.sx ? 0u : 4u) | (r.sy ? 0u : 2u) | (r.sz ? 0u : 1u);
    return r;
}

// BBox::intersects, geometry/bbox.rs:99-165, on precomputed slab distances.  Comparisons are
// transcribed literally (a NaN keeps the old bound); t_min/t_max are handed back for the
// near-to-far culling that the reference does not do.
DEV bool slab_combine(double t_min, double t_max, double ty_min, double ty_max, double tz_min, double tz_max,
                      double *t_near, double *t_far) {
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    *t_near = t_min;
    *t_far = t_max;
    return t_min < inf() && t_max > 0.0;
}
DEV bool slab(double minx, double miny, double minz, double maxx, double maxy, double maxz, const Ray &r,
              double *t_near, double *t_far) {
    double t_min = ((r.sx ? minx : maxx) - r.ox) * r.ix;
    double t_max = ((r.sx ? maxx : minx) - r.ox) * r.ix;
    double ty_min = ((r.sy ? miny : maxy) - r.oy) * r.iy;
    double ty_max = ((r.sy ? maxy : miny) - r.oy) * r.iy;
    double tz_min = ((r.sz ? minz : maxz) - r.oz) * r.iz;
    double tz_max = ((r.sz ? maxz : minz) - r.oz) * r.iz;
    return slab_combine(t_min, t_max, ty_min, ty_max, tz_min, tz_max, t_near, t_far);
}

// ------------------------------------------------------------------------------ prims
struct TriV {
    V3 v0, v1, v2;
};
DEV TriV load_tri(const DTri *tris, int idx, int *prim_id, int *material) {
    // 80-byte record, 16-byte aligned: five 16-byte loads
    const double2 *p = reinterpret_cast<const double2 *>(tris + idx);
    double2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriV t;
    t.v0 = v3(a.x, a.y, b.x);
    t.v1 = v3(b.y, c.x, c.y);
    t.v2 = v3(d.x, d.y, e.x);
    long long ids = __double_as_longlong(e.y);
    *prim_id = (int)(ids & 0xFFFFFFFFLL);
    *material = (int)(ids >> 32);
    return t;
}
// the same at a 32-bit byte offset from the array's base (the walks over the SAH hierarchies: the host keeps scenes of 2^24 triangles and more off them)
DEV TriV load_tri32(const DTri *tris, uint32_t idx, int *prim_id, int *material) {
    return load_tri(reinterpret_cast<const DTri *>(reinterpret_cast<const char *>(tris) + idx * (uint32_t)sizeof(DTri)), 0, prim_id, material);
}
// Triangle::intersects up to t, geometry/prims/triangle.rs:111-133 (Möller–Trumbore, two-sided)
DEV bool tri_hit(const TriV &tr, const Ray &r, double t_min, double t_max, double *t_out, double *beta_out,
                 double *gamma_out) {
    V3 d = v3(r.dx, r.dy, r.dz), o = v3(r.ox, r.oy, r.oz);
    V3 e1 = tr.v1 - tr.v0;
    V3 e2 = tr.v2 - tr.v0;
    V3 p = cross(d, e2);
    double det = dot(e1, p);
    if (det > -F64_EPSILON && det < F64_EPSILON) return false;
    double inv_det = 1.0 / det;
    V3 s = o - tr.v0;
    double beta = inv_det * dot(s, p);
    if (beta < 0.0 || beta > 1.0) return false;
    V3 q = cross(s, e1);
    double gamma = inv_det * dot(d, q);
    if (gamma < 0.0 || beta + gamma > 1.0) return false;
    double t = inv_det * dot(e2, q);
    if (t < t_min || t > t_max) return false;
    *t_out = t; *beta_out = beta; *gamma_out = gamma;
    return true;
}
// The leaf entry's own bbox (octree.rs:170) = the prim's partial_bounding_box
// (triangle.rs:102-106: union of the three vertices), recomputed instead of stored.
DEV bool tri_bbox_hit(const TriV &tr, const Ray &r) {
    double tn, tf;
    return slab(fmin(fmin(tr.v0.x, tr.v1.x), tr.v2.x), fmin(fmin(tr.v0.y, tr.v1.y), tr.v2.y),
                fmin(fmin(tr.v0.z, tr.v1.z), tr.v2.z), fmax(fmax(tr.v0.x, tr.v1.x), tr.v2.x),
                fmax(fmax(tr.v0.y, tr.v1.y), tr.v2.y), fmax(fmax(tr.v0.z, tr.v1.z), tr.v2.z), r, &tn, &tf);
}
// Sphere::intersects up to the choice of t, geometry/prims/sphere.rs:36-54.  *t_out is the root
// with the smaller |t| even when that root is outside the window (sphere.rs:54).
DEV bool sphere_hit(V3 c, double radius, const Ray &r, double t_min, double t_max, double *t_out) {
    V3 d = v3(r.dx, r.dy, r.dz);
    V3 i = v3(r.ox, r.oy, r.oz) - c;
    double a_ = 1.0;
    double b_ = 2.0 * dot(d, i);
    double c_ = dot(i, i) - radius * radius;
    double discriminant = b_ * b_ - 4.0 * a_ * c_;
    if (discriminant <= 0.0) return false;
    double disc_sqrt = sqrt(discriminant);
    double t1 = (-b_ + disc_sqrt) / 2.0 * a_;
    double t2 = (-b_ - disc_sqrt) / 2.0 * a_;
    if ((t1 >= t_min && t1 <= t_max) || (t2 >= t_min && t2 <= t_max)) {
        *t_out = fabs(t1) < fabs(t2) ? t1 : t2;
        return true;
    }
    return false;
}
DEV bool sphere_bbox_hit(V3 c, double radius, const Ray &r) {                            // sphere.rs:18-33
    double tn, tf;
    return slab(c.x - radius, c.y - radius, c.z - radius, c.x + radius, c.y + radius, c.z + radius, r, &tn, &tf);
}
DEV bool plane_hit(const DPlane &pl, const Ray &r, double t_min, double t_max, double *t_out) { // plane.rs:26-33
    V3 n = v3(pl.abcd[0], pl.abcd[1], pl.abcd[2]);
    double nrd = dot(n, v3(r.dx, r.dy, r.dz));
    double nro = dot(n, v3(r.ox, r.oy, r.oz));
    double t = (-pl.abcd[3] - nro) / nrd;
    if (t < t_min || t > t_max) return false;
    *t_out = t;
    return true;
}

// ------------------------------------------------------------------------------ octree traversal
// OctreeIterator (raytracer/octree.rs:135-189) restated as an iterative walk with a per-lane
// LEVEL stack in LDS: one 32-bit word per tree level = (first_child << 8) | pending-children
// mask.  Interior nodes always have 8 contiguous children (octree.rs:75-98), so a level needs
// nothing more, and the stack is <= depth words instead of the reference's 1+7*depth pointers.
//
// MODE_CLOSEST:      near-to-far, nodes culled against the best t, terminates when the best hit
//                    lies inside the current leaf's span (result-preserving: Appendix A).
// MODE_CLOSEST_REF:  the reference's exact visitation order (children popped 7..0, no culling),
//                    so exact-t ties between different prims resolve as in ray.rs:52-60.
// MODE_ANY:          shadow feeler when every material is opaque: stop at the first occluder
//                    inside [EPSILON, dist] (the product would be exactly 0 from there on).
// MODE_PRODUCT_FAST: the same multiset as MODE_PRODUCT, boxes classified in f32 (traverse_ww<WW_PRODUCT>).
// MODE_PRODUCT:      shadow feeler with transmissive materials: every candidate of the
//                    multiset, duplicates included (renderer.rs:209-217), order-free product.
// MODE_CLOSEST_BVH / MODE_ANY_BVH: MODE_CLOSEST / MODE_ANY over the 4-wide hierarchy instead of the octree (traverse_bvh).
// MODE_CLOSEST_BVH_DEALT: MODE_CLOSEST_BVH with the exact tests dealt out over the wave (traverse_bvh_dealt): scenes with transmissive materials.
// MODE_PRODUCT_BVH: MODE_PRODUCT_FAST's answer in two phases: the distinct hit prims over the 4-wide hierarchy, then their multiplicities.
enum { MODE_CLOSEST = 0, MODE_CLOSEST_REF = 1, MODE_ANY = 2, MODE_PRODUCT = 3, MODE_PRODUCT_FAST = 4, MODE_CLOSEST_BVH = 5, MODE_ANY_BVH = 6,
       MODE_PRODUCT_BVH = 7, MODE_CLOSEST_BVH_DEALT = 8 };

struct Hit {
    double t;
    int ref;       // >= 0 triangle, < 0 ~sphere, INT_MIN none; planes: 0x40000000 | index
#ifdef RTX_COUNT_WORK
    uint32_t w_nodes, w_entries, w_prims;   // nodes visited, leaf entries fetched, prim tests run
#endif
};
#ifdef RTX_COUNT_WORK
#define WORK(x) x
#else
#define WORK(x)
#endif
constexpr int REF_NONE = (int)0x80000000;
constexpr int REF_PLANE = 0x40000000;
constexpr int REF_RETRY = (int)0x80000001;   // k_closest -> k_closest_retry: this ray takes the all-f64 walk
// DEGENERATE rays: a direction component that is zero or tiny.  Such a ray can lie in a face plane of an octree cell, where the
// reference's own walk loses subtrees: with a zero component and the origin exactly on a node plane BBox::intersects computes
// 0 * inf = NaN and rejects the box (bbox.rs:124-164: a NaN near distance fails the final `t_min < inf`), and within an ulp of a cell
// face the f64 test of the very cell that lists the hit prim can fail.  The conservative f32 walks would reach those leaves and return
// the nearer hit: right geometry, not the reference's answer.  These rays therefore go through the all-f64 restatement of the
// reference's walk (k_closest_retry, k_shadow_retry), which is valid for every ray; the threshold only decides how many take it.
DEV bool degenerate_dir(double dx, double dy, double dz) { return !(fabs(dx) >= 1.0e-9 && fabs(dy) >= 1.0e-9 && fabs(dz) >= 1.0e-9); }

template <int MODE>
DEV void traverse_exact(const DScene &sc, const Ray &r, uint32_t *stk /* LDS, stride RTX_BLOCK */, double t_lo, double t_hi,
                        Hit *hit, V3 *prod) {
    double best_t = t_hi;           // closest: nearest_t (ray.rs:47); shadows: distance_to_light
    int best_ref = REF_NONE;
    V3 acc = v3(1.0, 1.0, 1.0);     // shadow product (renderer.rs:211)
    bool done = false;
    const uint32_t flip = (MODE == MODE_CLOSEST_REF) ? 7u : r.flip;

    WORK(uint32_t w_nodes = 0; uint32_t w_entries = 0; uint32_t w_prims = 0;)
    int sp = 0;
    int cur = 0;                    // root: never tested against the ray (octree.rs:149,177)
    bool have_cur = true;
    double cur_far = -inf();

    while (!done) {
        if (!have_cur) {
            while (sp > 0 && (stk[(sp - 1) * RTX_BLOCK] & 0xFFu) == 0u) --sp;
            if (sp == 0) break;
            uint32_t e = stk[(sp - 1) * RTX_BLOCK];
            uint32_t j = (uint32_t)__ffs((int)(e & 0xFFu)) - 1u;
            stk[(sp - 1) * RTX_BLOCK] = e & ~(1u << j);
            cur = (int)(e >> 8) + (int)(j ^ flip);
        }
        const DNode *np = sc.nodes + cur;
        // 64-byte node record: four 16-byte loads
        const double2 *nq = reinterpret_cast<const double2 *>(np);
        double2 q0 = nq[0], q1 = nq[1], q2 = nq[2];
        int4 meta = *reinterpret_cast<const int4 *>(nq + 3);
        double minx = q0.x, miny = q0.y, minz = q1.x, maxx = q1.y, maxy = q2.x, maxz = q2.y;
        int first_child = meta.x, entry_begin = meta.y, entry_count = meta.z;
        uint32_t child_mask = (uint32_t)meta.w;
        if (!have_cur) {
            if (MODE == MODE_CLOSEST || MODE == MODE_ANY) {
                // the parent already found this box hit; recompute its span to cull it
                double tn, tf;
                slab(minx, miny, minz, maxx, maxy, maxz, r, &tn, &tf);
                if (tn > best_t) continue;
                cur_far = tf;
            }
        }
        have_cur = false;
        WORK(++w_nodes;)

        if (first_child >= 0) {
            // children's boxes exactly as OctreeNode::subdivide computes them (octree.rs:79-92):
            // low half = [min, max - len/2], high half = [min + len/2, max]
            double hx = (maxx - minx) / 2.0, hy = (maxy - miny) / 2.0, hz = (maxz - minz) / 2.0;
            double mlx = maxx - hx, mhx = minx + hx;
            double mly = maxy - hy, mhy = miny + hy;
            double mlz = maxz - hz, mhz = minz + hz;
            // slab distances of the 4 planes per axis, arranged as (near half tmin, tmax), (far half tmin, tmax)
            double ax0 = ((r.sx ? minx : maxx) - r.ox) * r.ix, ax1 = ((r.sx ? mlx : mhx) - r.ox) * r.ix;
            double ax2 = ((r.sx ? mhx : mlx) - r.ox) * r.ix, ax3 = ((r.sx ? maxx : minx) - r.ox) * r.ix;
            double ay0 = ((r.sy ? miny : maxy) - r.oy) * r.iy, ay1 = ((r.sy ? mly : mhy) - r.oy) * r.iy;
            double ay2 = ((r.sy ? mhy : mly) - r.oy) * r.iy, ay3 = ((r.sy ? maxy : miny) - r.oy) * r.iy;
            double az0 = ((r.sz ? minz : maxz) - r.oz) * r.iz, az1 = ((r.sz ? mlz : mhz) - r.oz) * r.iz;
            double az2 = ((r.sz ? mhz : mlz) - r.oz) * r.iz, az3 = ((r.sz ? maxz : minz) - r.oz) * r.iz;
            uint32_t pending = 0;
            if (MODE == MODE_CLOSEST_REF) {
                // reference order: j = 7 - k; the near/far arrangement above is indexed by the ray's
                // own flip, so translate k -> near-order bits with r.flip
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    if (!((child_mask >> k) & 1u)) continue;
                    uint32_t b = k ^ r.flip;
                    double tn, tf;
                    bool ok = slab_combine((b & 4u) ? ax2 : ax0, (b & 4u) ? ax3 : ax1, (b & 2u) ? ay2 : ay0,
                                           (b & 2u) ? ay3 : ay1, (b & 1u) ? az2 : az0, (b & 1u) ? az3 : az1, &tn, &tf);
                    if (ok) pending |= 1u << (k ^ 7u);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    if (!((child_mask >> (j ^ flip)) & 1u)) continue;
                    double tn, tf;
                    bool ok = slab_combine((j & 4u) ? ax2 : ax0, (j & 4u) ? ax3 : ax1, (j & 2u) ? ay2 : ay0,
                                           (j & 2u) ? ay3 : ay1, (j & 1u) ? az2 : az0, (j & 1u) ? az3 : az1, &tn, &tf);
                    if (MODE == MODE_CLOSEST || MODE == MODE_ANY) ok = ok && !(tn > best_t);
                    if (ok) pending |= 1u << j;
                }
            }
            if (pending) {
                stk[sp * RTX_BLOCK] = ((uint32_t)first_child << 8) | pending;
                ++sp;
            }
        } else {
            for (int e = entry_begin; e < entry_begin + entry_count; ++e) {
                int ref = sc.leaf_refs[e];
                WORK(++w_entries;)
                if (ref >= 0) {
                    int pid, mat;
                    TriV tr = load_tri(sc.tris, ref, &pid, &mat);
                    if (!tri_bbox_hit(tr, r)) continue;
                    WORK(++w_prims;)
                    double t, be, ga;
                    if (MODE == MODE_CLOSEST || MODE == MODE_CLOSEST_REF) {
                        // prim.intersects(ray, t_min, nearest_t) then `t > t_min && t < nearest_t` (ray.rs:50-55)
                        if (tri_hit(tr, r, t_lo, best_t, &t, &be, &ga) && t > t_lo && t < best_t) {
                            best_t = t;
                            best_ref = ref;
                        }
                    } else if (tri_hit(tr, r, t_lo, t_hi, &t, &be, &ga)) {
                        if (MODE == MODE_ANY) { best_ref = ref; done = true; break; }
                        const rtx_material &m = sc.materials[mat];
                        if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                        else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                        if (acc.x == 0.0 && acc.y == 0.0 && acc.z == 0.0) { done = true; break; }
                    }
                } else {
                    const DSphere &sp_ = sc.spheres[~ref];
                    V3 c = v3(sp_.c[0], sp_.c[1], sp_.c[2]);
                    double radius = sp_.r;
                    if (!sphere_bbox_hit(c, radius, r)) continue;
                    WORK(++w_prims;)
                    double t;
                    if (MODE == MODE_CLOSEST || MODE == MODE_CLOSEST_REF) {
                        if (sphere_hit(c, radius, r, t_lo, best_t, &t) && t > t_lo && t < best_t) {
                            best_t = t;
                            best_ref = ref;
                        }
                    } else if (sphere_hit(c, radius, r, t_lo, t_hi, &t)) {
                        if (MODE == MODE_ANY) { best_ref = ref; done = true; break; }
                        const rtx_material &m = sc.materials[sp_.material];
                        if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                        else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                        if (acc.x == 0.0 && acc.y == 0.0 && acc.z == 0.0) { done = true; break; }
                    }
                }
            }
            // near-to-far: everything still pending starts behind this leaf's exit
            if (MODE == MODE_CLOSEST && best_ref != REF_NONE && best_t <= cur_far) done = true;
        }
    }

    // infinite prims last, in order (octree.rs:163-165,185-186)
    if (!(MODE == MODE_ANY && done) && !(MODE == MODE_PRODUCT && done)) {
        for (int p = 0; p < sc.n_planes; ++p) {
            const DPlane &pl = sc.planes[p];
            double t;
            if (MODE == MODE_CLOSEST || MODE == MODE_CLOSEST_REF) {
                if (plane_hit(pl, r, t_lo, best_t, &t) && t > t_lo && t < best_t) {
                    best_t = t;
                    best_ref = REF_PLANE | p;
                }
            } else if (plane_hit(pl, r, t_lo, t_hi, &t)) {
                if (MODE == MODE_ANY) { best_ref = REF_PLANE | p; break; }
                const rtx_material &m = sc.materials[pl.material];
                if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
            }
        }
    }
    hit->t = best_t;
    hit->ref = best_ref;
    WORK(hit->w_nodes = w_nodes; hit->w_entries = w_entries; hit->w_prims = w_prims;)
    if (MODE == MODE_ANY) *prod = (best_ref == REF_NONE) ? v3(1.0, 1.0, 1.0) : v3(0.0, 0.0, 0.0);
    else *prod = acc;
}

// Fast traversals (closest hit and per-lane any hit: traverse_ww, cooperative any hit: k_shadow_coop, shadow product: traverse_ww<WW_PRODUCT>).  Node and leaf-entry boxes are tested in f32
// against the outward-rounded DNode32 / DEntry32 twins, widened by a per-ray delta that bounds
// every rounding error of the f32 arithmetic (origin and inverse direction rounded to f32, one
// subtraction and one multiplication per plane: < 2^-22 * S in position units, S = largest
// coordinate magnitude involved; delta = 2^-20 * S).  The f32 test is therefore a SUPERSET of the
// reference's f64 BBox::intersects: it can only let extra candidates through, and every
// candidate still passes the exact f64 entry-bbox test (octree.rs:170) and the exact primitive
// test before it counts.  NaNs (0 * inf on axis-parallel rays) drop out of v_max3/v_min3, which
// only weakens the cull.  Near-to-far order by direction sign is exact for octree siblings.
// Where the ray enters the cube [centre +- radius] that holds every finite primitive.  The f32
// frame of the fast traversals is re-based there (o' = o + t_shift * d): a ray that starts far
// outside the scene (a ground plane seen near the horizon puts hit points 1e6 units away) would
// otherwise need a delta so large that every box is "hit".  A ray that misses the cube (widened
// by 1e-6) has no finite candidate at all.  Results are unaffected: only the conservative culling
// depends on it, the exact tests keep using the original ray.
struct RootEntry {
    double t_shift;
    bool miss;
};
DEV RootEntry root_entry(const DScene &sc, const Ray &r, double orx, double ory, double orz) {
    const double R = sc.radius * 1.000001;
    RootEntry e;
    // an origin inside the cube (every reflection ray and shadow feeler that starts on a finite prim): each axis has one plane
    // behind the origin, so tin stays 0 and the ray cannot miss; the slab distances are computed only if some lane is outside
    if (fabs(orx) <= R && fabs(ory) <= R && fabs(orz) <= R) {
        e.miss = false;
        e.t_shift = 0.0;
        return e;
    }
    double tin = 0.0, tout = inf();
    bool miss = false;
    const double oo[3] = {orx, ory, orz}, dd[3] = {r.dx, r.dy, r.dz}, ii[3] = {r.ix, r.iy, r.iz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] != 0.0) {
            double t1 = (-R - oo[a]) * ii[a], t2 = (R - oo[a]) * ii[a];
            tin = fmax(tin, fmin(t1, t2));
            tout = fmin(tout, fmax(t1, t2));
        } else if (fabs(oo[a]) > R) {
            miss = true;
        }
    }
    e.miss = miss || tin > tout;
    e.t_shift = tin;     // >= 0
    return e;
}
DEV float f32_above(double v) {   // a float >= v (v finite or +inf)
    float f = (float)v;
    return f + fabsf(f) * 1.1920929e-7f;
}
// LDS layout of the traversal kernels: the per-lane level stacks, then a 2 KB table that maps a
// node's child mask into near-to-far order for each of the 8 direction-sign patterns:
// perm[flip][mask] bit j = mask bit (j ^ flip).
#define RTX_STK_WORDS (RTX_MAX_LEVELS * RTX_BLOCK)
#define RTX_LDS_WORDS (RTX_STK_WORDS + 512)
DEV void init_lds(uint32_t *s_lds) {
    uint8_t *perm = reinterpret_cast<uint8_t *>(s_lds + RTX_STK_WORDS);
    for (uint32_t i = threadIdx.x; i < 2048u; i += RTX_BLOCK) {
        uint32_t flip = i >> 8, m = i & 0xFFu, o = 0;
        for (uint32_t j = 0; j < 8u; ++j) o |= ((m >> (j ^ flip)) & 1u) << j;
        perm[i] = (uint8_t)o;
    }
    __syncthreads();
}
// Shadow PRODUCT over the reference's exact candidate multiset (scenes with transmissive materials,
// renderer.rs:209-217), at f32 speed.  The product needs every primitive once per leaf that the reference's
// walk reaches and that lists it (no culling by distance; BBox::intersects, bbox.rs:99-165, has no ray range, only
// "in front of the origin").  Boxes are classified in f32:  the box widened by delta missed => the f64 test fails;
// the box shrunk by delta passed (cross-axis form, surely_hit) => the f64 test passes;  the few in between (a ray
// within delta of a face or an edge) are decided by the f64 test itself on the stored f64 box.  Because child boxes
// are nested exactly and the test is monotone in the box (see traverse_ww, leaf branch), only LEAF boxes need the
// exact decision; interior nodes are descended on the widened test.  Leaf entries go the same way (f32 reject,
// surely-in, or the exact entry-bbox test), then the exact primitive test.  The walk itself is
// traverse_ww<WW_PRODUCT> below; traverse_exact<MODE_PRODUCT> remains as the all-f64 restatement
// (rtx_shadow_products, RTX_FLAG_REFERENCE_ORDER, axis-parallel feelers) the tests compare this walk with.
DEV bool surely_hit(float nx, float ny, float nz, float fx, float fy, float fz) {
    // n*: upper bounds of the near slab distances, f*: lower bounds of the far ones.  BBox::intersects passes iff every
    // cross-axis pair is ordered and the smallest far distance is > 0 (an axis is never compared with itself).
    return nx <= fminf(fy, fz) && ny <= fminf(fx, fz) && nz <= fminf(fx, fy) && fminf(fminf(fx, fy), fz) > 0.0f;
}
// Closest hit as a WHILE-WHILE walk with DEFERRED exact tests (k_closest, k_nearest).  Same nodes, same
// conservative f32 tests and same exact f64 tests as the reference's walk, arranged so that the lanes of
// a wave do the same kind of work at the same time:
//   phase 1  every lane descends the octree towards its next leaf (a bounded number of nodes per round);
//   phase 2  every lane scans its leaf's entries; a candidate that passes the f32 entry test is PARKED in a
//            small per-lane LDS buffer instead of being tested on the spot;
//   phase 3  when a buffer is full, enough lanes hold candidates or nobody walks any more, the wave tests the
//            parked candidates, every lane one of its OWN per round, in visiting order.
// With "node or whole leaf per iteration" and tests in place, one lane's leaf loop or 200-instruction f64 test
// held the other 63 lanes (measured: 29 % of the lanes active).  Between flushes a lane walks on with a stale
// best t: that only weakens the culling (Appendix A); a lane finishes after a flush, when best t <= exit of
// the last leaf it has completed (near-to-far order: everything still pending starts behind that exit).
// Must be called by every lane of the wave that has not left the kernel; `active` = the lane has a ray.
#ifndef RTX_DEFER_CAP
#define RTX_DEFER_CAP 4       // candidates a lane can park
#endif
#ifndef RTX_DEFER_FLUSH
#define RTX_DEFER_FLUSH 24    // flush once this many lanes hold parked candidates
#endif
// Per-lane MAILBOX behind the parked candidates: a direct-mapped set of RTX_MAILBOX prim refs (slot = ref mod RTX_MAILBOX).  The entry-bbox
// test and the prim test do not depend on the leaf that lists the prim, and the reference's octree lists a prim in every leaf its box
// overlaps (heptoroid stand-in: 12.5 leaves per triangle, 47 exact tests per shadow feeler of which most repeat an earlier one), so:
//   closest / any hit: a prim enters the set when it is parked; a later copy would be tested with a window at most as wide and is skipped;
//   shadow product:    a prim enters the set when its exact tests have FAILED: its copies contribute nothing either.  Prims that were hit
//                      are tested again in every leaf (the product counts each copy, renderer.rs:209-217).
#ifndef RTX_MAILBOX
#define RTX_MAILBOX 8
#endif
#define RTX_CAND_WORDS ((RTX_DEFER_CAP + RTX_MAILBOX) * RTX_BLOCK)
#ifndef RTX_DESCENT_STEPS
#define RTX_DESCENT_STEPS 4   // nodes a lane may visit per round before the wave moves on to the entry scans
#endif
#ifndef RTX_ENTRY_BATCH
#define RTX_ENTRY_BATCH 4     // leaf entries fetched together
#endif
// ANY = true: the any-hit search of a shadow feeler (k_shadow<MODE_ANY>): the window is [t_lo, t_hi] for every candidate and
// the lane stops at the first flush that finds an occluder.
// WW_PRODUCT: the shadow product over the reference's exact candidate multiset (classification of boxes: see surely_hit above): nothing is
// culled by distance, the mailbox holds known misses only, every hit multiplies `*prod`; *fallback = the caller must run the all-f64 walk.
enum { WW_CLOSEST = 0, WW_ANY = 1, WW_PRODUCT = 2 };
template <int WW>
DEV void traverse_ww(const DScene &sc, const Ray &r, bool active, uint32_t *s_lds, uint32_t *s_cand, double t_lo, double t_hi, Hit *hit,
                     V3 *prod = nullptr, bool *fallback = nullptr, bool *tie_out = nullptr) {
    constexpr bool ANY = WW == WW_ANY, PRODUCT = WW == WW_PRODUCT;
    // TIES: two different prims hit at a bit-identical t (a ray through the shared edge or vertex of two triangles).  ray.rs:52-60 keeps the
    // one it VISITS first (`t < nearest_t` is strict), and its visiting order is the octree's (children popped 7 -> 0, a leaf's entries in
    // insertion order, octree.rs:177-187), which this walk does not follow.  It only notices the tie: the ray is then answered by the
    // all-f64 walk in the reference's order (k_closest_retry / traverse_exact<MODE_CLOSEST_REF>), like a degenerate ray.  A tie that a
    // smaller t supersedes is forgotten.  Planes come last in the reference too (octree.rs:163-165): a finite prim wins a tie with one.
    bool tie = false;
    V3 acc = v3(1.0, 1.0, 1.0);   // WW_PRODUCT: renderer.rs:211
    bool zero = false;            // ... has reached exactly (0,0,0)
    uint32_t *stk = s_lds + threadIdx.x;     // level stack, stride RTX_BLOCK
    uint32_t *cand = s_cand + threadIdx.x;   // parked candidates, stride RTX_BLOCK: (ref << 1) | surely_in
    const uint8_t *perm = reinterpret_cast<const uint8_t *>(s_lds + RTX_STK_WORDS);
    double best_t = ANY ? t_hi : inf();   // closest: nearest_t (ray.rs:47); any-hit: distance_to_light, never lowered
    int best_ref = REF_NONE;
    WORK(uint32_t w_nodes = 0; uint32_t w_entries = 0; uint32_t w_prims = 0;)

    const RootEntry re = root_entry(sc, r, r.ox - sc.centre[0], r.oy - sc.centre[1], r.oz - sc.centre[2]);
    const double tsh = re.t_shift;   // every f32 slab distance below is t - tsh
    const double orx = r.ox - sc.centre[0] + tsh * r.dx, ory = r.oy - sc.centre[1] + tsh * r.dy, orz = r.oz - sc.centre[2] + tsh * r.dz;
    const double big = fmax(fmax(fabs(orx), fabs(ory)), fabs(orz)) + sc.radius;
    const double delta = big * (1.0 / 1048576.0);
    const bool sx = r.sx, sy = r.sy, sz = r.sz;
    const float ix = (float)r.ix, iy = (float)r.iy, iz = (float)r.iz;
    const float hx = f32_above(delta * fabs(r.ix)), hy = f32_above(delta * fabs(r.iy)), hz = f32_above(delta * fabs(r.iz));
    const float cx = (float)(-(orx * r.ix)), cy = (float)(-(ory * r.iy)), cz = (float)(-(orz * r.iz));
    const float cnx = cx - hx, cfx = cx + hx, cny = cy - hy, cfy = cy + hy, cnz = cz - hz, cfz = cz + hz;
    const bool finite_dir = fabsf(ix) < 3.0e38f && fabsf(iy) < 3.0e38f && fabsf(iz) < 3.0e38f;
    const uint32_t flip = r.flip;
    const uint8_t *perm_row = perm + (flip << 8);
    float bf = PRODUCT ? __int_as_float(0x7F800000) : f32_above(best_t - tsh);   // the product culls nothing by distance
    uint32_t *mbx = s_cand + RTX_DEFER_CAP * RTX_BLOCK + threadIdx.x;   // mailbox slots, stride RTX_BLOCK
#pragma unroll
    for (uint32_t k = 0; k < RTX_MAILBOX; ++k) mbx[k * RTX_BLOCK] = (uint32_t)REF_NONE;

    int sp = 0;
    bool at_root = true;                       // root: never culled (octree.rs:149,177)
    bool running = active && !re.miss;         // a ray that misses the scene's cube has no finite candidate
    // WW_PRODUCT: a ray with an infinite inverse direction component (0 * inf = NaN slab distances) goes to the caller's all-f64 walk
    const bool fb = PRODUCT && active && !finite_dir;
    if (fb) running = false;
    uint32_t cnt = 0;                          // parked candidates
    uint32_t e_cur = 0, e_end = 0;             // position in the current leaf's entry list (absolute indices)
    const float NEG_INF = __int_as_float(0xFF800000);
    float cur_exit = NEG_INF, last_exit = NEG_INF;   // lower bounds of the exit distance of the current / last completed leaf

    for (;;) {
        if (!__any(running || cnt > 0u)) break;
        // ---- phase 1: descend towards the next non-empty leaf, at most RTX_DESCENT_STEPS nodes per round: descents differ
        // in length by 3x between the lanes of a wave (culled nodes, depth), and lanes that already stand in a leaf
        // must not wait for the longest one (uncapped: 80 -> capped at 4: 65 ms per sponza frame)
        for (int step = 0; step < RTX_DESCENT_STEPS && running && e_cur >= e_end; ++step) {
            uint32_t cur = 0;
            if (!at_root) {
                while (sp > 0 && (stk[(sp - 1) * RTX_BLOCK] & 0xFFu) == 0u) --sp;
                if (sp == 0) { running = false; break; }
                uint32_t e = stk[(sp - 1) * RTX_BLOCK];
                uint32_t j = (uint32_t)__ffs((int)(e & 0xFFu)) - 1u;
                stk[(sp - 1) * RTX_BLOCK] = e & ~(1u << j);
                cur = (e >> 8) + (j ^ flip);
            }
            const float4 *nq = reinterpret_cast<const float4 *>(sc.nodes32 + cur);
            const float4 qa = nq[0], qb = nq[1];
            const float minx = qa.x, miny = qa.y, minz = qa.z, maxx = qa.w, maxy = qb.x, maxz = qb.y;
            const uint32_t link = __float_as_uint(qb.z), aux = __float_as_uint(qb.w);
            const float pnx = sx ? minx : maxx, pfx = sx ? maxx : minx;
            const float pny = sy ? miny : maxy, pfy = sy ? maxy : miny;
            const float pnz = sz ? minz : maxz, pfz = sz ? maxz : minz;
            const float t0x = fmaf(pnx, ix, cnx), t0y = fmaf(pny, iy, cny), t0z = fmaf(pnz, iz, cnz);
            // the parent found this box hit; cull it if it now starts behind the best hit
            const bool culled = !at_root && fmaxf(fmaxf(t0x, t0y), t0z) > bf;
            at_root = false;
            if (culled) continue;
            WORK(++w_nodes;)
            if (aux == 0xFFFFFFFFu) {
                // child boxes = halves of this box (octree.rs:79-92); the f64 mid planes max - len/2 and min + len/2 differ from
                // 0.5*(min+max) by rounding only, far inside delta.  x carries the clamps (entry >= 0, exit <= bf)
                const float mx = 0.5f * (minx + maxx), my = 0.5f * (miny + maxy), mz = 0.5f * (minz + maxz);
                const float n0x = fmaxf(t0x, 0.0f), n1x = fmaxf(fmaf(mx, ix, cnx), 0.0f);
                const float f0x = fminf(fmaf(mx, ix, cfx), bf), f1x = fminf(fmaf(pfx, ix, cfx), bf);
                const float n0y = t0y, n1y = fmaf(my, iy, cny), f0y = fmaf(my, iy, cfy), f1y = fmaf(pfy, iy, cfy);
                const float n0z = t0z, n1z = fmaf(mz, iz, cnz), f0z = fmaf(mz, iz, cfz), f1z = fmaf(pfz, iz, cfz);
                uint32_t pending = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    float lo = fmaxf(fmaxf((j & 4u) ? n1x : n0x, (j & 2u) ? n1y : n0y), (j & 1u) ? n1z : n0z);
                    float hi = fminf(fminf((j & 4u) ? f1x : f0x, (j & 2u) ? f1y : f0y), (j & 1u) ? f1z : f0z);
                    pending |= (lo <= hi) ? (1u << j) : 0u;
                }
                pending &= (uint32_t)perm_row[link & 0xFFu];
                if (pending) {
                    stk[sp * RTX_BLOCK] = (link & 0xFFFFFF00u) | pending;
                    ++sp;
                }
            } else {
                // exit taken with delta the OTHER way (cn on a far plane): a lower bound of the true exit
                cur_exit = finite_dir ? fminf(fminf(fmaf(pfx, ix, cnx), fmaf(pfy, iy, cny)), fmaf(pfz, iz, cnz)) : NEG_INF;
                uint32_t n_entries = aux;
                if constexpr (PRODUCT) {
                    // The product needs EXACTLY the leaves the reference's walk reaches.  A child's box lies inside its parent's
                    // (octree.rs:81-92: min or min + len/2, max or max - len/2) and BBox::intersects is monotone in the box when no
                    // slab distance is a NaN (rounding of (bound - o) * inv is monotone in bound; bbox.rs:99-165 only compares near
                    // with far distances), so for a ray with a finite inverse direction a leaf is reached iff ITS OWN box passes:
                    // interior nodes are descended on the widened f32 test alone, and the exact decision is taken here, once per
                    // leaf: shrunk box passed => reached; otherwise the reference's f64 test on the leaf's stored box.  The root
                    // is never tested (octree.rs:149,177).
                    if (cur != 0u && aux != 0u &&
                        !surely_hit(fmaf(pnx, ix, cfx), fmaf(pny, iy, cfy), fmaf(pnz, iz, cfz), fmaf(pfx, ix, cnx), fmaf(pfy, iy, cny),
                                    fmaf(pfz, iz, cnz))) {
                        const double2 *dq = reinterpret_cast<const double2 *>(sc.nodes + cur);
                        const double2 q0 = dq[0], q1 = dq[1], q2 = dq[2];
                        double tn, tf;
                        if (!slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, r, &tn, &tf)) n_entries = 0u;
                    }
                }
                e_cur = link;
                e_end = link + n_entries;
                if (n_entries == 0u) last_exit = cur_exit;
            }
        }
        // ---- phase 2: the leaf's entries, until it is finished or the lane's buffer is full.  The walk is bound by
        // memory latency (one dependent load per step), so the boxes of up to RTX_ENTRY_BATCH entries are fetched
        // together before any of them is tested.
        while (running && e_cur < e_end && cnt < RTX_DEFER_CAP) {
            float4 ea_[RTX_ENTRY_BATCH], eb_[RTX_ENTRY_BATCH];
            const uint32_t nb = e_end - e_cur;
#pragma unroll
            for (uint32_t k = 0; k < RTX_ENTRY_BATCH; ++k) {
                const float4 *eq = reinterpret_cast<const float4 *>(sc.entries32 + e_cur + (k < nb ? k : 0u));
                ea_[k] = eq[0];
                eb_[k] = eq[1];
            }
#pragma unroll
            for (uint32_t k = 0; k < RTX_ENTRY_BATCH; ++k) {
                if (k < nb && cnt < RTX_DEFER_CAP) {
                    const float4 ea = ea_[k], eb = eb_[k];
                    ++e_cur;
                    WORK(++w_entries;)
                    // conservative f32 version of the entry's own bbox test (octree.rs:170)
                    const float enx = sx ? ea.x : ea.w, efx = sx ? ea.w : ea.x;
                    const float eny = sy ? ea.y : eb.x, efy = sy ? eb.x : ea.y;
                    const float enz = sz ? ea.z : eb.y, efz = sz ? eb.y : ea.z;
                    const float lo = fmaxf(fmaxf(fmaxf(fmaf(enx, ix, cnx), 0.0f), fmaf(eny, iy, cny)), fmaf(enz, iz, cnz));
                    const float hi = fminf(fminf(fminf(fmaf(efx, ix, cfx), bf), fmaf(efy, iy, cfy)), fmaf(efz, iz, cfz));
                    const int ref = __float_as_int(eb.z);
                    // mailbox: already parked or tested with a window at least as wide
                    uint32_t *mslot = mbx + ((uint32_t)ref & (RTX_MAILBOX - 1u)) * RTX_BLOCK;
                    if (lo <= hi && *mslot != (uint32_t)ref) {
                        if (!PRODUCT) *mslot = (uint32_t)ref;
                        // the box shrunk by delta instead of widened: a ray through it passes the exact f64 entry
                        // test with a margin far above f64 rounding, so that test can be skipped
                        bool surely_in = false;
                        if (finite_dir) {
                            float li = fmaxf(fmaxf(fmaf(enx, ix, cfx), fmaf(eny, iy, cfy)), fmaf(enz, iz, cfz));
                            float hi2 = fminf(fminf(fmaf(efx, ix, cnx), fmaf(efy, iy, cny)), fmaf(efz, iz, cnz));
                            surely_in = li <= hi2 && hi2 > 0.0f;
                        }
                        cand[cnt * RTX_BLOCK] = ((uint32_t)ref << 1) | (surely_in ? 1u : 0u);
                        ++cnt;
                    }
                }
            }
        }
        if (running && e_cur >= e_end) last_exit = cur_exit;   // leaf completed
        // ---- phase 3: flush
        const bool stalled = running && e_cur < e_end && cnt >= RTX_DEFER_CAP;   // buffer full in mid-leaf
        const uint32_t holders = (uint32_t)__popcll(__ballot(cnt > 0u));
        if (__any(stalled) || holders >= RTX_DEFER_FLUSH || !__any(running)) {
            for (uint32_t i = 0; __any(i < cnt); ++i) {
                if (i < cnt && !(ANY && best_ref != REF_NONE) && !zero) {   // any-hit: one occluder is enough
                    const uint32_t c = cand[i * RTX_BLOCK];
                    const int ref = (int)c >> 1;
                    const bool surely_in = (c & 1u) != 0u;
                    if (ref >= 0) {
                        int pid, mat;
                        TriV tr = load_tri(sc.tris, ref, &pid, &mat);
                        bool occl = false;   // product: both the entry-bbox test and the prim test passed
                        if (surely_in || tri_bbox_hit(tr, r)) {
                            WORK(++w_prims;)
                            double t, be, ga;
                            if (PRODUCT) {
                                if (tri_hit(tr, r, t_lo, t_hi, &t, &be, &ga)) {
                                    const rtx_material &m = sc.materials[mat];
                                    if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                                    else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                                    zero = acc.x == 0.0 && acc.y == 0.0 && acc.z == 0.0;   // finite factors: it stays exactly 0
                                    occl = true;
                                }
                            } else if (ANY) {   // renderer.rs:212: t in [EPSILON, distance_to_light]
                                if (tri_hit(tr, r, t_lo, t_hi, &t, &be, &ga)) best_ref = ref;
                            } else if (tri_hit(tr, r, t_lo, best_t, &t, &be, &ga) && t > t_lo) {
                                // prim.intersects(ray, t_min, nearest_t) then `t > t_min && t < nearest_t` (ray.rs:50-55)
                                if (t < best_t) { best_t = t; best_ref = ref; tie = false; }
                                else if (t == best_t && ref != best_ref) tie = true;   // see `tie` above
                            }
                        }
                        if (PRODUCT && !occl) mbx[((uint32_t)ref & (RTX_MAILBOX - 1u)) * RTX_BLOCK] = (uint32_t)ref;   // its copies miss too
                    } else {
                        const DSphere &sp_ = sc.spheres[~ref];
                        V3 cc = v3(sp_.c[0], sp_.c[1], sp_.c[2]);
                        double radius = sp_.r;
                        bool occl = false;
                        if (surely_in || sphere_bbox_hit(cc, radius, r)) {
                            WORK(++w_prims;)
                            double t;
                            if (PRODUCT) {
                                if (sphere_hit(cc, radius, r, t_lo, t_hi, &t)) {
                                    const rtx_material &m = sc.materials[sp_.material];
                                    if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                                    else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                                    zero = acc.x == 0.0 && acc.y == 0.0 && acc.z == 0.0;
                                    occl = true;
                                }
                            } else if (ANY) {
                                if (sphere_hit(cc, radius, r, t_lo, t_hi, &t)) best_ref = ref;
                            } else if (sphere_hit(cc, radius, r, t_lo, best_t, &t) && t > t_lo) {
                                if (t < best_t) { best_t = t; best_ref = ref; tie = false; }
                                else if (t == best_t && ref != best_ref) tie = true;
                            }
                        }
                        if (PRODUCT && !occl) mbx[((uint32_t)ref & (RTX_MAILBOX - 1u)) * RTX_BLOCK] = (uint32_t)ref;
                    }
                }
            }
            cnt = 0u;
            if constexpr (PRODUCT) {
                if (zero) running = false;
            } else {
                bf = f32_above(best_t - tsh);
                // (last_exit is the far distance of the leaf's box SHRUNK by delta: a hit on the far face itself does not end the walk, so
                // a prim that ties with it and is listed in the next leaf only is still met)
                if (ANY ? best_ref != REF_NONE : (best_ref != REF_NONE && best_t - tsh <= (double)last_exit)) running = false;
            }
        }
    }

    // infinite prims last, in order (octree.rs:163-165,185-186)
    if (active) {
        for (int p = 0; p < sc.n_planes; ++p) {
            const DPlane &pl = sc.planes[p];
            double t;
            if (PRODUCT) {
                if (!zero && plane_hit(pl, r, t_lo, t_hi, &t)) {
                    const rtx_material &m = sc.materials[pl.material];
                    if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                    else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                }
            } else if (ANY) {
                if (best_ref == REF_NONE && plane_hit(pl, r, t_lo, t_hi, &t)) best_ref = REF_PLANE | p;
            } else if (plane_hit(pl, r, t_lo, best_t, &t) && t > t_lo && t < best_t) {
                best_t = t;
                best_ref = REF_PLANE | p;
            }
        }
    }
    hit->t = best_t;
    hit->ref = best_ref;
    if (tie_out) *tie_out = tie;
    if (PRODUCT) { *prod = acc; *fallback = fb; }
    WORK(hit->w_nodes = w_nodes; hit->w_entries = w_entries; hit->w_prims = w_prims;)
}

// Closest hit / any hit over the 4-WIDE HIERARCHY (DNode4), one ray per lane: the walk of k_closest, k_nearest and k_shadow<MODE_ANY>
// in scenes whose hierarchy was built (DScene.bvh4).  Both searches depend only on the SET of primitives that truly intersect the
// ray, each decided by the reference's exact f64 entry-bbox and primitive tests (octree.rs:170, ray.rs:50-55 / renderer.rs:212); the
// walk only has to reach a superset of them, and any conservative structure over the finite prims does (the
// argument the box lists of the cooperative shadow walk rest on).  The reference's octree lists a prim in every leaf its box
// overlaps (5 entries per triangle in the sponza stand-in) and a lane's walk over it fetched 26 nodes and 33 leaf entries per ray;
// the SAH tree holds every prim once: 14 node visits per ray, no mailbox.
//   Same three phases as traverse_ww (bounded descent, candidates PARKED in LDS, wave-wide flushes with the exact tests), same f32
// frame and margins.  A visit fetches the near and far planes of the four children with six 16-byte loads whose row (lo or hi) the
// lane picks by its direction signs, tests the four boxes, parks the prim children that are hit and continues with the nearest
// node child; the other node children go onto the lane's stack in LDS together with their entry distance, quantised DOWNWARDS into
// the bits the node index leaves free, so that a pop can drop what lies behind the best hit by then.  A full stack (never seen:
// 16 entries, 3.4 needed on average) sets *overflow and ends the walk: the caller hands the ray to the all-f64 walk.
#ifndef RTX_BVH_STACK
#define RTX_BVH_STACK 16     // (3 in the forced-overflow test build, tests/test_gpu_variants.py)
#endif
#define RTX_BVH_CAND 8        // candidates a lane can park; a visit needs room for 4
#ifndef RTX_BVH_STEPS
#define RTX_BVH_STEPS 6       // node visits per lane and round before the wave looks at the parked candidates (2: +30 %, 3..8: within 1.5 %)
#endif
#ifndef RTX_BVH_FLUSH
#define RTX_BVH_FLUSH 24
#endif
#ifndef RTX_BVH_WAVES
#define RTX_BVH_WAVES 5       // waves per SIMD of the kernels built on this walk: 96 VGPRs (at 6: 32 of them spilled, closest + shade 40.8 -> 37.1 ms)
#endif
//   WW_PRODUCT: the shadow product of scenes with transmissive materials (renderer.rs:209-217), in two phases.  The reference multiplies
// transmission() once per (reached leaf, entry of the leaf whose own bbox and primitive tests pass): the product is  PROD over the
// distinct prims p the feeler hits of  T(p) ^ m(p),  m(p) = number of leaves that list p and that the reference's walk reaches.  Phase 1
// finds the distinct hit prims with this walk (every candidate in [t_lo, t_hi], no best hit, no early end; refs kept in a per-lane LDS list);
// phase 2 counts, for each of them, the leaves listing it (DScene.prim_leaves, the inverse of the octree's leaf lists) that the walk
// reaches: for a feeler with a finite inverse direction that is "the leaf's own box passes BBox::intersects" (child boxes are nested
// exactly and the test is monotone in the box, see traverse_ww: the decision is the same one, f32 where the box is clearly hit or
// missed, the reference's f64 test on the stored box in between).  A root that is a leaf is never tested (octree.rs:149,177).  The
// octree walk it replaces fetched 55 nodes and 125 leaf entries and ran 20 exact tests per heptoroid feeler, most of them copies.
// A lane's list holds RTX_BVH_HITS refs; when one is full the wave runs phase 2 on what all lists hold and goes on.
// *overflow: the caller must run the all-f64 walk (an infinite inverse direction component, a full stack).
#ifndef RTX_BVH_HITS
#define RTX_BVH_HITS 8      // (1 in the forced-failure test build, tests/test_gpu_variants.py: phase 2 after every hit, a second hit in a round hands over)
#endif
template <int WW>
DEV void traverse_bvh(const DScene &sc, const Ray &r, bool active, uint32_t *s_stack, uint32_t *s_cand, double t_lo, double t_hi, Hit *hit,
                      bool *overflow) {
    static_assert(WW == WW_CLOSEST || WW == WW_ANY, "the shadow product walks with traverse_bvh_dealt");
    constexpr bool ANY = WW == WW_ANY;   // the window is [t_lo, t_hi] for every candidate, nothing is culled by a best hit
    // TIES: two different prims hit at a bit-identical t (a ray through the shared edge or vertex of two triangles).  ray.rs:52-60 keeps the
    // one it VISITS first (`t < nearest_t` is strict), and its visiting order is the octree's (children popped 7 -> 0, a leaf's entries in
    // insertion order, octree.rs:177-187), which this walk does not follow.  It only notices the tie: the ray is then answered by the
    // all-f64 walk in the reference's order (k_closest_retry / traverse_exact<MODE_CLOSEST_REF>), like a degenerate ray.  A tie that a
    // smaller t supersedes is forgotten.  Planes come last in the reference too (octree.rs:163-165): a finite prim wins a tie with one.
    bool tie = false;
    uint32_t *stk = s_stack + threadIdx.x;   // stride RTX_BLOCK
    uint32_t *cand = s_cand + threadIdx.x;   // parked candidates, stride RTX_BLOCK: (ref << 1) | surely_in
    double best_t = ANY ? t_hi : inf();      // closest: nearest_t (ray.rs:47); any-hit: distance_to_light, never lowered
    int best_ref = REF_NONE;
    WORK(uint32_t w_nodes = 0; uint32_t w_entries = 0; uint32_t w_prims = 0;)

    const RootEntry re = root_entry(sc, r, r.ox - sc.centre[0], r.oy - sc.centre[1], r.oz - sc.centre[2]);
    const double tsh = re.t_shift;   // every f32 slab distance below is t - tsh
    const double orx = r.ox - sc.centre[0] + tsh * r.dx, ory = r.oy - sc.centre[1] + tsh * r.dy, orz = r.oz - sc.centre[2] + tsh * r.dz;
    const double big = fmax(fmax(fabs(orx), fabs(ory)), fabs(orz)) + sc.radius;
    const double delta = big * (1.0 / 1048576.0);
    const float ix = (float)r.ix, iy = (float)r.iy, iz = (float)r.iz;
    const float hx = f32_above(delta * fabs(r.ix)), hy = f32_above(delta * fabs(r.iy)), hz = f32_above(delta * fabs(r.iz));
    const float cx = (float)(-(orx * r.ix)), cy = (float)(-(ory * r.iy)), cz = (float)(-(orz * r.iz));
    const float cnx = cx - hx, cfx = cx + hx, cny = cy - hy, cfy = cy + hy, cnz = cz - hz, cfz = cz + hz;
    const bool finite_dir = fabsf(ix) < 3.0e38f && fabsf(iy) < 3.0e38f && fabsf(iz) < 3.0e38f;
    float bf = f32_above(best_t - tsh);
    // rows of a DNode4 in 16-byte units: lo.x lo.y lo.z hi.x hi.y hi.z ref
    // (in BYTES: a node's rows are fetched at 32-bit byte offsets from the array's base, which sits in scalar registers: one 32-bit add per
    // row instead of a 64-bit one on a per-lane pointer; the host keeps hierarchies that pass 4 GB off these walks)
    const uint32_t onx = r.sx ? 0u : 48u, ofx = r.sx ? 48u : 0u, ony = r.sy ? 16u : 64u, ofy = r.sy ? 64u : 16u, onz = r.sz ? 32u : 80u, ofz = r.sz ? 80u : 32u;
    // stack entry = node index | quantised entry distance << node_bits; the distance is rounded DOWN (a pop never drops a node that
    // could still hold the closest hit); every entry distance lies in [0, 4 * big]
    const uint32_t nbits = sc.bvh4_node_bits, nmask = (1u << nbits) - 1u;
    const float qmax = (float)((1u << (32u - nbits)) - 1u);
    const float qscale = qmax / (4.0f * (float)big) * 0.999999f;
    const float qinv = 4.0f * (float)big / qmax * 0.999999f;

    uint32_t sp = 0u, cnt = 0u, cur = 0u;
    bool running = active && !re.miss;   // a ray that misses the scene's cube has no finite candidate
    bool ovf = false;
    bool have_cur = running;             // the root (octree.rs:149,177 never tests the root's box; neither does this walk)
    const float INF = __int_as_float(0x7F800000);

    for (;;) {
        if (!__any(running || cnt > 0u)) break;
        for (int step = 0; step < RTX_BVH_STEPS && running && cnt <= RTX_BVH_CAND - RTX_BVH4; ++step) {
            if (!have_cur) {
                // the next stack entry that does not start behind the best hit: popped in a loop of its own (a culled entry used to cost
                // the lane one whole node step of the wave; once a ray has its hit, most of what its stack still holds is culled):
                // k_closest<5> sponza 14.85 -> 14.54 ms, sibenik 1.01 -> 0.98 ms
                bool got = false;
                while (sp > 0u) {
                    --sp;
                    const uint32_t e = stk[sp * RTX_BLOCK];
                    if (ANY || !((float)(e >> nbits) * qinv > bf)) { cur = e & nmask; got = true; break; }
                }
                if (!got) { running = false; break; }
            }
            have_cur = false;
            WORK(++w_nodes;)
            const char *nq = reinterpret_cast<const char *>(sc.bvh4);
            const uint32_t nb = cur * (uint32_t)sizeof(DNode4);
            #define BVH4_ROW(off) (*reinterpret_cast<const float4 *>(nq + (nb + (off))))
            const float4 pnx = BVH4_ROW(onx), pny = BVH4_ROW(ony), pnz = BVH4_ROW(onz), pfx = BVH4_ROW(ofx), pfy = BVH4_ROW(ofy), pfz = BVH4_ROW(ofz);
            const uint4 rf = *reinterpret_cast<const uint4 *>(nq + (nb + 96u));
#undef BVH4_ROW
            const float nxs[4] = {pnx.x, pnx.y, pnx.z, pnx.w}, nys[4] = {pny.x, pny.y, pny.z, pny.w}, nzs[4] = {pnz.x, pnz.y, pnz.z, pnz.w};
            const float fxs[4] = {pfx.x, pfx.y, pfx.z, pfx.w}, fys[4] = {pfy.x, pfy.y, pfy.z, pfy.w}, fzs[4] = {pfz.x, pfz.y, pfz.z, pfz.w};
            const uint32_t refs[4] = {rf.x, rf.y, rf.z, rf.w};
            float key[4];   // entry distance of a node child that is hit, else +inf
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (an empty slot's box is (+inf, -inf): lo > hi or NaN, never hit)
                const float lo = fmaxf(fmaxf(fmaxf(fmaf(nxs[c], ix, cnx), 0.0f), fmaf(nys[c], iy, cny)), fmaf(nzs[c], iz, cnz));
                const float hi = fminf(fminf(fminf(fmaf(fxs[c], ix, cfx), bf), fmaf(fys[c], iy, cfy)), fmaf(fzs[c], iz, cfz));
                const bool hit_c = lo <= hi;
                const uint32_t kind = refs[c] & RTX_WREF_KIND;
                key[c] = (hit_c && kind == RTX_WREF_NODE) ? lo : INF;
                if (hit_c && (kind == RTX_WREF_TRI || kind == RTX_WREF_SPHERE)) {
                    WORK(++w_entries;)
                    // the box shrunk by delta instead of widened: a ray through it passes the exact f64 entry test (octree.rs:170)
                    // with a margin far above f64 rounding, so that test can be skipped
                    bool surely_in = false;
                    if (finite_dir) {
                        const float li = fmaxf(fmaxf(fmaf(nxs[c], ix, cfx), fmaf(nys[c], iy, cfy)), fmaf(nzs[c], iz, cfz));
                        const float hi2 = fminf(fminf(fmaf(fxs[c], ix, cnx), fmaf(fys[c], iy, cny)), fmaf(fzs[c], iz, cnz));
                        surely_in = li <= hi2 && hi2 > 0.0f;
                    }
                    const uint32_t idx = refs[c] & RTX_WREF_INDEX;
                    const int ref = kind == RTX_WREF_TRI ? (int)idx : ~(int)idx;
                    cand[cnt * RTX_BLOCK] = ((uint32_t)ref << 1) | (surely_in ? 1u : 0u);
                    ++cnt;
                }
            }
            // the nearest node child is visited next, the others are pushed (in slot order)
            int near = 0;
            float knear = key[0];
#pragma unroll
            for (int c = 1; c < 4; ++c)
                if (key[c] < knear) { knear = key[c]; near = c; }
            if (knear < INF) {
                have_cur = true;
                cur = (near == 0 ? refs[0] : (near == 1 ? refs[1] : (near == 2 ? refs[2] : refs[3]))) & RTX_WREF_INDEX;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c != near && key[c] < INF) {
                        if (sp < RTX_BVH_STACK) {
                            const uint32_t q = ANY ? 0u : (uint32_t)fminf(key[c] * qscale, qmax);
                            stk[sp * RTX_BLOCK] = (refs[c] & RTX_WREF_INDEX) | (q << nbits);
                            ++sp;
                        } else {
                            ovf = true;
                        }
                    }
                }
                if (ovf) { running = false; have_cur = false; }
            }
        }
        // ---- flush (as traverse_ww): the parked candidates through the exact tests, every lane one of its own per round
        const bool stalled = running && cnt > RTX_BVH_CAND - RTX_BVH4;
        const uint32_t holders = (uint32_t)__popcll(__ballot(cnt > 0u));
        if (__any(stalled) || holders >= RTX_BVH_FLUSH || !__any(running)) {
            for (uint32_t i = 0; __any(i < cnt); ++i) {
                if (i < cnt && !(ANY && best_ref != REF_NONE) && !ovf) {   // any-hit: one occluder is enough
                    const uint32_t c = cand[i * RTX_BLOCK];
                    const int ref = (int)c >> 1;
                    const bool surely_in = (c & 1u) != 0u;
                    if (ref >= 0) {
                        int pid, mat;
                        TriV tr = load_tri32(sc.tris, (uint32_t)ref, &pid, &mat);
                        if (surely_in || tri_bbox_hit(tr, r)) {
                            WORK(++w_prims;)
                            double t, be, ga;
                            if (ANY) {   // renderer.rs:212: t in [EPSILON, distance_to_light]
                                if (tri_hit(tr, r, t_lo, t_hi, &t, &be, &ga)) best_ref = ref;
                            } else if (tri_hit(tr, r, t_lo, best_t, &t, &be, &ga) && t > t_lo) {
                                // prim.intersects(ray, t_min, nearest_t) then `t > t_min && t < nearest_t` (ray.rs:50-55)
                                if (t < best_t) { best_t = t; best_ref = ref; tie = false; }
                                else if (t == best_t && ref != best_ref) tie = true;   // see `tie` above
                            }
                        }
                    } else {
                        const DSphere &sp_ = sc.spheres[~ref];
                        V3 cc = v3(sp_.c[0], sp_.c[1], sp_.c[2]);
                        double radius = sp_.r;
                        if (surely_in || sphere_bbox_hit(cc, radius, r)) {
                            WORK(++w_prims;)
                            double t;
                            if (ANY) {
                                if (sphere_hit(cc, radius, r, t_lo, t_hi, &t)) best_ref = ref;
                            } else if (sphere_hit(cc, radius, r, t_lo, best_t, &t) && t > t_lo) {
                                if (t < best_t) { best_t = t; best_ref = ref; tie = false; }
                                else if (t == best_t && ref != best_ref) tie = true;
                            }
                        }
                    }
                }
            }
            cnt = 0u;
            bf = f32_above(best_t - tsh);
            if ((ANY && best_ref != REF_NONE) || ovf) running = false;
        }
    }

    // infinite prims last, in order (octree.rs:163-165,185-186)
    if (active && !ovf) {
        for (int p = 0; p < sc.n_planes; ++p) {
            const DPlane &pl = sc.planes[p];
            double t;
            if (ANY) {
                if (best_ref == REF_NONE && plane_hit(pl, r, t_lo, t_hi, &t)) best_ref = REF_PLANE | p;
            } else if (plane_hit(pl, r, t_lo, best_t, &t) && t > t_lo && t < best_t) {
                best_t = t;
                best_ref = REF_PLANE | p;
            }
        }
    }
    hit->t = best_t;
    hit->ref = best_ref;
    *overflow = ovf || tie;   // (either way: the caller runs the all-f64 walk in the reference's order)
    WORK(hit->w_nodes = w_nodes; hit->w_entries = w_entries; hit->w_prims = w_prims;)
}

// traverse_bvh_dealt: THE EXACT TESTS ARE DEALT OUT OVER THE WAVE.  In traverse_bvh a lane tests its own parked candidates, and a round of a
// flush runs as long as the lane with the most candidates has some: measured 18 % of the lanes busy in a round on heptoroid feelers
// (14 exact tests per feeler), 23 % on its closest-hit rays (11 per ray), 28 % on sponza's (4 per ray).  Here the walk appends (owner
// lane, primitive) pairs to ONE list per wave (a ballot and a prefix count per child slot), and a round hands 64 pairs to the 64
// lanes, whoever owns the ray: the tester fetches the owner's ray with ds_bpermute (20 words), runs the reference's tests on it and
// reports through LDS: closest hit = a 64-bit atomic minimum on the bits of t (t > 0: the order of the bit patterns is the order of the
// values), then the lowest index among the prims that hold the minimum; any hit = the ref; product = a slot in the owner's hit list,
// taken with an atomic counter.  95 % of the lanes are busy in a round; the price is 20 more VGPRs (the foreign ray next to the own
// one: 4 waves per SIMD instead of 5) and a dearer step, so it runs where the exact tests dominate: the shadow product (heptoroid
// 393 -> 313 ms of shadows per frame) and the closest-hit rays of scenes with transmissive materials, whose rays run along the inside
// of thin shells (251 -> 235 ms of closest hit + shading); with 1 to 4 exact tests per ray (bunny, sibenik, sponza) it is 2 to 7 % slower.
// pair word: owner lane << 26 | surely_in << 25 | sphere << 24 | index (the host keeps scenes with 2^24 triangles or spheres off this walk)
#define RTX_BVH_LIST (RTX_BVH_CAND * 64)   // words in a wave's pair list
#ifndef RTX_BVH_DEALT_WAVES
#define RTX_BVH_DEALT_WAVES 5   // k_closest<MODE_CLOSEST_BVH_DEALT>: 96 VGPRs and 40 B of scratch per lane (at 4 waves: 114 VGPRs, no scratch; heptoroid closest + shade 247 against 235 ms)
#endif
#ifndef RTX_BVH_ROUND_MIN
#define RTX_BVH_ROUND_MIN 16u   // closest hit: a round runs once this many pairs wait (heptoroid closest + shade: 16: 233, 32: 235, 64: 238 ms)
#endif
template <int WW>
DEV void traverse_bvh_dealt(const DScene &sc, const Ray &r, bool active, uint32_t *s_stack, uint32_t *s_cand, uint32_t *s_aux, double t_lo, double t_hi,
                      Hit *hit, bool *overflow, uint32_t *s_hits = nullptr, V3 *prod = nullptr) {
    constexpr bool PRODUCT = WW == WW_PRODUCT;
    constexpr bool CLOSEST = WW == WW_CLOSEST;
    constexpr bool ANY = !CLOSEST;   // the window is [t_lo, t_hi] for every candidate, nothing is culled by a best hit
    const uint32_t lane = threadIdx.x & 63u, wbase = threadIdx.x & ~63u;
    uint32_t *stk = s_stack + threadIdx.x;                         // stride RTX_BLOCK
    uint32_t *list = s_cand + (threadIdx.x >> 6) * RTX_BVH_LIST;   // the wave's pair list
    // per-lane result words, written by whichever lane tests a pair of this lane's ray:
    // closest: u64 best[RTX_BLOCK] (bits of t), int ref[RTX_BLOCK], u32 tie[RTX_BLOCK]; any: int ref[RTX_BLOCK]; product: u32 n_hits[RTX_BLOCK]
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(s_aux);
    int *s_ref = reinterpret_cast<int *>(CLOSEST ? s_aux + 2 * RTX_BLOCK : s_aux);
    uint32_t *s_nh = s_aux;
    uint32_t *s_tie = CLOSEST ? s_aux + 3 * RTX_BLOCK : s_aux;   // closest: two different prims hold the minimum (see `tie` in traverse_bvh)
#define BVH_FENCE() asm volatile("" ::: "memory")   /* LDS traffic between lanes of ONE wave: executed in order, only the compiler must not move or cache it */
    double best_t = ANY ? t_hi : inf();      // closest: nearest_t (ray.rs:47); any-hit: distance_to_light, never lowered
    int best_ref = REF_NONE;
    WORK(uint32_t w_nodes = 0; uint32_t w_entries = 0; uint32_t w_prims = 0;)
    if constexpr (CLOSEST) { s_best[threadIdx.x] = (unsigned long long)__double_as_longlong(inf()); s_ref[threadIdx.x] = -1; s_tie[threadIdx.x] = 0u; }
    else if constexpr (PRODUCT) s_nh[threadIdx.x] = 0u;
    else s_ref[threadIdx.x] = REF_NONE;
    BVH_FENCE();

    const RootEntry re = root_entry(sc, r, r.ox - sc.centre[0], r.oy - sc.centre[1], r.oz - sc.centre[2]);
    const double tsh = re.t_shift;   // every f32 slab distance below is t - tsh
    const double orx = r.ox - sc.centre[0] + tsh * r.dx, ory = r.oy - sc.centre[1] + tsh * r.dy, orz = r.oz - sc.centre[2] + tsh * r.dz;
    const double big = fmax(fmax(fabs(orx), fabs(ory)), fabs(orz)) + sc.radius;
    const double delta = big * (1.0 / 1048576.0);
    const float ix = (float)r.ix, iy = (float)r.iy, iz = (float)r.iz;
    const float hx = f32_above(delta * fabs(r.ix)), hy = f32_above(delta * fabs(r.iy)), hz = f32_above(delta * fabs(r.iz));
    const float cx = (float)(-(orx * r.ix)), cy = (float)(-(ory * r.iy)), cz = (float)(-(orz * r.iz));
    const float cnx = cx - hx, cfx = cx + hx, cny = cy - hy, cfy = cy + hy, cnz = cz - hz, cfz = cz + hz;
    const bool finite_dir = fabsf(ix) < 3.0e38f && fabsf(iy) < 3.0e38f && fabsf(iz) < 3.0e38f;
    float bf = f32_above(best_t - tsh);
    // rows of a DNode4 in 16-byte units: lo.x lo.y lo.z hi.x hi.y hi.z ref
    // (in BYTES: a node's rows are fetched at 32-bit byte offsets from the array's base, which sits in scalar registers: one 32-bit add per
    // row instead of a 64-bit one on a per-lane pointer; the host keeps hierarchies that pass 4 GB off these walks)
    const uint32_t onx = r.sx ? 0u : 48u, ofx = r.sx ? 48u : 0u, ony = r.sy ? 16u : 64u, ofy = r.sy ? 64u : 16u, onz = r.sz ? 32u : 80u, ofz = r.sz ? 80u : 32u;
    // stack entry = node index | quantised entry distance << node_bits; the distance is rounded DOWN (a pop never drops a node that
    // could still hold the closest hit); every entry distance lies in [0, 4 * big]
    const uint32_t nbits = sc.bvh4_node_bits, nmask = (1u << nbits) - 1u;
    const float qmax = (float)((1u << (32u - nbits)) - 1u);
    const float qscale = qmax / (4.0f * (float)big) * 0.999999f;
    const float qinv = 4.0f * (float)big / qmax * 0.999999f;

    uint32_t sp = 0u, cur = 0u;
    uint32_t n_pairs = 0u;               // pairs waiting in the wave's list (wave-uniform)
#ifdef RTX_COUNT_FLUSH
    uint32_t f_iters = 0, f_rounds = 0, f_tests = 0;   // experiment: wave-steps, test rounds, pairs tested
#endif
    bool running = active && !re.miss;   // a ray that misses the scene's cube has no finite candidate
    bool ovf = PRODUCT && active && !finite_dir;   // 0 * inf = NaN slab distances: the all-f64 walk
    if (ovf) running = false;
    bool have_cur = running;             // the root (octree.rs:149,177 never tests the root's box; neither does this walk)
    uint32_t *hits = PRODUCT ? s_hits + threadIdx.x : nullptr;   // WW_PRODUCT: the distinct prims hit so far, stride RTX_BLOCK
    const float INF = __int_as_float(0x7F800000);

    // ---- WW_PRODUCT, phase 2: every distinct hit prim counts once per reached leaf that lists it.  Every lane works through its own
    // list: when some list is half full (a round may add several hits to one list) and after the walk.
    V3 acc = v3(1.0, 1.0, 1.0);   // renderer.rs:211
    bool zero = false;            // ... has reached exactly (0,0,0): finite factors, it stays there
    auto count_hits = [&]() {
        const bool root_is_leaf = sc.nodes32[0].aux != 0xFFFFFFFFu;
        BVH_FENCE();
        uint32_t n_hits = s_nh[threadIdx.x];
        if (n_hits > RTX_BVH_HITS) n_hits = RTX_BVH_HITS;   // (hits that found the list full wait in the pair list and are tested again)
        for (uint32_t h = 0; h < n_hits && !ovf && !zero; ++h) {
            const int ref = (int)hits[h * RTX_BLOCK];
            int pid, mat;
            if (ref >= 0) {
                const long long ids = __double_as_longlong(reinterpret_cast<const double *>(sc.tris + ref)[9]);   // (prim_id, material)
                pid = (int)(ids & 0xFFFFFFFFLL);
                mat = (int)(ids >> 32);
            } else {
                pid = sc.spheres[~ref].prim_id;
                mat = sc.spheres[~ref].material;
            }
            uint32_t m = 1u;
#ifdef RTX_ABL_NOCOUNT
            if (false) {       // ablation (wrong products): phase 1 alone
#else
            if (!root_is_leaf) {
#endif
                m = 0u;
                const int e0 = sc.prim_leaf_begin[pid], e1 = sc.prim_leaf_begin[pid + 1];
                // the f32 boxes of the prim's leaves lie NEXT TO EACH OTHER in prim_leaf_boxes (copies of their nodes32 records, in the order of
                // prim_leaves): a run of 32-byte records instead of an index and a scattered record per leaf, fetched two at a time
                auto leaf_reached = [&](int e, const float4 &qa, const float4 &qb) -> uint32_t {
                    const float pnx = r.sx ? qa.x : qa.w, pfx = r.sx ? qa.w : qa.x;
                    const float pny = r.sy ? qa.y : qb.x, pfy = r.sy ? qb.x : qa.y;
                    const float pnz = r.sz ? qa.z : qb.y, pfz = r.sz ? qb.y : qa.z;
                    // the box widened by delta missed => the f64 test fails (as the octree walk descends)
                    const float lo = fmaxf(fmaxf(fmaxf(fmaf(pnx, ix, cnx), 0.0f), fmaf(pny, iy, cny)), fmaf(pnz, iz, cnz));
                    const float hi = fminf(fminf(fmaf(pfx, ix, cfx), fmaf(pfy, iy, cfy)), fmaf(pfz, iz, cfz));
                    if (!(lo <= hi)) return 0u;
                    // the box shrunk by delta passed => the f64 test passes; in between: the reference's test on the stored box
                    bool reached = surely_hit(fmaf(pnx, ix, cfx), fmaf(pny, iy, cfy), fmaf(pnz, iz, cfz), fmaf(pfx, ix, cnx), fmaf(pfy, iy, cny),
                                              fmaf(pfz, iz, cnz));
                    if (!reached) {
                        const double2 *dq = reinterpret_cast<const double2 *>(sc.nodes + (uint32_t)sc.prim_leaves[e]);
                        const double2 q0 = dq[0], q1 = dq[1], q2 = dq[2];
                        double tn, tf;
                        reached = slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, r, &tn, &tf);
                    }
                    return reached ? 1u : 0u;
                };
                const float4 *lb = reinterpret_cast<const float4 *>(sc.prim_leaf_boxes);
                int e = e0;
                for (; e + 1 < e1; e += 2) {
                    const float4 a0 = lb[2 * e], a1 = lb[2 * e + 1], b0 = lb[2 * e + 2], b1 = lb[2 * e + 3];
                    m += leaf_reached(e, a0, a1);
                    m += leaf_reached(e + 1, b0, b1);
                }
                if (e < e1) {
                    const float4 a0 = lb[2 * e], a1 = lb[2 * e + 1];
                    m += leaf_reached(e, a0, a1);
                }
            }
            const rtx_material &mt = sc.materials[mat];
            const V3 tr_ = mt.kind == RTX_MAT_FLAT ? v3(0.0, 0.0, 0.0) : v3(mt.transmission[0], mt.transmission[1], mt.transmission[2]);
            for (uint32_t k = 0; k < m; ++k) acc = acc * tr_;
            zero = acc.x == 0.0 && acc.y == 0.0 && acc.z == 0.0;   // finite factors: it stays exactly 0
        }
        if (zero) running = false;
        s_nh[threadIdx.x] = 0u;
        BVH_FENCE();
    };

    // ---- one round of exact tests: up to 64 waiting pairs, one per lane.  Must be called by all 64 lanes.
    auto test_round = [&]() {
        if constexpr (PRODUCT) {
            BVH_FENCE();
            if (__any(s_nh[threadIdx.x] > RTX_BVH_HITS / 2u)) count_hits();
        }
        const uint32_t take = n_pairs < 64u ? n_pairs : 64u;
        n_pairs -= take;
        const bool mine = lane < take;
#ifdef RTX_COUNT_FLUSH
        ++f_rounds;
        if (mine) ++f_tests;
#endif
        BVH_FENCE();
        const uint32_t pw = list[n_pairs + (mine ? lane : 0u)];
        const int ol = (int)(pw >> 26);          // the owner's lane in the wave
        const uint32_t ot = wbase + (uint32_t)ol;   // ... and in the workgroup
        Ray q;                                   // the owner's ray (ray.rs:23-42: the same bits, fetched instead of rebuilt)
        q.ox = __shfl(r.ox, ol); q.oy = __shfl(r.oy, ol); q.oz = __shfl(r.oz, ol);
        q.dx = __shfl(r.dx, ol); q.dy = __shfl(r.dy, ol); q.dz = __shfl(r.dz, ol);
        q.ix = __shfl(r.ix, ol); q.iy = __shfl(r.iy, ol); q.iz = __shfl(r.iz, ol);   // (rebuilt with three divisions where needed: no faster)
        q.sx = q.ix > 0.0; q.sy = q.iy > 0.0; q.sz = q.iz > 0.0;
        q.flip = 0u;                             // (the octree walks' child order: not read here)
        const double q_hi = ANY ? __shfl(t_hi, ol) : inf();
        bool run = mine;
        if constexpr (ANY && !PRODUCT) run = run && s_ref[ot] == REF_NONE;   // any-hit: one occluder is enough
        bool found = false;
        double t = 0.0;
        const bool surely_in = (pw & (1u << 25)) != 0u;
        const uint32_t idx = pw & 0x00FFFFFFu;
        const int ref = (pw & (1u << 24)) ? ~(int)idx : (int)idx;
        if (run) {
            if (ref >= 0) {
                int pid, mat;
                TriV tr = load_tri32(sc.tris, (uint32_t)ref, &pid, &mat);
                if (surely_in || tri_bbox_hit(tr, q)) {
                    WORK(++w_prims;)
                    double be, ga;
                    // closest hit: prim.intersects(ray, t_min, nearest_t) then `t > t_min && t < nearest_t` (ray.rs:50-55) is a running
                    // minimum over the candidates with t > t_min: the window's upper end only rejects what the minimum rejects
                    found = tri_hit(tr, q, t_lo, q_hi, &t, &be, &ga);     // any hit / product: renderer.rs:212, t in [EPSILON, distance_to_light]
                }
            } else {
                const DSphere &sp_ = sc.spheres[~ref];
                V3 cc = v3(sp_.c[0], sp_.c[1], sp_.c[2]);
                double radius = sp_.r;
                if (surely_in || sphere_bbox_hit(cc, radius, q)) {
                    WORK(++w_prims;)
                    found = sphere_hit(cc, radius, q, t_lo, q_hi, &t);
                }
            }
        }
        if constexpr (CLOSEST) {
            // s_ref holds the winner as (sphere << 24 | index), 0xFFFFFFFF = none.  Prims that tie at exactly the same t (a ray through
            // the shared edge of two triangles): the lowest index is kept, whichever round or lane tests it, and s_tie is set: ray.rs:52-60
            // keeps the first it visits in the octree's order, so the caller hands a ray whose final minimum is tied to the all-f64
            // walk in that order (as traverse_bvh does)
            found = found && t > t_lo;
            const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
            uint32_t *s_key = reinterpret_cast<uint32_t *>(s_ref);
            const unsigned long long before = s_best[ot];
            if (found) atomicMin(&s_best[ot], tb);
            BVH_FENCE();
            const bool wins = found && s_best[ot] == tb;
            if (wins && tb < before) { s_key[ot] = 0xFFFFFFFFu; s_tie[ot] = 0u; }   // a new minimum: the old winner goes (every lane that ties writes the same)
            BVH_FENCE();
            if (wins) {
                const uint32_t mine_k = pw & 0x01FFFFFFu;
                const uint32_t old = atomicMin(&s_key[ot], mine_k);
                if (old != 0xFFFFFFFFu && old != mine_k) s_tie[ot] = 1u;   // (set after every reset of this round: the LDS runs a wave's operations in order)
            }
        } else if constexpr (PRODUCT) {
            // a hit that finds its owner's list full (more than RTX_BVH_HITS / 2 hits of one feeler in one round) goes back into the
            // pair list: the next round starts with phase 2 on the full list, then the pair is tested again and fits
            bool again = false;
            if (found) {
                const uint32_t slot = atomicAdd(&s_nh[ot], 1u);
                if (slot < RTX_BVH_HITS) s_hits[slot * RTX_BLOCK + ot] = (uint32_t)ref;
                else again = true;
            }
            const unsigned long long am = __ballot(again);
            if (am != 0ull) {
                if (again) list[n_pairs + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u))] = pw;
                n_pairs += (uint32_t)__popcll(am);
            }
        } else {
            if (found) s_ref[ot] = ref;
        }
        BVH_FENCE();
    };

    for (;;) {
        for (int step = 0; step < RTX_BVH_STEPS; ++step) {
            if (n_pairs > RTX_BVH_LIST - 4u * 64u || !__any(running)) break;   // (wave-uniform) room for one more step of every lane
#ifdef RTX_COUNT_FLUSH
            if (lane == 0u) ++f_iters;
#endif
            bool visit = running;
            if (visit && !have_cur) {
                if (sp == 0u) { running = false; visit = false; }
                else {
                    --sp;
                    const uint32_t e = stk[sp * RTX_BLOCK];
                    if (CLOSEST && (float)(e >> nbits) * qinv > bf) visit = false;   // it starts behind the best hit (popping on in a loop of its own, as traverse_bvh does: heptoroid closest hit 147.3 -> 155.5 ms)
                    else cur = e & nmask;
                }
            }
            bool push[4] = {false, false, false, false};
            uint32_t word[4] = {0u, 0u, 0u, 0u};
            if (visit) {
                have_cur = false;
                WORK(++w_nodes;)
                const char *nq = reinterpret_cast<const char *>(sc.bvh4);
                const uint32_t nb = cur * (uint32_t)sizeof(DNode4);
                #define BVH4_ROW(off) (*reinterpret_cast<const float4 *>(nq + (nb + (off))))
                const float4 pnx = BVH4_ROW(onx), pny = BVH4_ROW(ony), pnz = BVH4_ROW(onz), pfx = BVH4_ROW(ofx), pfy = BVH4_ROW(ofy), pfz = BVH4_ROW(ofz);
                const uint4 rf = *reinterpret_cast<const uint4 *>(nq + (nb + 96u));
#undef BVH4_ROW
                const float nxs[4] = {pnx.x, pnx.y, pnx.z, pnx.w}, nys[4] = {pny.x, pny.y, pny.z, pny.w}, nzs[4] = {pnz.x, pnz.y, pnz.z, pnz.w};
                const float fxs[4] = {pfx.x, pfx.y, pfx.z, pfx.w}, fys[4] = {pfy.x, pfy.y, pfy.z, pfy.w}, fzs[4] = {pfz.x, pfz.y, pfz.z, pfz.w};
                const uint32_t refs[4] = {rf.x, rf.y, rf.z, rf.w};
                float key[4];   // entry distance of a node child that is hit, else +inf
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // (an empty slot's box is (+inf, -inf): lo > hi or NaN, never hit)
                    const float lo = fmaxf(fmaxf(fmaxf(fmaf(nxs[c], ix, cnx), 0.0f), fmaf(nys[c], iy, cny)), fmaf(nzs[c], iz, cnz));
                    const float hi = fminf(fminf(fminf(fmaf(fxs[c], ix, cfx), bf), fmaf(fys[c], iy, cfy)), fmaf(fzs[c], iz, cfz));
                    const bool hit_c = lo <= hi;
                    const uint32_t kind = refs[c] & RTX_WREF_KIND;
                    key[c] = (hit_c && kind == RTX_WREF_NODE) ? lo : INF;
                    if (hit_c && (kind == RTX_WREF_TRI || kind == RTX_WREF_SPHERE)) {
                        WORK(++w_entries;)
                        // the box shrunk by delta instead of widened: a ray through it passes the exact f64 entry test (octree.rs:170)
                        // with a margin far above f64 rounding, so that test can be skipped
                        bool surely_in = false;
                        if (finite_dir) {
                            const float li = fmaxf(fmaxf(fmaf(nxs[c], ix, cfx), fmaf(nys[c], iy, cfy)), fmaf(nzs[c], iz, cfz));
                            const float hi2 = fminf(fminf(fmaf(fxs[c], ix, cnx), fmaf(fys[c], iy, cny)), fmaf(fzs[c], iz, cnz));
                            surely_in = li <= hi2 && hi2 > 0.0f;
                        }
                        push[c] = true;
                        word[c] = (lane << 26) | (surely_in ? (1u << 25) : 0u) | (kind == RTX_WREF_SPHERE ? (1u << 24) : 0u) | (refs[c] & 0x00FFFFFFu);
                    }
                }
                // the nearest node child is visited next, the others are pushed (in slot order)
                int near = 0;
                float knear = key[0];
#pragma unroll
                for (int c = 1; c < 4; ++c)
                    if (key[c] < knear) { knear = key[c]; near = c; }
                if (knear < INF) {
                    have_cur = true;
                    cur = (near == 0 ? refs[0] : (near == 1 ? refs[1] : (near == 2 ? refs[2] : refs[3]))) & RTX_WREF_INDEX;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c != near && key[c] < INF) {
                            if (sp < RTX_BVH_STACK) {
                                const uint32_t q = ANY ? 0u : (uint32_t)fminf(key[c] * qscale, qmax);
                                stk[sp * RTX_BLOCK] = (refs[c] & RTX_WREF_INDEX) | (q << nbits);
                                ++sp;
                            } else {
                                ovf = true;
                            }
                        }
                    }
                    if (ovf) { running = false; have_cur = false; }
                }
            }
            // the candidates of this step join the wave's list
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned long long pm = __ballot(push[c]);
                if (push[c]) list[n_pairs + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u))] = word[c];
                n_pairs += (uint32_t)__popcll(pm);
            }
        }
        // ---- the exact tests: full rounds as soon as they can be had (closest hit: sooner, a best t found early culls nodes), the rest
        // once no lane walks any more
        const bool over = !__any(running);
        while (n_pairs >= (CLOSEST ? RTX_BVH_ROUND_MIN : 64u) || (over && n_pairs > 0u)) test_round();
        BVH_FENCE();
        if constexpr (CLOSEST) {
            best_t = __longlong_as_double((long long)s_best[threadIdx.x]);
            bf = f32_above(best_t - tsh);
        } else if constexpr (!PRODUCT) {
            if (s_ref[threadIdx.x] != REF_NONE) running = false;   // any-hit: one occluder is enough
        }
        if (over) break;
    }

    if constexpr (PRODUCT) count_hits();   // what the lists still hold
    if constexpr (CLOSEST) {
        BVH_FENCE();
        const uint32_t kw = (uint32_t)s_ref[threadIdx.x];
        best_ref = kw == 0xFFFFFFFFu ? REF_NONE : ((kw & (1u << 24)) ? ~(int)(kw & 0x00FFFFFFu) : (int)kw);
    } else if constexpr (!PRODUCT) {
        BVH_FENCE();
        best_ref = s_ref[threadIdx.x];
    }
    // infinite prims last, in order (octree.rs:163-165,185-186)
    if (active && !ovf) {
        for (int p = 0; p < sc.n_planes; ++p) {
            const DPlane &pl = sc.planes[p];
            double t;
            if (PRODUCT) {
                if (plane_hit(pl, r, t_lo, t_hi, &t)) {
                    const rtx_material &m = sc.materials[pl.material];
                    if (m.kind == RTX_MAT_FLAT) acc = acc * v3(0.0, 0.0, 0.0);
                    else acc = acc * v3(m.transmission[0], m.transmission[1], m.transmission[2]);
                }
            } else if (ANY) {
                if (best_ref == REF_NONE && plane_hit(pl, r, t_lo, t_hi, &t)) best_ref = REF_PLANE | p;
            } else if (plane_hit(pl, r, t_lo, best_t, &t) && t > t_lo && t < best_t) {
                best_t = t;
                best_ref = REF_PLANE | p;
            }
        }
    }
    hit->t = best_t;
    hit->ref = best_ref;
    *overflow = ovf;
    if constexpr (CLOSEST) { if (s_tie[threadIdx.x] != 0u) *overflow = true; }
    if (PRODUCT) *prod = acc;
    WORK(hit->w_nodes = w_nodes; hit->w_entries = w_entries; hit->w_prims = w_prims;)
#ifdef RTX_COUNT_FLUSH   // (with RTX_COUNT_WORK) experiment: wave-steps and test rounds instead of lane-steps and entries
    hit->w_nodes = f_iters; hit->w_entries = lane == 0u ? f_rounds : 0u; hit->w_prims = f_tests;
#endif
#undef BVH_FENCE
}

template <int MODE>
DEV void traverse(const DScene &sc, const Ray &r, uint32_t *s_lds, double t_lo, double t_hi, Hit *hit, V3 *prod) {
    // MODE_CLOSEST has its own entry point (traverse_ww: wave-wide flushes, every lane must call it)
    static_assert(MODE != MODE_CLOSEST, "use traverse_ww");
    static_assert(MODE != MODE_ANY && MODE != MODE_PRODUCT_FAST, "use traverse_ww<WW_ANY / WW_PRODUCT>");
    traverse_exact<MODE>(sc, r, s_lds + threadIdx.x, t_lo, t_hi, hit, prod);
}

// ------------------------------------------------------------------------------ textures
struct C4 {
    double r, g, b, a;
};
// p / 255.0 (colorrgba.rs:116-124) for p = 0 .. 255 without the division sequence: the product with RN(1 / 255) and one residual
// correction, bit-identical to the IEEE quotient for every one of the 256 arguments (checked exhaustively: tests/test_oracle_known_answers.py)
DEV double u8_over_255(uint8_t p) {
    const double c = 0x1.010101010101p-8;   // RN(1 / 255)
    const double pd = (double)p;
    const double q = pd * c;
    return fma(fma(-255.0, q, pd), c, q);
}
DEV C4 px_f64(const uint8_t *atlas, const DImage &im, long long x, long long y) {
    // the reference panics out of range (surface.rs:102-109; only 1-pixel-wide/high images): clamp
    if (x >= im.w) x = im.w - 1;
    if (y >= im.h) y = im.h - 1;
    uchar4 p = *reinterpret_cast<const uchar4 *>(atlas + im.offset + (y * (long long)im.w + x) * 4);
    return C4{u8_over_255(p.x), u8_over_255(p.y), u8_over_255(p.z), u8_over_255(p.w)};  // colorrgba.rs:116-124
}
DEV C4 cmul(C4 c, double s) { return C4{c.r * s, c.g * s, c.b * s, c.a}; }               // colorrgba.rs:178-189
DEV C4 cadd(C4 a, C4 b) { return C4{a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }       // colorrgba.rs:138-148
DEV long long as_usize(double v) {                                                       // saturating `as usize`
    if (!(v > 0.0)) return 0;
    if (v >= 9.0e18) return 0x7FFFFFFFFFFFFFFFLL;
    return (long long)v;
}
DEV V3 image_color(const DScene &sc, int image, double u, double v) {                    // imagetexture.rs:26-47
    const DImage im = sc.images[image];
    double s = fmod(u, 1.0) * ((double)im.w - 1.0);
    double t = fmod(v, 1.0) * ((double)im.h - 1.0);
    long long x = as_usize(floor(s));
    long long y = as_usize(floor(t));
    double u_ratio = s - (double)x;
    double v_ratio = t - (double)y;
    double u_opposite = 1.0 - u_ratio;
    double v_opposite = 1.0 - v_ratio;
    C4 c = cadd(cmul(cadd(cmul(px_f64(sc.atlas, im, x, y), u_opposite), cmul(px_f64(sc.atlas, im, x + 1, y), u_ratio)), v_opposite),
                cmul(cadd(cmul(px_f64(sc.atlas, im, x, y + 1), u_opposite), cmul(px_f64(sc.atlas, im, x + 1, y + 1), u_ratio)), v_ratio));
    return v3(c.r / 1.0, c.g / 1.0, c.b / 1.0);                                          // to_vec3, colorrgba.rs:128-135
}
DEV V3 texture_color(const DScene &sc, int tex, double u, double v) {
    const rtx_texture &tx = sc.textures[tex];
    if (tx.kind == RTX_TEX_CHECKER) {                                                    // checkertexture.rs:13-23
        double s = fabs(fmod(u, tx.scale));
        double t = fabs(fmod(v, tx.scale));
        double half = tx.scale / 2.0;
        if ((s > half && t < half) || (s < half && t > half))
            return v3(tx.color1[0] / 1.0, tx.color1[1] / 1.0, tx.color1[2] / 1.0);
        return v3(tx.color2[0] / 1.0, tx.color2[1] / 1.0, tx.color2[2] / 1.0);
    }
    if (tx.kind == RTX_TEX_UV) return v3((fmod(u, 1.0) + 0.0) / 1.0, (fmod(v, 1.0) + 0.0) / 1.0, 0.0 / 1.0); // uvtexture.rs:10-14
    return image_color(sc, tx.image, u, v);
}
DEV V3 cubemap_color(const DScene &sc, V3 dir) {                                         // cubemap.rs:45-84
    double x_mag = fabs(dir.x), y_mag = fabs(dir.y), z_mag = fabs(dir.z);
    int face = -1;
    double s = 0.0, t = 0.0;
    if (x_mag >= y_mag && x_mag >= z_mag) {
        face = dir.x <= 0.0 ? 0 : 1;
        double sc_ = dir.x < 0.0 ? 1.0 : -1.0;
        s = sc_ * dir.z / fabs(dir.x);
        t = dir.y / fabs(dir.x);
    } else if (y_mag >= x_mag && y_mag >= z_mag) {
        face = dir.y <= 0.0 ? 2 : 3;
        double sc_ = dir.y < 0.0 ? 1.0 : -1.0;
        s = sc_ * dir.x / fabs(dir.y);
        t = dir.z / fabs(dir.y);
    } else if (z_mag >= y_mag && z_mag >= x_mag) {
        face = dir.z <= 0.0 ? 4 : 5;
        double sc_ = dir.z < 0.0 ? -1.0 : 1.0;
        s = sc_ * dir.x / fabs(dir.z);
        t = dir.y / fabs(dir.z);
    }
    double seam_delta = 0.0001;
    s = fmin(fmax(1.0 - (s * 0.5 + 0.5), seam_delta), 1.0 - seam_delta);
    t = fmin(fmax(1.0 - (t * 0.5 + 0.5), seam_delta), 1.0 - seam_delta);
    if (face < 0) return v3(0.0, 0.0, 0.0); // NaN direction: the reference panics (cubemap.rs:79-81)
    return image_color(sc, sc.skybox[face], s, t);
}

// ------------------------------------------------------------------------------ materials
// What Material::sample needs of a hit that does not depend on the light: computed once per hit, not once per light (the texture
// lookup alone is a chain of three dependent fetches and a bilinear blend; same operations, same order, same bits).
struct MatHit {
    int kind;
    V3 texc;                 // diffuse_texture colour at (u, v), or white
    V3 ambient;              // ambient.scale(k_a)
    V3 diffuse_kd;           // diffuse.scale(k_d)
    V3 specular_c;
    double k_s, shine, gauss_constant;
    double f0;               // cooktorrancematerial.rs:43
    double sqrt_shine;       // roughness.sqrt(), cooktorrancematerial.rs:48
};
DEV MatHit material_prepare(const DScene &sc, const rtx_material &m, double u, double v) {
    MatHit h;
    h.kind = m.kind;
    h.diffuse_kd = v3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    h.texc = v3(1.0, 1.0, 1.0);
    h.ambient = h.specular_c = v3(0.0, 0.0, 0.0);
    h.k_s = h.shine = h.gauss_constant = h.f0 = h.sqrt_shine = 0.0;
    if (m.kind == RTX_MAT_FLAT) return h;                                                // flatmaterial.rs:11-13
    h.specular_c = v3(m.specular[0], m.specular[1], m.specular[2]);
    h.texc = m.texture >= 0 ? texture_color(sc, m.texture, u, v) : v3(1.0 / 1.0, 1.0 / 1.0, 1.0 / 1.0);
    h.ambient = scale(v3(m.ambient[0], m.ambient[1], m.ambient[2]), m.k_a);
    h.diffuse_kd = scale(h.diffuse_kd, m.k_d);
    h.k_s = m.k_s; h.shine = m.shine; h.gauss_constant = m.gauss_constant;
    if (m.kind == RTX_MAT_COOK_TORRANCE) {
        double n1 = 1.0, n2 = m.ior;
        double f0r = (n1 - n2) / (n1 + n2);
        h.f0 = f0r * f0r;
        h.sqrt_shine = sqrt(m.shine);
    }
    return h;
}
// PHONG = false: the instance for scenes without a Phong material (k_shade<false>), without the generic pow() of phongmaterial.rs:33.
template <bool PHONG = true>
DEV V3 material_sample(const MatHit &m, V3 n, V3 i, V3 l) {
    if (m.kind == RTX_MAT_FLAT) return m.diffuse_kd;                                     // flatmaterial.rs:11-13
    if (m.kind == RTX_MAT_COOK_TORRANCE) {                                               // cooktorrancematerial.rs:26-58
        V3 diffuse = scale(m.diffuse_kd, dot(n, l)) * m.texc;
        V3 h = unit(l + i);
        double n_dot_h = dot(n, h);
        double n_dot_l = dot(n, l);
        double v_dot_h = dot(i, h);
        double n_dot_v = dot(n, i);
        // powf(5.0) by multiplication: four roundings, within 2 ulp of the exact power (libm's pow, which the reference calls, promises
        // < 1 ulp and is platform-dependent in its last bit anyway; the parity bar for shading is 1e-12 relative)
        const double c1 = 1.0 - v_dot_h, c2 = c1 * c1;
        double f = (c2 * c2 * c1) * (1.0 - m.f0) + m.f0;
        double alpha = acos(n_dot_h);
        double d = m.gauss_constant * exp(-alpha / m.sqrt_shine);
        double g1 = (2.0 * n_dot_h * n_dot_v) / v_dot_h;
        double g2 = (2.0 * n_dot_h * n_dot_l) / v_dot_h;
        double g = fmin(g1, g2);
        double brdf = f * d * g / (n_dot_v * n_dot_l * PI);
        return scale(m.specular_c, m.k_s * brdf) + diffuse + m.ambient;
    }
    if constexpr (!PHONG) return m.diffuse_kd;   // (not reached: the host launches this instance for scenes without Phong materials only)
    V3 h = unit(l + i);                                                                  // phongmaterial.rs:24-36
    V3 diffuse = scale(m.diffuse_kd, dot(n, l)) * m.texc;
    V3 specular = scale(scale(m.specular_c, m.k_s), pow(dot(n, h), m.shine));
    return m.ambient + diffuse + specular;
}
DEV double fresnel_reflect(double ior, V3 i, V3 n, bool inside) {                       // renderer.rs:226-247
    double n1 = inside ? ior : 1.0, n2 = inside ? 1.0 : ior;
    V3 actual_n = inside ? -n : n;
    double r0_sqrt = (n1 - n2) / (n1 + n2);
    double r0 = r0_sqrt * r0_sqrt;
    double cos_angle;
    if (n1 <= n2) {
        cos_angle = dot(i, actual_n);
    } else {
        V3 t;
        if (!refract(i, -actual_n, ior, inside, &t)) return 1.0;
        cos_angle = -dot(actual_n, t);
    }
    double cos_term = 1.0 - cos_angle;
    return fmin(fmax(r0 + ((1.0 - r0) * cos_term * cos_term * cos_term * cos_term * cos_term), 0.0), 1.0);
}

// ------------------------------------------------------------------------------ queues
DEV uint32_t lane_id() { return __lane_id(); }
// Wave-aggregated append: one atomic per wave per queue.  Must be reached by every lane that
// is active at the call site with the same `counter`.
DEV uint32_t wave_append(bool want, uint32_t *counter) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0u;
    uint32_t n = (uint32_t)__popcll(mask);
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, n);
    base = (uint32_t)__shfl((int)base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
}
DEV void count_rays(bool valid, unsigned long long *counter) {
    unsigned long long mask = __ballot(valid);
    if (mask != 0ull && (int)lane_id() == __ffsll((long long)mask) - 1) atomicAdd(counter, (unsigned long long)__popcll(mask));
}
#ifdef RTX_COUNT_WORK
DEV void count_work(bool valid, const Hit &h, unsigned long long *w) {
    if (!valid) return;
    atomicAdd(w + 0, (unsigned long long)h.w_nodes);
    atomicAdd(w + 1, (unsigned long long)h.w_entries);
    atomicAdd(w + 2, (unsigned long long)h.w_prims);
}
#endif
DEV void acc_add(double *acc, uint32_t sample, V3 c) {
#ifdef RTX_EXP_NOACC
    if (c.x > -1.0) return;   // timing experiment only (wrong images)
#endif
#ifdef RTX_EXP_NOATOMIC
    if (c.x != 1.2345e300) return;   // timing experiment only (wrong images): the operands are fetched and multiplied, nothing is added
#endif
    // (a zero leaves the sum as it is: no atomic for it.  Half of a frame's shadow groups are fully occluded: tools/exp/occl_count.py)
    if (c.x == 0.0 && c.y == 0.0 && c.z == 0.0) return;
    double *p = acc + (size_t)sample * 3;
    unsafeAtomicAdd(p + 0, c.x);   // global_atomic_add_f64 on gfx950
    unsafeAtomicAdd(p + 1, c.y);
    unsafeAtomicAdd(p + 2, c.z);
}

// The same for a whole wave (EVERY lane of the wave must call it, `has` = this lane has something to add): lanes next to each other
// that add to the same sample are summed first (a segmented suffix sum over the runs of equal samples, 6 shuffle steps, skipped when
// no run is longer than one lane) and the run's first lane adds the total.  Children sit in the pool in RAY order, so deep in a
// refractive object all descendants of one primary sample are neighbours: 64 lanes x 3 atomics on ONE address, which the memory side
// executes one after the other (heptoroid: 229 of the shadow product's 472 ms per frame, 22 of k_shade's 63).
// The order of the additions differs from lane-by-lane atomics in the last bit, as the order of atomics itself does.
DEV void acc_add_runs(double *acc, bool has, uint32_t sample, V3 c) {
    const uint32_t lane = lane_id();
    const uint32_t key = has ? sample : 0xFFFFFFFFu;   // (no sample has this index: a call renders fewer than 2^32 samples)
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool head = lane == 0u || prev != key;
    if (__ballot(has && !head) != 0ull) {
        const unsigned long long heads = __ballot(head);
        const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull));   // runs numbered from lane 0 on
        if (!has) c = v3(0.0, 0.0, 0.0);
#pragma unroll
        for (uint32_t off = 1u; off < 64u; off <<= 1) {
            const uint32_t run2 = (uint32_t)__shfl_down((int)run, off);
            const double x = __shfl_down(c.x, off), y = __shfl_down(c.y, off), z = __shfl_down(c.z, off);
            if (lane + off < 64u && run2 == run) { c.x += x; c.y += y; c.z += z; }
        }
    }
    if (has && head) acc_add(acc, sample, c);
}

// What a shadow launch found for a queue group (shadow_intensity's return value, renderer.rs:220).  DEFERRED (sq.res != null): a plain
// store into the queue; k_accumulate, launched after the step's last light, adds  sum over the lights of  weight * shadow  to the sample:
// one set of atomics per ray instead of one per ray and light, issued by a kernel that does nothing else.  (The f64 atomics on the
// 800 MB accumulator of a 1080p x 16 frame are executed at the memory side; issued from the shadow kernels they cost the sponza frame
// 11 of k_shadow_coop's 164 ms.)  Else (launches whose tries are not a power of two, RTX_DIRECT_ACC=1): added at once.
// Every lane of the wave must call it.
DEV void shadow_result(const DShadowQ &sq, double *acc, uint32_t light, bool has, uint32_t group, uint32_t sample, V3 w, V3 shadow) {
    if (sq.res) {
        if (has) {
            double *r = sq.res + (size_t)light * 3 * sq.cap + group;
            r[0] = shadow.x; r[sq.cap] = shadow.y; r[2 * (size_t)sq.cap] = shadow.z;
        }
    } else {
        acc_add_runs(acc, has, sample, w * shadow);
    }
}

// sample index -> pixel: samples are laid out so that one wave64 = one 8x8 pixel tile of one
// sub-sample (coherent primary rays): s = ((tile * spp + sub) * 64 + lane)
DEV bool decode_sample(const DFrame &fr, uint32_t s, uint32_t *px, uint32_t *row, uint32_t *sub) {
    uint32_t per_tile = 64u * fr.spp;
    uint32_t tile = s / per_tile, rem = s % per_tile;
    *sub = rem >> 6;
    uint32_t lane = rem & 63u;
    uint32_t tx = tile % fr.tiles_x, j = tile / fr.tiles_x;
    uint32_t block_row = fr.shard_index + j * fr.shard_count;
    *px = tx * 8u + (lane & 7u);
    *row = block_row * 8u + (lane >> 3);
    return *px < fr.W && *row < fr.H;
}

// the row of an output buffer that holds image row `row` (row 0 = top): the row itself, or with RTX_FLAG_COMPACT_OUTPUT its place among
// the shard's own 8-row blocks
DEV uint32_t out_row(const DFrame &fr, uint32_t row) {
    if (!(fr.flags & RTX_FLAG_COMPACT_OUTPUT)) return row;
    return ((row >> 3) - fr.shard_index) / fr.shard_count * 8u + (row & 7u);
}

DEV void pool_store(const DPool &pool, uint32_t slot, V3 o, V3 d, V3 thr, uint64_t key, uint32_t sample, uint32_t flags) {
    size_t cap = pool.cap;
    double2 *r = reinterpret_cast<double2 *>(pool.rec + slot);   // (DRayRec: four 16-byte stores)
    r[0] = make_double2(o.x, o.y);
    r[1] = make_double2(o.z, d.x);
    r[2] = make_double2(d.y, d.z);
    r[3] = make_double2(__longlong_as_double((long long)key), __longlong_as_double((long long)(((uint64_t)flags << 32) | sample)));
    double *t = pool.thr + slot;
    t[0 * cap] = thr.x; t[1 * cap] = thr.y; t[2 * cap] = thr.z;
    pool.flags[slot] = flags;
}
struct PoolRay { V3 o, d; uint64_t key; uint32_t sample, flags; };
DEV PoolRay pool_load(const DPool &pool, uint32_t slot) {
    const double2 *r = reinterpret_cast<const double2 *>(pool.rec + slot);
    const double2 a = r[0], b = r[1], c = r[2], e = r[3];
    PoolRay p;
    p.o = v3(a.x, a.y, b.x);
    p.d = v3(b.y, c.x, c.y);
    p.key = (uint64_t)__double_as_longlong(e.x);
    const uint64_t sf = (uint64_t)__double_as_longlong(e.y);
    p.sample = (uint32_t)sf;
    p.flags = (uint32_t)(sf >> 32);
    return p;
}

} // namespace

// ================================================================================ kernels

// K1 primary_gen: render_tile's loops (renderer.rs:72-92) + Camera::get_ray (camera.rs:62-68).  The primary ray of sample s is a function of
// the frame alone: the kernels that need it (closest hit, shading) compute it where they would read a pool record, so a step of primaries
// writes and reads no ray records at all (k_generate, which stored them, is kept for RTX_GENERATE=1: A/B).
DEV uint32_t primary_flags(const DFrame &fr) {
    return (fr.reflect_depth << RAYF_REFLECT_SHIFT) | (fr.refract_depth << RAYF_REFRACT_SHIFT) | ((uint32_t)RAY_PRIMARY << RAYF_TYPE_SHIFT);
}
DEV PoolRay primary_ray(const DFrame &fr, uint32_t s) {
    PoolRay pr;
    pr.o = v3(0, 0, 0); pr.d = v3(0, 0, 1); pr.key = 0; pr.sample = s; pr.flags = RAYF_DEAD;
    uint32_t px, row, sub;
    if (!decode_sample(fr, s, &px, &row, &sub)) return pr;
    uint32_t abs_x = px, abs_y = fr.H - row - 1u;                                        // renderer.rs:73
    uint32_t x_subpixel = sub % fr.ps, y_subpixel = sub / fr.ps;
    uint64_t key = root_key(fr.seed, fr.frame, abs_y * fr.W + abs_x, sub);
    double j_x = 0.0, j_y = 0.0;
    if (fr.ps > 1) {                                                                     // renderer.rs:84-89
        double ua, ub;
        draw2(fr.rng_mode, key, STREAM_JITTER, 0, 0, &ua, &ub);
        j_x = (double)x_subpixel * fr.inv_ps + ua * fr.inv_ps;
        j_y = (double)y_subpixel * fr.inv_ps + ub * fr.inv_ps;
    }
    double x = (double)abs_x + j_x, y = (double)abs_y + j_y;
    V3 eye = v3(fr.eye[0], fr.eye[1], fr.eye[2]), right = v3(fr.right[0], fr.right[1], fr.right[2]);
    V3 up = v3(fr.up[0], fr.up[1], fr.up[2]);
    pr.d = unit(eye + scale(right, x * fr.pixel_width - fr.half_width) + scale(up, y * fr.pixel_height - fr.half_height));
    pr.o = v3(fr.pos[0], fr.pos[1], fr.pos[2]);
    pr.key = key;
    pr.flags = primary_flags(fr);
    return pr;
}
// a ray of the step: pool slot `slot`, or for a step of primaries (prim_s0 != RTX_NOT_PRIMARY) the ray of sample prim_s0 + slot
#define RTX_NOT_PRIMARY 0xFFFFFFFFu
DEV PoolRay step_ray(const DPool &pool, const DFrame &fr, uint32_t slot, uint32_t prim_s0) {
    if (prim_s0 != RTX_NOT_PRIMARY) return primary_ray(fr, prim_s0 + slot);
    return pool_load(pool, slot);
}
// (the kernels that do a frame's work have an instance of their own for steps of primaries: the others are the code they were)
template <bool PRIM>
DEV PoolRay step_ray_t(const DPool &pool, const DFrame &fr, uint32_t slot, uint32_t prim_s0) {
    if constexpr (PRIM) return primary_ray(fr, prim_s0 + slot);
    else return pool_load(pool, slot);
}
__global__ void __launch_bounds__(RTX_BLOCK) k_generate(DFrame fr, DPool pool, uint32_t pool_base, uint32_t s0, uint32_t count) {
    uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (gid >= count) return;
    uint32_t slot = pool_base + gid;
    const PoolRay pr = primary_ray(fr, s0 + gid);
    if (pr.flags & RAYF_DEAD) {
        pool.rec[slot].flags = RAYF_DEAD;
        pool.flags[slot] = RAYF_DEAD;
        return;
    }
    pool_store(pool, slot, pr.o, pr.d, v3(1.0, 1.0, 1.0), pr.key, pr.sample, pr.flags);
}

// K2 trace_closest: Ray::get_nearest_hit (ray.rs:44-66) for pool rays [begin, begin+count):
// writes the winning t and primitive ref per ray.  Kept apart from the shading so that the
// traversal runs at its own (higher) occupancy.
#ifndef RTX_CLOSEST_WAVES
#define RTX_CLOSEST_WAVES 6
#endif
template <int MODE, bool PRIM>
__global__ void __launch_bounds__(RTX_BLOCK, MODE == MODE_CLOSEST_BVH ? RTX_BVH_WAVES : (MODE == MODE_CLOSEST_BVH_DEALT ? RTX_BVH_DEALT_WAVES : RTX_CLOSEST_WAVES)) k_closest(DScene sc, DPool pool, uint32_t begin, uint32_t count,
                                                                          DCounters *counters, double *hit_t, int32_t *hit_ref, const uint32_t *order,
                                                                          DFrame fr, uint32_t prim_s0) {
    constexpr bool BVH = MODE == MODE_CLOSEST_BVH || MODE == MODE_CLOSEST_BVH_DEALT;
    __shared__ uint32_t s_stk[BVH ? RTX_BVH_STACK * RTX_BLOCK : RTX_LDS_WORDS];
    __shared__ uint32_t s_cand[MODE == MODE_CLOSEST ? RTX_CAND_WORDS : (BVH ? RTX_BVH_CAND * RTX_BLOCK : 1)];
    __shared__ __align__(8) uint32_t s_aux[MODE == MODE_CLOSEST_BVH_DEALT ? 4 * RTX_BLOCK : 2];   // traverse_bvh_dealt: best t (u64), ref and tie flag per lane
    if constexpr (!BVH) init_lds(s_stk);
    // order != NULL: the launch takes the step's rays in the order of their cells (rtxk_sort_rays): lane gid walks ray order[gid]
    // (one 64-byte record, wherever it lies) and writes that ray's hit record; everything after this kernel runs in ray order again
    uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    bool valid = gid < count;
    if (order && valid) {
        gid = order[gid];
        valid = gid < count;   // (never false for a permutation; an index past the step is not followed)
    }
    uint32_t slot = begin + (valid ? gid : 0u);
    PoolRay pr;
    pr.o = v3(0, 0, 0); pr.d = v3(0, 0, 1); pr.key = 0; pr.sample = 0; pr.flags = RAYF_DEAD;
    if (valid) pr = step_ray_t<PRIM>(pool, fr, slot, prim_s0);
    uint32_t flags = pr.flags;
    valid = valid && !(flags & RAYF_DEAD);
    uint32_t type = (flags >> RAYF_TYPE_SHIFT) & 7u;
    {
        unsigned long long *cs = counters->ray_slots[blockIdx.x % RTX_RAY_SLOTS];
        count_rays(valid && type == RAY_PRIMARY, cs + RAY_PRIMARY);
        count_rays(valid && type == RAY_REFLECT, cs + RAY_REFLECT);
        count_rays(valid && type == RAY_GLOSS, cs + RAY_GLOSS);
        count_rays(valid && type == RAY_REFRACT, cs + RAY_REFRACT);
    }
    Hit hit;
    hit.t = inf();
    hit.ref = REF_NONE;
    WORK(hit.w_nodes = hit.w_entries = hit.w_prims = 0;)
    if constexpr (MODE == MODE_CLOSEST) {
        // every lane of the wave takes part in the while-while walk (its flushes are wave-wide)
        Ray r = valid ? make_ray(pr.o, pr.d) : make_ray(v3(0, 0, 0), v3(0, 0, 1));
        const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_closest_retry, launched next
        bool tie;
        traverse_ww<WW_CLOSEST>(sc, r, valid && !degen, s_stk, s_cand, T_MIN, inf(), &hit, nullptr, nullptr, &tie);
        if (degen || tie) hit.ref = REF_RETRY;
    } else if constexpr (MODE == MODE_CLOSEST_BVH) {
        Ray r = valid ? make_ray(pr.o, pr.d) : make_ray(v3(0, 0, 0), v3(0, 0, 1));
        const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_closest_retry, launched next
        bool ovf;
        traverse_bvh<WW_CLOSEST>(sc, r, valid && !degen, s_stk, s_cand, T_MIN, inf(), &hit, &ovf);
        if (degen || ovf) hit.ref = REF_RETRY;                          // (a full stack: the all-f64 walk answers the ray)
    } else if constexpr (MODE == MODE_CLOSEST_BVH_DEALT) {
        Ray r = valid ? make_ray(pr.o, pr.d) : make_ray(v3(0, 0, 0), v3(0, 0, 1));
        const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_closest_retry, launched next
        bool ovf;
        traverse_bvh_dealt<WW_CLOSEST>(sc, r, valid && !degen, s_stk, s_cand, s_aux, T_MIN, inf(), &hit, &ovf);
        if (degen || ovf) hit.ref = REF_RETRY;
    } else if (valid) {
        Ray r = make_ray(pr.o, pr.d);
        V3 unused;
        traverse<MODE>(sc, r, s_stk, T_MIN, inf(), &hit, &unused);
    }
    WORK(count_work(valid, hit, counters->work[0]);)
    if (gid < count) {
        hit_t[gid] = hit.t;
        hit_ref[gid] = valid ? hit.ref : REF_NONE;
    }
}

// ---- the slots of a step's CHILDREN in the ray pool, without atomics.  Which children a ray spawns (renderer.rs:122-135) follows from the
// material of the prim it hit (DScene.ref_class: RTX_CLS_* per prim, built with the scene) and from its depth counters, both known as soon
// as the closest hit is: the pass that sees every ray's final hit (k_closest_retry, else k_group_count) sums them per workgroup next to
// the number of rays that hit something, k_group_base scans both, and k_shade puts a ray's children at the step's first free slot + its
// workgroup's base + the ray's rank.  (One returning atomic per workgroup on the pool's tail cost k_shade more than everything else it
// does: 32 ns per workgroup of 256 rays, serialised at one address: sponza 20.6 -> ... ms)  Children are in RAY
// ORDER, so the pool's contents no longer depend on the order in which workgroups happen to run.
DEV uint32_t ref_class_of(const DScene &sc, int ref) {
    if (ref >= 0 && !(ref & REF_PLANE)) return sc.ref_class[ref];
    if (ref >= 0) return sc.ref_class[sc.cls_planes + (ref & ~REF_PLANE)];
    return sc.ref_class[sc.cls_spheres + ~ref];
}
// the children of a ray that hit a prim of class `cls`: mirror / gloss rays (renderer.rs:126-129, 155-164), then one transmitted ray (:131-134);
// a child whose depth counter reaches 0 returns zero at renderer.rs:106 and is not a ray
DEV uint32_t spawn_counts(uint32_t cls, uint32_t flags, uint32_t gloss_samples, uint32_t *n_reflect, bool *transmit) {
    const uint32_t reflect_depth = (flags >> RAYF_REFLECT_SHIFT) & RAYF_DEPTH_MASK;
    const uint32_t refract_depth = (flags >> RAYF_REFRACT_SHIFT) & RAYF_DEPTH_MASK;
    const bool spawn_reflect = (cls & RTX_CLS_REFLECT) && reflect_depth > 1u && refract_depth > 0u;
    *transmit = (cls & RTX_CLS_REFRACT) && refract_depth > 1u && reflect_depth > 0u;
    *n_reflect = spawn_reflect ? ((cls & RTX_CLS_GLOSSY) ? gloss_samples : 1u) : 0u;
    return *n_reflect + (*transmit ? 1u : 0u);
}
// the two per-workgroup sums (every thread of the workgroup calls it): rays that hit something, children they will spawn
DEV void block_counts(const DScene &sc, const DPool &pool, uint32_t begin, uint32_t count, uint32_t gloss_samples, int32_t ref,
                      uint32_t *block_count, uint32_t *child_count, uint32_t *s_w /* 8 words */, uint32_t prim_flags /* 0: the rays' own */) {
    const uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const bool hit = gid < count && ref != REF_NONE;   // (k_closest writes REF_NONE for dead rays)
    uint32_t nc = 0u;
    if (hit) {
        uint32_t n_reflect;
        bool transmit;
        nc = spawn_counts(ref_class_of(sc, ref), prim_flags ? prim_flags : pool.flags[begin + gid], gloss_samples, &n_reflect, &transmit);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nc += (uint32_t)__shfl_xor((int)nc, off);
    const unsigned long long m = __ballot(hit);
    if (lane == 0u) { s_w[wid] = (uint32_t)__popcll(m); s_w[4u + wid] = nc; }
    __syncthreads();
    if (threadIdx.x == 0u) {
        if (block_count) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        child_count[blockIdx.x] = s_w[4] + s_w[5] + s_w[6] + s_w[7];
    }
}

// The all-f64 walk in the reference's visitation order for the rays k_closest<MODE_CLOSEST> has marked REF_RETRY (degenerate_dir).
__global__ void __launch_bounds__(RTX_BLOCK) k_closest_retry(DScene sc, DPool pool, uint32_t begin, uint32_t count, double *hit_t,
                                                             int32_t *hit_ref, uint32_t gloss_samples, uint32_t *block_count, uint32_t *child_count,
                                                             DFrame fr, uint32_t prim_s0) {
    __shared__ uint32_t s_stk[RTX_LDS_WORDS];
    __shared__ uint32_t s_w[8];
    const uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    int32_t ref = gid < count ? hit_ref[gid] : REF_NONE;
    const bool mine = ref == REF_RETRY;
    if (__syncthreads_or(mine ? 1 : 0)) {   // hardly any block
        init_lds(s_stk);
        if (mine) {
            const PoolRay pr = step_ray(pool, fr, begin + gid, prim_s0);
            const Ray r = make_ray(pr.o, pr.d);
            Hit hit;
            V3 unused;
            traverse_exact<MODE_CLOSEST_REF>(sc, r, s_stk + threadIdx.x, T_MIN, inf(), &hit, &unused);
            hit_t[gid] = hit.t;
            hit_ref[gid] = hit.ref;
            ref = hit.ref;
        }
    }
    // this pass sees every ray's final hit: the workgroup's number of rays that hit something, for the compact shadow queue, and of
    // the children they spawn, for their slots in the pool (as k_group_count, which runs where this kernel does not)
    block_counts(sc, pool, begin, count, gloss_samples, ref, block_count, child_count, s_w, prim_s0 != RTX_NOT_PRIMARY ? primary_flags(fr) : 0u);
}

// A degenerate shadow feeler (degenerate_dir) is left out of the any-hit kernels' sums and appended to the launch's retry list:
// (feeler index = group * tries + k, primary sample); k_shadow_retry answers it with the all-f64 walk.
DEV void retry_append(const DShadowQ &sq, uint32_t light, uint32_t feeler, uint32_t sample, DCounters *counters) {
    uint32_t *list = sq.retry + (size_t)light * (2u + 2u * (size_t)sq.retry_cap);
    const uint32_t pos = atomicAdd(list, 1u);
    if (pos < sq.retry_cap) {
        list[2u + 2u * pos] = feeler;
        list[3u + 2u * pos] = sample;
    } else {
        // no feeler is dropped silently: the host reads this flag after the frame's last shadow launch and renders the frame
        // again with every shadow launch on the all-f64 walk (RTX_FLAG_INTERNAL_EXACT_SHADOWS)
        counters->retry_overflow = 1u;
    }
}

// The slots of the compact shadow queue, in RAY order (groups appended in the order workgroups happen to finish would scatter
// neighbouring screen tiles over the queue: measured, sibenik shadows 6.7 -> 7.2 ms): the number of rays of each workgroup that hit
// something (k_closest_retry, or this kernel where that one does not run), the exclusive scan of these counts and their total
// (k_group_base); k_shade adds the ray's rank within its workgroup.  The children's slots in the pool: the same with the numbers of children.
static_assert(RTX_BLOCK == 256, "the workgroup counts are summed over four waves");
__global__ void __launch_bounds__(RTX_BLOCK) k_group_count(DScene sc, DPool pool, uint32_t begin, uint32_t count, const int32_t *hit_ref,
                                                           uint32_t gloss_samples, uint32_t *block_count, uint32_t *child_count, uint32_t prim_flags) {
    __shared__ uint32_t s_w[8];
    const uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    block_counts(sc, pool, begin, count, gloss_samples, gid < count ? hit_ref[gid] : REF_NONE, block_count, child_count, s_w, prim_flags);
}
// One workgroup per 256 counts, every load coalesced: a workgroup first sums the counts before its segment, then scans its own.
// (One workgroup whose threads each walked 32 consecutive counts touched a cache line per lane and load: 17 us on one CU.  Workgroups
// of 1024 threads waited 0.35 ms per launch for a CU with 16 free wave slots while shadow launches filled the chip.)
// Two scans at once: hits (block_count -> block_base, total -> *total_out; skipped when block_count is null: a frame without shadow
// feelers) and children (child_count -> child_base; the last workgroup moves the pool's tail from child_begin, where the host has put it
// for this step, past the step's children, or raises the overflow flag if they do not fit: k_shade then writes none of those).
__global__ void __launch_bounds__(RTX_BLOCK) k_group_base(const uint32_t *block_count, const uint32_t *child_count, uint32_t n_blocks, uint32_t *block_base,
                                                          uint32_t *child_base, uint32_t *total_out, uint32_t child_begin, uint32_t pool_cap,
                                                          DCounters *counters, uint32_t *host_step, uint32_t host_seq) {
    __shared__ uint32_t s_pre[2][RTX_BLOCK / 64], s_wave[2][RTX_BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t seg0 = blockIdx.x * RTX_BLOCK, i = seg0 + threadIdx.x;
    uint32_t pre[2] = {0u, 0u}, c[2], x[2];
    // (four segments per round: the loads of a round are in flight together; one by one the last workgroup of an 8 Mi-ray step took 59 us)
    uint32_t j = threadIdx.x;
    for (; j + 3u * RTX_BLOCK < seg0; j += 4u * RTX_BLOCK) {
        uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u;
        if (block_count) { a0 = block_count[j]; a1 = block_count[j + RTX_BLOCK]; a2 = block_count[j + 2u * RTX_BLOCK]; a3 = block_count[j + 3u * RTX_BLOCK]; }
        const uint32_t b0 = child_count[j], b1 = child_count[j + RTX_BLOCK], b2 = child_count[j + 2u * RTX_BLOCK], b3 = child_count[j + 3u * RTX_BLOCK];
        pre[0] += (a0 + a1) + (a2 + a3);
        pre[1] += (b0 + b1) + (b2 + b3);
    }
    for (; j < seg0; j += RTX_BLOCK) {
        if (block_count) pre[0] += block_count[j];
        pre[1] += child_count[j];
    }
    c[0] = (block_count && i < n_blocks) ? block_count[i] : 0u;
    c[1] = i < n_blocks ? child_count[i] : 0u;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        x[a] = c[a];   // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x[a], off);
            if ((int)lane >= off) x[a] += y;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pre[a] += (uint32_t)__shfl_xor((int)pre[a], off);
        if (lane == 63u) s_wave[a][wid] = x[a];
        if (lane == 0u) s_pre[a][wid] = pre[a];
    }
    __syncthreads();
    uint32_t before[2] = {0u, 0u}, total[2] = {0u, 0u};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (uint32_t wv = 0; wv < RTX_BLOCK / 64; ++wv) {
            before[a] += s_pre[a][wv] + (wv < wid ? s_wave[a][wv] : 0u);
            total[a] += s_pre[a][wv] + s_wave[a][wv];
        }
    if (i < n_blocks) {
        if (block_count) block_base[i] = before[0] + x[0] - c[0];
        child_base[i] = before[1] + x[1] - c[1];
    }
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0u) {
        if (block_count) *total_out = total[0];
        const bool ovf = (uint64_t)child_begin + total[1] > (uint64_t)pool_cap;
        if (ovf) counters->overflow = 1u;
        else counters->pool_tail = child_begin + total[1];
        // What the host needs to size the NEXT step is known here, before k_shade has run: told to it at once, by a store into pinned host
        // memory that it polls (tail, overflow, then the step's sequence number with release semantics at system scope), instead of a
        // copy and an event after k_shade: the host enqueues the next closest-hit launch while this step is still shading.
        if (host_step) {
            host_step[0] = ovf ? child_begin : child_begin + total[1];
            host_step[1] = ovf ? 1u : 0u;
            __hip_atomic_store(host_step + 2, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// K3 (+K5) shade_and_spawn: the body of Renderer::trace after the closest-hit search
// (renderer.rs:108-144): local shading per light -> one shadow group per (ray that hit, light), in ray order without gaps
// (k_closest_retry / k_group_base give every ray its slot);
// reflection / gloss / transmission children -> ray pool; miss -> skybox / background.
#define SHADOW_GROUP_EMPTY 0xFFFFFFFFu
#ifndef RTX_SHADE_WAVES
#define RTX_SHADE_WAVES 2
#endif
#ifndef RTX_SHADE_WAVES_LEAN
#define RTX_SHADE_WAVES_LEAN 5
#endif
// GENERAL = false: the SMALL instance, 96 VGPRs at 5 waves per SIMD (the general one: 208 at 2), for frames whose shadow feelers run in the
// cooperative kernel next to it (RTX_FLAG_INTERNAL_SMALL_SHADE: 16 tries per hit point or a multiple, overlap on).  k_shadow_coop keeps six
// 80-VGPR waves per SIMD resident and the VALUs 94 % busy; a workgroup of the general instance needs 208 VGPRs on all four SIMDs of a CU at
// once and waits for them (9 ms per launch under overlap for 1.2 ms alone), the small one fits into the slot of one or two shadow waves.  It is
// slower alone (spills: bunny 5.98 -> 7.6 ms if it were used there), so it runs only where the shadow kernel dominates: sponza frame 208.6 ->
// 204.6 ms.  Compiled without the f64 pow / acos / sin / cos of Ray::perturb (glossy reflection) and of the Phong term, which would spill far
// more: scenes that need them (DScene.shade_features != 0) always take the general instance.
// the cell of a hit point in the scene's cube, 8 bits per axis, bits interleaved (Morton order): the sort key of its shadow groups
// (DShadowQ.cell).  Points outside the cube (planes that run to the horizon) clamp to its faces.
DEV uint32_t spread8(uint32_t x) {
    x = (x | (x << 8)) & 0x00F00Fu;
    x = (x | (x << 4)) & 0x0C30C3u;
    x = (x | (x << 2)) & 0x249249u;
    return x;
}
DEV uint32_t hit_cell(const DScene &sc, V3 p);
// sort key of a child ray (DPool.cell): direction octant over the 21-bit cell of the origin
DEV void pool_store_cell(const DPool &pool, const DScene &sc, uint32_t slot, V3 o, V3 d) {
    if (!pool.cell) return;
    const uint32_t oct = (d.x > 0.0 ? 1u : 0u) | (d.y > 0.0 ? 2u : 0u) | (d.z > 0.0 ? 4u : 0u);
    pool.cell[slot] = (oct << 21) | (hit_cell(sc, o) >> 3);
}
DEV uint32_t hit_cell(const DScene &sc, V3 p) {
    const float s = 128.0f / (float)sc.radius;
    const float fx = fminf(fmaxf(((float)(p.x - sc.centre[0])) * s + 128.0f, 0.0f), 255.0f);
    const float fy = fminf(fmaxf(((float)(p.y - sc.centre[1])) * s + 128.0f, 0.0f), 255.0f);
    const float fz = fminf(fmaxf(((float)(p.z - sc.centre[2])) * s + 128.0f, 0.0f), 255.0f);
    return spread8((uint32_t)fx) | (spread8((uint32_t)fy) << 1) | (spread8((uint32_t)fz) << 2);
}
template <bool GENERAL, bool PRIM>
__global__ void __launch_bounds__(RTX_BLOCK, GENERAL ? RTX_SHADE_WAVES : RTX_SHADE_WAVES_LEAN) k_shade(DScene sc, DFrame fr, DPool pool, uint32_t begin, uint32_t count,
                                                     const double *hit_t, const int32_t *hit_ref, DShadowQ sq, DCounters *counters,
                                                     double *acc, int32_t *prim_id_out, const uint32_t *group_base,
                                                     const uint32_t *child_base, uint32_t child_begin, uint32_t prim_s0) {
    __shared__ uint32_t s_tmp[8];
    uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    bool in_range = gid < count;
    bool valid = in_range;
    uint32_t slot = begin + (valid ? gid : 0u);
    // The kernel waits for its fetches at 2 waves per SIMD, so what can be in flight together is asked for together: the ray's record
    // and hit with its flags (a dead ray's slot holds a ray all the same), then, below, the prim's class and its two records.
    uint32_t flags = RAYF_DEAD;

    size_t cap = pool.cap;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1), thr = v3(0, 0, 0);
    uint64_t key = 0;
    uint32_t sample = 0;
    Hit hit;
    hit.t = inf();
    hit.ref = REF_NONE;
    if (in_range) {
        const PoolRay pr = step_ray_t<PRIM>(pool, fr, slot, prim_s0);
        const double *t3 = pool.thr + slot;
        o = pr.o; d = pr.d; key = pr.key; sample = pr.sample; flags = pr.flags;
        if constexpr (PRIM) thr = v3(1.0, 1.0, 1.0);
        else thr = v3(t3[0 * cap], t3[1 * cap], t3[2 * cap]);
        hit.t = hit_t[gid];
        hit.ref = hit_ref[gid];
    }
    valid = valid && !(flags & RAYF_DEAD);
    if (!valid) { o = v3(0, 0, 0); d = v3(0, 0, 1); thr = v3(0, 0, 0); key = 0; sample = 0; hit.t = inf(); hit.ref = REF_NONE; }
    // The ray's shadow groups (one per light, all at the same index) in the compact queue and its children in the pool, both in ray
    // order: base of the workgroup (k_closest_retry / k_group_base) + rank within it (a ballot and a wave scan, one barrier).  What the
    // ray spawns follows from the class of the prim it hit and its depth counters (spawn_counts: the same function the counting pass calls).
    const bool is_hit = valid && hit.ref != REF_NONE;
    const bool is_tri = is_hit && hit.ref >= 0 && !(hit.ref & REF_PLANE);
    double2 tv[5] = {}, ts[8] = {};   // the triangle's vertex and shading records, asked for before the barrier below (used after it)
    if (is_tri) {
        const double2 *p5 = reinterpret_cast<const double2 *>(sc.tris + hit.ref), *p8 = reinterpret_cast<const double2 *>(sc.tri_shade + hit.ref);
#pragma unroll
        for (int k = 0; k < 5; ++k) tv[k] = p5[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) ts[k] = p8[k];
    }
    const uint32_t cls = is_hit ? ref_class_of(sc, hit.ref) : 0u;
    uint32_t n_reflect = 0u;
    bool spawn_transmit = false;
    uint32_t n_children = is_hit ? spawn_counts(cls, flags, fr.gloss_samples, &n_reflect, &spawn_transmit) : 0u;
    uint32_t gslot = 0u, cs = 0u;
    {
        const uint32_t ln = threadIdx.x & 63u, wv_ = threadIdx.x >> 6;
        const unsigned long long hm = __ballot(is_hit);
        uint32_t x = n_children;   // inclusive scan over the wave
        if (__any(n_children != 0u)) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)x, off);
                if ((int)ln >= off) x += y;
            }
        }
        uint32_t before_g = (fr.shadow_samples != 0u && sc.n_lights > 0) ? group_base[blockIdx.x] : 0u;
        uint32_t before_c = child_begin + child_base[blockIdx.x];
        if (ln == 0u) s_tmp[wv_] = (uint32_t)__popcll(hm);
        if (ln == 63u) s_tmp[4u + wv_] = x;
        __syncthreads();
        for (uint32_t q = 0; q < wv_; ++q) { before_g += s_tmp[q]; before_c += s_tmp[4u + q]; }
        gslot = before_g + (uint32_t)__popcll(hm & ((1ull << ln) - 1ull));
        cs = before_c + x - n_children;
    }
    uint32_t reflect_depth = (flags >> RAYF_REFLECT_SHIFT) & RAYF_DEPTH_MASK;
    uint32_t refract_depth = (flags >> RAYF_REFRACT_SHIFT) & RAYF_DEPTH_MASK;
    bool inside = (flags & RAYF_INSIDE) != 0u;
    uint32_t type = (flags >> RAYF_TYPE_SHIFT) & 7u;
    Ray r = make_ray(o, d);

    // ---- full Intersection of the winning prim (recomputed: same arithmetic, same bits)
    V3 hn = v3(0, 0, 1), hp = v3(0, 0, 0);
    double hu = 0.0, hv = 0.0;
    int material = 0, prim_id = -1;
    if (is_hit) {
        double t = hit.t;
        hp = o + scale(d, t);                                                            // origin + direction.scale(t)
        if (is_tri) {                                                                    // triangle.rs:136-145
            TriV tr;   // (as load_tri)
            tr.v0 = v3(tv[0].x, tv[0].y, tv[1].x);
            tr.v1 = v3(tv[1].y, tv[2].x, tv[2].y);
            tr.v2 = v3(tv[3].x, tv[3].y, tv[4].x);
            const long long ids = __double_as_longlong(tv[4].y);
            prim_id = (int)(ids & 0xFFFFFFFFLL);
            material = (int)(ids >> 32);
            double tt, beta, gamma;
            tri_hit(tr, r, -inf(), inf(), &tt, &beta, &gamma);
            double alpha = 1.0 - beta - gamma;
            double2 a0 = ts[0], a1 = ts[1], a2 = ts[2], a3 = ts[3], a4 = ts[4], a5 = ts[5], a6 = ts[6], a7 = ts[7];
            V3 n0 = v3(a0.x, a0.y, a1.x), n1 = v3(a1.y, a2.x, a2.y), n2 = v3(a3.x, a3.y, a4.x);
            double u0 = a4.y, v0 = a5.x, u1 = a5.y, v1 = a6.x, u2 = a6.y, v2 = a7.x;
            hn = scale(n0, alpha) + scale(n1, beta) + scale(n2, gamma);
            hu = u0 * alpha + u1 * beta + u2 * gamma;
            hv = v0 * alpha + v1 * beta + v2 * gamma;
        } else if (hit.ref >= 0) {                                                       // plane.rs:35-39
            const DPlane &pl = sc.planes[hit.ref & ~REF_PLANE];
            prim_id = pl.prim_id; material = pl.material;
            V3 n = v3(pl.abcd[0], pl.abcd[1], pl.abcd[2]);
            V3 u_axis = v3(n.y, n.z, -n.x);
            V3 v_axis = cross(u_axis, n);
            hn = n;
            hu = dot(hp, u_axis);
            hv = dot(hp, v_axis);
        } else {                                                                         // sphere.rs:55-59
            const DSphere &sp_ = sc.spheres[~hit.ref];
            prim_id = sp_.prim_id; material = sp_.material;
            V3 n = unit(hp - v3(sp_.c[0], sp_.c[1], sp_.c[2]));
            hn = n;
            hu = 0.5 + atan2(n.z, n.x) / (PI * 2.0);
            hv = 0.5 - asin(n.y) / PI;
        }
    }

    if (valid && type == RAY_PRIMARY && prim_id_out) {
        uint32_t px, row, sub;
        decode_sample(fr, sample, &px, &row, &sub);
        if (sub == 0u) prim_id_out[(size_t)out_row(fr, row) * fr.W + px] = is_hit ? prim_id : -1;
    }

    // ---- miss: skybox / background (renderer.rs:139-144)
    {
        V3 c = v3(0.0, 0.0, 0.0);
        if (valid && !is_hit) c = sc.has_skybox ? cubemap_color(sc, d) : v3(sc.background[0], sc.background[1], sc.background[2]);
        acc_add_runs(acc, valid && !is_hit, sample, thr * c);
    }

    // ---- hit: local lighting (renderer.rs:110-119)
    const rtx_material &m = sc.materials[material];
    V3 n = v3(0, 0, 1), iv = v3(0, 0, 1);
    if (is_hit) {
        n = unit(hn);
        iv = unit(-d);
    }

    MatHit mh{};
    if (is_hit && sc.n_lights > 0) mh = material_prepare(sc, m, hu, hv);
    for (int li = 0; li < sc.n_lights; ++li) {
        bool emit = false;
        V3 w = v3(0, 0, 0);
        if (is_hit) {
            const rtx_light &lt = sc.lights[li];
            V3 l = unit(v3(lt.position[0], lt.position[1], lt.position[2]) - hp);
#ifdef RTX_ABL_SHADE_NOBRDF
            V3 smp = v3(dot(n, l), hu, hv);
#else
            V3 smp = material_sample<GENERAL>(mh, n, iv, l);
#endif
            w = thr * (v3(lt.color[0], lt.color[1], lt.color[2]) * smp);
            emit = fr.shadow_samples != 0u;
        }
        if (fr.shadow_samples == 0u) acc_add_runs(acc, is_hit, sample, w);               // renderer.rs:191: shadow = one
        if (fr.shadow_samples != 0u) count_rays(emit, counters->ray_slots[blockIdx.x % RTX_RAY_SLOTS] + RTX_RAY_SLOT_GROUPS + li);
        if (emit) {
            size_t qc = sq.cap;
            sq.sample[(size_t)li * qc + gslot] = sample;
            if (li == 0 && sq.sample0) sq.sample0[gslot] = sample;
            if (li == 0 && sq.cell) sq.cell[gslot] = hit_cell(sc, hp);
            double *f = sq.f + (size_t)li * 6 * qc + gslot;
            f[0 * qc] = hp.x; f[1 * qc] = hp.y; f[2 * qc] = hp.z;
            f[3 * qc] = w.x; f[4 * qc] = w.y; f[5 * qc] = w.z;
            sq.key[(size_t)li * qc + gslot] = key;
        }
    }

    // ---- global lighting: spawn the child trace() calls (renderer.rs:122-135, 148-186)
    const bool reflective = (cls & RTX_CLS_REFLECT) != 0u, refractive = (cls & RTX_CLS_REFRACT) != 0u;   // (cls = 0 for a miss)
    double reflect_fresnel = 0.0;
    if (reflective || refractive) reflect_fresnel = fresnel_reflect(m.ior, iv, n, inside);
    double refract_fresnel = 1.0 - reflect_fresnel;

    const bool spawn_reflect = n_reflect != 0u;
    const bool glossy = GENERAL && (cls & RTX_CLS_GLOSSY) != 0u;
    if (n_children && (uint64_t)cs + n_children > (uint64_t)pool.cap) {   // (k_group_base has raised the overflow flag: the host gives the frame up)
        counters->overflow = 1u;
        n_reflect = 0u;
        spawn_transmit = false;
    }
    V3 rdir = v3(0, 0, 1);
    if (spawn_reflect) rdir = reflect(iv, n);
    uint32_t child_flags_reflect = ((reflect_depth - 1u) << RAYF_REFLECT_SHIFT) | (refract_depth << RAYF_REFRACT_SHIFT) |
                                   (inside ? RAYF_INSIDE : 0u);
    if (n_reflect && !glossy) {   // mirror reflection (renderer.rs:164)
        pool_store(pool, cs, hp, rdir, scale(scale(thr, m.k_sg), reflect_fresnel), child_key(key, CODE_MIRROR), sample,
                   child_flags_reflect | ((uint32_t)RAY_REFLECT << RAYF_TYPE_SHIFT));
        pool_store_cell(pool, sc, cs, hp, rdir);
    } else if (GENERAL && n_reflect) {       // glossy reflection: gloss_samples perturbed rays, averaged (renderer.rs:155-161)
        double inv_g = 1.0 / (double)fr.gloss_samples;
        for (uint32_t g = 0; g < n_reflect; ++g) {
            // Ray::perturb (ray.rs:68-81) with Vec3::random (vec3.rs:112-126)
            uint64_t ck = child_key(key, CODE_GLOSS0 + g);
            double u_phi, u_cos, u_r;
            draw3(fr.rng_mode, ck, STREAM_GLOSS, 0, 0, &u_phi, &u_cos, &u_r);
            double phi = 0.0 + (2.0 * PI - 0.0) * u_phi;
            double costheta = -1.0 + (1.0 - -1.0) * u_cos;
            double uu = 0.0 + (1.0 - 0.0) * u_r;
            double theta = acos(costheta);
            double rr = pow(uu, 1.0 / 3.0);
            V3 rand_vec = scale(v3(rr * sin(theta) * cos(phi), rr * sin(theta) * sin(phi), rr * cos(theta)), m.glossiness);
            V3 corrected = dot(rand_vec, rdir) < 0.0 ? scale(rand_vec, -1.0) : rand_vec;
            V3 gdir = unit(corrected + rdir);
            pool_store(pool, cs + g, hp, gdir, scale(scale(scale(thr, inv_g), m.k_sg), reflect_fresnel), ck, sample,
                       child_flags_reflect | ((uint32_t)RAY_GLOSS << RAYF_TYPE_SHIFT));
            pool_store_cell(pool, sc, cs + g, hp, gdir);
        }
    }
    if (spawn_transmit) {   // transmission (renderer.rs:170-186): TIR -> mirror direction with weight 1.0, `!inside` either way
        V3 tdir = v3(0, 0, 1);
        double actual = refract_fresnel;
        if (!refract(iv, n, m.ior, inside, &tdir)) {
            tdir = reflect(iv, n);
            actual = 1.0;
        }
        uint32_t cf = (reflect_depth << RAYF_REFLECT_SHIFT) | ((refract_depth - 1u) << RAYF_REFRACT_SHIFT) |
                      (inside ? 0u : RAYF_INSIDE) | ((uint32_t)RAY_REFRACT << RAYF_TYPE_SHIFT);
        pool_store(pool, cs + n_reflect, hp + scale(tdir, EPSILON), tdir, scale(scale(thr, m.k_tg), actual),
                   child_key(key, CODE_TRANSMIT), sample, cf);
        pool_store_cell(pool, sc, cs + n_reflect, hp, tdir);
    }
}

// K4 trace_shadow: Renderer::shadow_intensity (renderer.rs:188-221), one feeler per lane; the
// `tries` feelers of one (hit, light) pair sit in adjacent lanes and are reduced with
// cross-lane shuffles before ONE set of atomics per group.
#ifndef RTX_SHADOW_WAVES
#define RTX_SHADOW_WAVES 6
#endif
#ifndef RTX_PRODUCT_WAVES
#define RTX_PRODUCT_WAVES 5   // the shadow product walk over the octree: 96 VGPRs, no spills; heptoroid 1 420 -> 1 371 ms against 6
#endif
#ifndef RTX_PRODUCT_BVH_WAVES
#define RTX_PRODUCT_BVH_WAVES 4   // the two-phase shadow product: 128 VGPRs, no spills (at 5: 84 B of scratch per lane); heptoroid 703 -> 666 ms
#endif
template <int MODE, bool LOOP>
DEV void shadow_kernel_body(const DScene &sc, const DFrame &fr, const DShadowQ &sq, uint32_t light, uint32_t n_groups, uint32_t tries, uint32_t pow2_tries,
                            DCounters *counters, double *acc, const uint32_t *group_list) {
    // group_list != nullptr: the launch works through a list of queue groups (what a cooperative kernel left over: count, then the
    // groups) instead of the whole queue: with a full-size grid whose blocks past the list's end leave at once (k_shadow), or with a
    // small grid whose workgroups take the list's blocks in turn (k_shadow_list: LOOP, below)
    constexpr bool BVH = MODE == MODE_ANY_BVH || MODE == MODE_PRODUCT_BVH;
    __shared__ uint32_t s_stk[BVH ? RTX_BVH_STACK * RTX_BLOCK : RTX_LDS_WORDS];
    __shared__ uint32_t s_cand[(MODE == MODE_ANY || MODE == MODE_PRODUCT_FAST) ? RTX_CAND_WORDS : (BVH ? RTX_BVH_CAND * RTX_BLOCK : 1)];
    __shared__ uint32_t s_hits[MODE == MODE_PRODUCT_BVH ? RTX_BVH_HITS * RTX_BLOCK : 1];
    __shared__ uint32_t s_aux[MODE == MODE_PRODUCT_BVH ? RTX_BLOCK : 1];   // traverse_bvh_dealt: the length of the hit list, per lane
    if constexpr (!BVH) init_lds(s_stk);
    // LOOP: a launch that works through a list (group_list) with a SMALL grid: its workgroups take the list's blocks of RTX_BLOCK
    // feelers in turn (the list's length is known on the device only; a full-size grid of workgroups that find nothing to do costs
    // 0.14 ms per launch).  An instance of its own: the loop costs the per-lane walk 24 spilled VGPRs, which the launches that do
    // the frame's work must not pay.  !LOOP: one pass, blk = blockIdx.x.
    static_assert(!LOOP || MODE == MODE_ANY_BVH || MODE == MODE_ANY, "list launches with a small grid: the any-hit walks");
    for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {
    uint64_t gid = (uint64_t)blk * RTX_BLOCK + threadIdx.x;
    const uint32_t tries_shift = (uint32_t)__ffs((int)tries) - 1u;   // (no 64-bit division for the usual try counts)
    uint32_t group = pow2_tries ? (uint32_t)(gid >> tries_shift) : (uint32_t)(gid / tries);
    uint32_t k = pow2_tries ? ((uint32_t)gid & (tries - 1u)) : (uint32_t)(gid % tries);
    // the queue is compact: *sq.count groups (<= n_groups, the rays of the step); fetched together with the group's entry, see k_shadow_coop
    const uint32_t n_act = group_list ? n_groups : *sq.count;
    bool valid = group < n_groups;
    if (group_list) {
        const uint32_t n_list = group_list[0];
        if ((uint64_t)blk * RTX_BLOCK >= (uint64_t)n_list * tries) return;   // (the whole workgroup)
        valid = group < n_list;
        group = valid ? group_list[1u + group] : 0u;
        valid = valid && group < n_groups;   // (a list is the device's own work: an index past the queue is never followed)
    }
    const uint32_t feeler = group * tries + k;   // its index in the launch's queue (retry list entries)
    size_t qc = sq.cap;
    V3 hp = v3(0, 0, 0), w = v3(0, 0, 0);
    uint64_t key = 0;
    uint32_t sample = 0;
    if (valid) sample = sq.sample[(size_t)light * qc + group];
    if (!group_list && (uint64_t)blk * RTX_BLOCK >= (uint64_t)n_act * tries) return;   // (the whole workgroup)
    valid = valid && group < n_act && sample != SHADOW_GROUP_EMPTY;
    if (valid) {
        const double *f = sq.f + (size_t)light * 6 * qc + group;
        hp = v3(f[0 * qc], f[1 * qc], f[2 * qc]);
        w = v3(f[3 * qc], f[4 * qc], f[5 * qc]);
        key = sq.key[(size_t)light * qc + group];
    }

    V3 v = v3(0, 0, 0);
    Ray r = make_ray(v3(0, 0, 0), v3(0, 0, 1));
    double distance_to_light = 0.0;
    if (valid) {
        const rtx_light &lt = sc.lights[light];
        V3 sampled = v3(lt.position[0], lt.position[1], lt.position[2]);
        if (lt.kind != RTX_LIGHT_POINT) {                                                // spherelight.rs:13-23
            double ux, uy, uz;
            draw3(fr.rng_mode, key, STREAM_LIGHT, light, k, &ux, &uy, &uz);
            V3 jitter = v3(lt.radius * (ux - 0.5), lt.radius * (uy - 0.5), lt.radius * (uz - 0.5));
            sampled = sampled + jitter;
        }
        V3 shadow_l = unit(sampled - hp);                                                // renderer.rs:203
        r = make_ray(hp, shadow_l);                                                      // NO origin offset (renderer.rs:204)
        distance_to_light = len(sampled - hp);
    }
    if constexpr (MODE == MODE_ANY) {
        // every lane of the wave takes part in the while-while walk (its flushes are wave-wide)
        Hit hit;
        const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_shadow_retry; counts as 0 here
        traverse_ww<WW_ANY>(sc, r, valid && !degen, s_stk, s_cand, EPSILON, distance_to_light, &hit);
        if (valid) v = (hit.ref == REF_NONE && !degen) ? v3(1.0, 1.0, 1.0) : v3(0.0, 0.0, 0.0);
        if (degen) retry_append(sq, light, feeler, sample, counters);
        WORK(count_work(valid, hit, counters->work[1]);)
    } else if constexpr (MODE == MODE_ANY_BVH) {
        Hit hit;
        const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_shadow_retry; counts as 0 here
        bool ovf;
        traverse_bvh<WW_ANY>(sc, r, valid && !degen, s_stk, s_cand, EPSILON, distance_to_light, &hit, &ovf);
        const bool retry = degen || (valid && ovf);                     // (a full stack: the all-f64 walk answers the feeler)
        if (valid) v = (hit.ref == REF_NONE && !retry) ? v3(1.0, 1.0, 1.0) : v3(0.0, 0.0, 0.0);
        if (retry) retry_append(sq, light, feeler, sample, counters);
        WORK(count_work(valid, hit, counters->work[1]);)
    } else if constexpr (MODE == MODE_PRODUCT_BVH) {
        Hit hit;
        V3 pr = v3(0, 0, 0);
        bool fb = false;
        traverse_bvh_dealt<WW_PRODUCT>(sc, r, valid, s_stk, s_cand, s_aux, EPSILON, distance_to_light, &hit, &fb, s_hits, &pr);
        if (valid) v = pr;
        if (valid && fb) {   // axis-parallel feeler or a full stack: the all-f64 restatement answers it (k_shadow_retry: its share of the
                             // group's average is added there, the sum over the tries is linear)
            v = v3(0.0, 0.0, 0.0);
            retry_append(sq, light, feeler, sample, counters);
        }
        WORK(count_work(valid, hit, counters->work[1]);)
    } else if constexpr (MODE == MODE_PRODUCT_FAST) {
        Hit hit;
        V3 pr = v3(0, 0, 0);
        bool fb = false;
        traverse_ww<WW_PRODUCT>(sc, r, valid, s_stk, s_cand, EPSILON, distance_to_light, &hit, &pr, &fb);
        if (valid) v = pr;
        if (valid && fb) {   // axis-parallel feeler: the all-f64 restatement of the reference's walk
            Hit h2;
            traverse_exact<MODE_PRODUCT>(sc, r, s_stk + threadIdx.x, EPSILON, distance_to_light, &h2, &v);
        }
        WORK(count_work(valid, hit, counters->work[1]);)
    } else if (valid) {
        Hit hit;
        traverse<MODE>(sc, r, s_stk, EPSILON, distance_to_light, &hit, &v);
        WORK(count_work(true, hit, counters->work[1]);)
    }
    double inv_tries = 1.0 / (double)tries;
    if ((MODE == MODE_ANY || MODE == MODE_ANY_BVH) && pow2_tries) {
        // any-hit answers are 0 or 1: the group's sum is a population count (exact, like the butterfly below)
        const uint32_t ln = threadIdx.x & 63u;
        const unsigned long long clear = __ballot(valid && v.x != 0.0);
        const unsigned long long mine = tries >= 64u ? ~0ull : (((1ull << tries) - 1ull) << (ln & ~(tries - 1u)));
        const double n_clear = (double)__popcll(clear & mine);
        shadow_result(sq, acc, light, valid && k == 0u, group, sample, w, scale(v3(n_clear, n_clear, n_clear), inv_tries));   // renderer.rs:220,118
    } else if (pow2_tries) {
        // groups are lane-aligned (tries divides 64 and RTX_BLOCK): butterfly sum over the group
        for (uint32_t off = tries >> 1; off > 0; off >>= 1) {
            v.x += __shfl_xor(v.x, (int)off);
            v.y += __shfl_xor(v.y, (int)off);
            v.z += __shfl_xor(v.z, (int)off);
        }
        shadow_result(sq, acc, light, valid && k == 0u, group, sample, w, scale(v, inv_tries));   // renderer.rs:220,118
    } else {
        acc_add_runs(acc, valid, sample, w * scale(v, inv_tries));   // (the host launches these with sq.res = null: a group's tries span waves)
    }
    if constexpr (!LOOP) return;
    }
}

#define RTX_SHADOW_BOUNDS(MODE) __launch_bounds__(RTX_BLOCK, MODE == MODE_ANY_BVH ? RTX_BVH_WAVES : (MODE == MODE_PRODUCT_BVH ? RTX_PRODUCT_BVH_WAVES : (MODE == MODE_PRODUCT_FAST ? RTX_PRODUCT_WAVES : RTX_SHADOW_WAVES)))
template <int MODE>
__global__ void RTX_SHADOW_BOUNDS(MODE) k_shadow(DScene sc, DFrame fr, DShadowQ sq, uint32_t light, uint32_t n_groups, uint32_t tries, uint32_t pow2_tries,
                                                 DCounters *counters, double *acc, const uint32_t *group_list) {
    shadow_kernel_body<MODE, false>(sc, fr, sq, light, n_groups, tries, pow2_tries, counters, acc, group_list);
}
// the same through a list with a small grid (see LOOP above)
template <int MODE>
__global__ void RTX_SHADOW_BOUNDS(MODE) k_shadow_list(DScene sc, DFrame fr, DShadowQ sq, uint32_t light, uint32_t n_groups, uint32_t tries, uint32_t pow2_tries,
                                                      DCounters *counters, double *acc, const uint32_t *group_list) {
    shadow_kernel_body<MODE, true>(sc, fr, sq, light, n_groups, tries, pow2_tries, counters, acc, group_list);
}

// The retry list of one any-hit shadow launch (degenerate feelers, see degenerate_dir): each feeler is rebuilt exactly as k_shadow
// builds it and answered by the all-f64 restatement of the reference's walk; its share of the group's average is added on its own
// (the sum over the tries is linear).  A small fixed grid: the list is almost always empty.
__global__ void __launch_bounds__(RTX_BLOCK) k_shadow_retry(DScene sc, DFrame fr, DShadowQ sq, uint32_t light, uint32_t tries, double *acc) {
    __shared__ uint32_t s_stk[RTX_LDS_WORDS];
    const uint32_t *list = sq.retry + (size_t)light * (2u + 2u * (size_t)sq.retry_cap);
    uint32_t n = list[0];
    if (n == 0u) return;
    if (n > sq.retry_cap) n = sq.retry_cap;   // overflow: counters->retry_overflow is set and the host renders the frame again
    init_lds(s_stk);
    const size_t qc = sq.cap;
    const double inv_tries = 1.0 / (double)tries;
    for (uint32_t i = blockIdx.x * RTX_BLOCK + threadIdx.x; i < n; i += gridDim.x * RTX_BLOCK) {
        const uint32_t feeler = list[2u + 2u * i], sample = list[3u + 2u * i];
        const uint32_t group = feeler / tries, k = feeler % tries;
        const double *f = sq.f + (size_t)light * 6 * qc + group;
        const V3 hp = v3(f[0 * qc], f[1 * qc], f[2 * qc]), w = v3(f[3 * qc], f[4 * qc], f[5 * qc]);
        const uint64_t key = sq.key[(size_t)light * qc + group];
        const rtx_light &lt = sc.lights[light];
        V3 sampled = v3(lt.position[0], lt.position[1], lt.position[2]);
        if (lt.kind != RTX_LIGHT_POINT) {                                                // spherelight.rs:13-23
            double ux, uy, uz;
            draw3(fr.rng_mode, key, STREAM_LIGHT, light, k, &ux, &uy, &uz);
            sampled = sampled + v3(lt.radius * (ux - 0.5), lt.radius * (uy - 0.5), lt.radius * (uz - 0.5));
        }
        const Ray r = make_ray(hp, unit(sampled - hp));
        Hit hit;
        V3 v;
        traverse_exact<MODE_PRODUCT>(sc, r, s_stk + threadIdx.x, EPSILON, len(sampled - hp), &hit, &v);
        acc_add(acc, sample, w * scale(v, inv_tries));
    }
}

// K4' cooperative any-hit shadows over the 16-wide box lists (all-opaque scenes).
//
// 16 consecutive lanes (with shadow_samples = 16: the 16 feelers of one hit point towards one
// light) walk the hierarchy TOGETHER: the group keeps one stack in LDS, every step each lane
// tests ONE of the node's 16 child boxes against the group's BEAM — the hull of the 16 rays,
// origins in [omin, omax], inverse directions in [imin, imax] per axis — and a ballot turns the 16
// answers into "children to push" and "primitives to test".  Primitives that the beam touches
// are queued in LDS and then tested by every lane against its OWN ray: conservative f32 box
// test, then the reference's exact f64 entry-bbox and primitive tests (octree.rs:170,
// renderer.rs:212), so the per-ray answer is the exact any-hit answer.  A beam test is interval
// arithmetic, hence conservative; an axis on which the group's directions change sign is left
// unconstrained.
#define RTX_COOP_STACK 48     // words: 24 spilled levels of 2 words (host checks 2 * depth + 2 <= 24)
#define RTX_COOP_QUEUE 40     // RTX_COOP_QT - 1 + 2 * 16 at most
#ifndef RTX_COOP_QT
#define RTX_COOP_QT 1      // drain the candidate queue once it holds this many primitives (with the DP-collapsed lists: 1: 176.2, 2..4: 177.5, 6: 178.2, 9: 180.3 ms; an occluder found early ends walks early)
#endif
#ifndef RTX_COOP_WAVES
#define RTX_COOP_WAVES 6     // one hit point per group: 80 VGPRs (5 spilled around the exact tests) and 26 KB of LDS; 194.8 against 198.5 ms at 5
#endif
#ifndef RTX_COOP_WAVES_MULTI
#define RTX_COOP_WAVES_MULTI 5   // several hit points per group: 16 spills at 6 (sibenik frame 12.7 -> 14.0 ms)
#endif
#define RTX_COOP_PAIRS 128    // 63 waiting + 64 pushed in one step at most
#ifndef RTX_COOP_DEFER_ENTRY
#define RTX_COOP_DEFER_ENTRY 1   // the f64 entry-bbox test of a shadow candidate runs once per stopped feeler, after the walk (see run_pairs)
#endif
static_assert(RTX_COOP_QT - 1 + 32 <= RTX_COOP_QUEUE, "primitive queue too small for the drain threshold");
// a waiting pair in one word: prim kind (2 bits) | owner lane and surely_in (7 bits) | prim index (23 bits: the host keeps scenes with
// 2^23 triangles or spheres or more off this kernel)
#define RTX_PAIR_INDEX 0x007FFFFFu
DEV uint32_t pair_pack(uint32_t pref, uint32_t powner) { return (pref & RTX_WREF_KIND) | ((powner & 127u) << 23) | (pref & RTX_PAIR_INDEX); }
#ifndef RTX_COOP_PT
#define RTX_COOP_PT 1         // after a queue drain, run the exact tests once this many pairs wait
#endif
DEV float f32_below(double v) {   // a float <= v
    float f = (float)v;
    return f - fabsf(f) * 1.1920929e-7f;
}
// min / max over the 16 lanes of a DPP row (= one cooperative group), result in every lane: an xor
// butterfly of DPP-modified v_min/v_max (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror): four VALU instructions per value and no LDS traffic (a __shfl_xor is a ds_bpermute).
// Written as asm because fminf() of a DPP move costs three instructions per step (move, canonicalise,
// min).  A VALU result needs two wait states before a DPP operand reads it: the pair form interleaves
// two butterflies so that one s_nop per step is enough.  All 64 lanes must be active.
DEV void grp_min_max(float lo_in, float hi_in, float *lo_out, float *hi_out) {
    float a, b;
    asm("s_nop 1\n\t"
        "v_min_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(a), "=&v"(b)
        : "v"(lo_in), "v"(hi_in));
    *lo_out = a;
    *hi_out = b;
}
DEV float grp_max(float v) {
    float a;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(a)
        : "v"(v));
    return a;
}
// MULTI = false: the 16 lanes of a group are feelers of ONE hit point (tries a multiple of 16).
// MULTI = true:  tries is 4 or 8 and a group holds the feelers of 16 / tries neighbouring hit points.  Such a
//                group is walked only if its hit points lie close together (RTX_COOP_WIDE); the others (a ground
//                plane seen near the horizon, a silhouette) go on a list (DShadowQ.wide) for the per-lane kernel that
//                runs next; every queue entry this kernel has finished is also marked empty.
#ifndef RTX_COOP_WIDE
#define RTX_COOP_WIDE 0.02f   // largest spread of a group's hit points, as a fraction of the scene's radius
#endif
// DEFER (always with MULTI): groups with a wide beam go on the list sq.wide for the per-lane kernel launched next (see `wide` below).
template <bool MULTI, bool DEFER>
__global__ void __launch_bounds__(RTX_BLOCK, MULTI ? RTX_COOP_WAVES_MULTI : RTX_COOP_WAVES) k_shadow_coop(DScene sc, DFrame fr, DShadowQ sq, uint32_t light,
                                                                            uint32_t n_groups, uint32_t tries, uint32_t pow2_tries,
                                                                            DCounters *counters, double *acc) {
    // group-shared LDS, written and read by lanes of ONE wave: the LDS executes a wave's operations in
    // order, COOP_FENCE() keeps the compiler from moving them across each other (a `volatile` array would
    // turn them into flat, cache-bypassing accesses)
    __shared__ uint32_t s_stack[RTX_BLOCK / 16][RTX_COOP_STACK];
    __shared__ uint32_t s_queue[RTX_BLOCK / 16][RTX_COOP_QUEUE];
    // wave-shared: (primitive ref, owner lane | surely_in << 6) pairs that passed the owner's f32 box test and
    // wait for the exact f64 tests; s_hit[lane] != 0 once an exact test has found an occluder for that lane's ray
    __shared__ uint32_t s_pair[RTX_BLOCK / 64][RTX_COOP_PAIRS];
    __shared__ uint32_t s_hit[RTX_BLOCK];
    // every lane's feeler (origin, direction, inverse direction), read back by the lane that runs its exact tests
    __shared__ double s_ray[3][RTX_BLOCK];      // direction per lane (the inverse is recomputed where an exact entry-bbox test needs it)
    // every lane's f32 traversal constants, read by whichever lanes test a queued primitive's box against this lane's ray:
    // (ix, iy, iz, tmax_own | cnx, cny, cnz, cfx) + (cfy, cfz)
    __shared__ float4 s_rayf[2][RTX_BLOCK];
    __shared__ float2 s_rayg[RTX_BLOCK];
    __shared__ double s_org[3][MULTI ? RTX_BLOCK : RTX_BLOCK / 16];   // origin: per group (one hit point), or per lane
    // each group's beam (11 floats), re-read at every node step: they would otherwise sit in registers during the
    // exact tests, where the pressure peaks, and be spilled to scratch memory
    __shared__ float s_beam[RTX_BLOCK / 16][12];
#define COOP_FENCE() asm volatile("" ::: "memory")
#ifdef RTX_COOP_LDS_PAD
    __shared__ uint32_t s_pad[RTX_COOP_LDS_PAD];   // experiment: fewer workgroups per CU, so that another kernel's workgroups fit next to them
    if (n_groups == 0xFFFFFFFFu) s_pad[threadIdx.x] = tries;
#endif
    uint64_t gid = (uint64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    // (a 64-bit division by a run-time value is ~40 instructions; the usual try counts are powers of two)
    const uint32_t tries_shift = (uint32_t)__ffs((int)tries) - 1u;
    uint32_t group = pow2_tries ? (uint32_t)(gid >> tries_shift) : (uint32_t)(gid / tries);
    uint32_t k = pow2_tries ? ((uint32_t)gid & (tries - 1u)) : (uint32_t)(gid % tries);
    // the queue is compact: *sq.count groups (<= n_groups, the rays of the step).  The count is fetched together with the group's
    // entry (slots past the count hold stale entries of earlier steps: read, then ignored), not before it
    const uint32_t n_act = *sq.count;
    bool valid = group < n_groups;
    size_t qc = sq.cap;
    V3 hp = v3(0, 0, 0);
    uint64_t key = 0;
    uint32_t sample = 0;
    if (valid) sample = sq.sample[(size_t)light * qc + group];
    if ((uint64_t)blockIdx.x * RTX_BLOCK >= (uint64_t)n_act * tries) return;   // (the whole workgroup)
    valid = valid && group < n_act && sample != SHADOW_GROUP_EMPTY;
    if (valid) {
        const double *f = sq.f + (size_t)light * 6 * qc + group;
        hp = v3(f[0 * qc], f[1 * qc], f[2 * qc]);
        key = sq.key[(size_t)light * qc + group];
    }

    // the lane's own feeler, exactly as k_shadow builds it (renderer.rs:198-206), but for the inverse direction: this kernel needs
    // 1 / d only for its conservative f32 tests, and takes it from v_rcp_f32 (1 ulp) instead of three f64 divisions.  The exact
    // tests build their own ray (run_pairs, confirm pass); a feeler that starts outside the scene's cube (rare) pays for the f64
    // inverse in root_entry.
    Ray r;
    r.ox = r.oy = r.oz = 0.0; r.dx = r.dy = 0.0; r.dz = 1.0;
    double distance_to_light = 0.0;
    if (valid) {
        const rtx_light &lt = sc.lights[light];
        V3 sampled = v3(lt.position[0], lt.position[1], lt.position[2]);
        if (lt.kind != RTX_LIGHT_POINT) {                                                // spherelight.rs:13-23
            double ux, uy, uz;
            draw3(fr.rng_mode, key, STREAM_LIGHT, light, k, &ux, &uy, &uz);
            V3 jitter = v3(lt.radius * (ux - 0.5), lt.radius * (uy - 0.5), lt.radius * (uz - 0.5));
            sampled = sampled + jitter;
        }
        V3 shadow_l = unit(sampled - hp);
        r.ox = hp.x; r.oy = hp.y; r.oz = hp.z;
        r.dx = shadow_l.x; r.dy = shadow_l.y; r.dz = shadow_l.z;
        distance_to_light = len(sampled - hp);
    }
    r.ix = r.iy = r.iz = 0.0; r.flip = 0u;   // (not used below)

    // ---- own ray in f32 (as traverse_ww) for the per-lane primitive box tests
    RootEntry re;
    re.t_shift = 0.0;
    re.miss = false;
    {
        const double R = sc.radius * 1.000001;
        const double ocx = r.ox - sc.centre[0], ocy = r.oy - sc.centre[1], ocz = r.oz - sc.centre[2];
        if (!(fabs(ocx) <= R && fabs(ocy) <= R && fabs(ocz) <= R)) re = root_entry(sc, make_ray(v3(r.ox, r.oy, r.oz), v3(r.dx, r.dy, r.dz)), ocx, ocy, ocz);
    }
    const double tsh = re.t_shift;   // f32 frame re-based to where the ray enters the scene's cube
    const double orx = r.ox - sc.centre[0] + tsh * r.dx, ory = r.oy - sc.centre[1] + tsh * r.dy, orz = r.oz - sc.centre[2] + tsh * r.dz;
    // Ray.signs (ray.rs:35-39: 1 / d > 0) = d > 0 for every feeler that walks (|d| >= 1e-9, degenerate_dir)
    const bool sx = r.dx > 0.0, sy = r.dy > 0.0, sz = r.dz > 0.0;
    r.sx = sx; r.sy = sy; r.sz = sz;
    // relative error of ix: 2^-24 (d to f32) + 2^-23 (v_rcp_f32) < 2^-22.4
    const float ix = __builtin_amdgcn_rcpf((float)r.dx), iy = __builtin_amdgcn_rcpf((float)r.dy), iz = __builtin_amdgcn_rcpf((float)r.dz);
    float cnx, cfx, cny, cfy, cnz, cfz;
    {
        // slab distance = fma(plane, ix, c), c = -(o * ix) with the SAME ix: (plane - o) * ix, off by |plane - o| * |1/d| * 2^-22.4 <=
        // big * |ix| * 2^-21.4; with the rounding of c (2^-24) and of the FMA (2^-23): 0.54 * 2^-20 * big * |ix|, inside the margin h
        const double big = fmax(fmax(fabs(orx), fabs(ory)), fabs(orz)) + sc.radius;
        const double delta = big * (1.0 / 1048576.0);
        const float hx = f32_above(delta * fabs((double)ix)), hy = f32_above(delta * fabs((double)iy)), hz = f32_above(delta * fabs((double)iz));
        const float cx = (float)(-(orx * (double)ix)), cy = (float)(-(ory * (double)iy)), cz = (float)(-(orz * (double)iz));
        cnx = cx - hx; cfx = cx + hx; cny = cy - hy; cfy = cy + hy; cnz = cz - hz; cfz = cz + hz;
    }
    const float tmax_own = f32_above(distance_to_light - tsh);
    // a feeler that misses the scene's cube (or ends before reaching it) has no finite candidate: it takes no
    // part in the beam and in the per-ray tests
    const bool degen = valid && degenerate_dir(r.dx, r.dy, r.dz);   // answered by k_shadow_retry; counts as 0 here
    const bool walk = valid && !degen && !re.miss && distance_to_light >= tsh;

    // ---- the group's beam, as a frustum along its dominant axis a0 (the axis with the largest
    // |direction| that every ray of the group travels in the same sense).  With u = distance along
    // a0 from the origin, ray k is at  o_b + u * m_b  on each other axis b, m_b = d_b / d_a0; the
    // beam is  [obmin + u * mL_b, obmax + u * mH_b]  for u >= 0.  A box is kept if its u range
    // (clipped to [0, umax]) is non-empty and on both other axes the beam's extent over that u
    // range overlaps the box: exact per axis pair, conservative overall.
    const float INF = __int_as_float(0x7F800000);
    const uint32_t lane = threadIdx.x & 63u, l16 = lane & 15u, gbase = lane & 48u, sub = threadIdx.x >> 4;
    const double dd[3] = {r.dx, r.dy, r.dz}, oo[3] = {orx, ory, orz};
    // smallest |d_a| over the group per axis (0 if the sign is mixed): the dominant axis maximises it
    float dmin[3];
    bool gneg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo, hi;
        grp_min_max(walk ? (float)dd[a] : INF, walk ? (float)dd[a] : -INF, &lo, &hi);
        dmin[a] = lo > 0.0f ? lo : (hi < 0.0f ? -hi : 0.0f);
        gneg[a] = hi < 0.0f;
    }
    int a0 = dmin[0] >= dmin[1] ? (dmin[0] >= dmin[2] ? 0 : 2) : (dmin[1] >= dmin[2] ? 1 : 2);
#ifdef RTX_COOP_NOBEAM
    const bool beam_ok = false;                     // debugging: keep every box
#else
    const bool beam_ok = dmin[a0] > 1.0e-3f;       // else (mixed directions): every box is kept
#endif
    const int a1 = a0 == 0 ? 1 : 0, a2 = a0 == 2 ? 1 : 2;
    const double d0 = a0 == 0 ? dd[0] : (a0 == 1 ? dd[1] : dd[2]);
    const double o0 = a0 == 0 ? oo[0] : (a0 == 1 ? oo[1] : oo[2]);
    const double d1 = a1 == 0 ? dd[0] : dd[1], o1 = a1 == 0 ? oo[0] : oo[1];
    const double d2 = a2 == 1 ? dd[1] : dd[2], o2 = a2 == 1 ? oo[1] : oo[2];
    // the sense of travel along a0, from the group reduction: lanes without a ray of their own
    // (tail of the launch) still test boxes for the group and must agree with it
    const bool neg0 = a0 == 0 ? gneg[0] : (a0 == 1 ? gneg[1] : gneg[2]);
    // u = s0 * (coord - origin), s0 = +-1; origin interval along a0, widened below
    float o0min, o0max, o1min, o1max, o2min, o2max, m1L, m1H, m2L, m2H;
    // One hit point per group and no feeler re-based to its entry into the scene's cube (always, but for hit points on planes outside it):
    // the 16 origins of a group are one point and need no reduction (a scalar branch; a group without a walker then holds a finite beam
    // instead of an empty one, and does not walk either way)
    if (!MULTI && __all(tsh == 0.0)) {
        o0min = f32_below(o0); o0max = f32_above(o0);
        o1min = f32_below(o1); o1max = f32_above(o1);
        o2min = f32_below(o2); o2max = f32_above(o2);
    } else {
        grp_min_max(walk ? f32_below(o0) : INF, walk ? f32_above(o0) : -INF, &o0min, &o0max);
        grp_min_max(walk ? f32_below(o1) : INF, walk ? f32_above(o1) : -INF, &o1min, &o1max);
        grp_min_max(walk ? f32_below(o2) : INF, walk ? f32_above(o2) : -INF, &o2min, &o2max);
    }
    // A group whose origins are spread out (MULTI: its hit points; one hit point: a point OUTSIDE the scene's cube, on a plane that
    // runs to the horizon, whose feelers enter the cube far apart: the origins above are the entry points) or whose feelers point in
    // mixed directions makes a beam that touches much of the hierarchy: it is left to the per-lane kernel that runs next, through the
    // list sq.wide.  (Without this a bunny on a ground plane with 16 tries spent 10.7 ms per frame in this kernel for the feelers of
    // its floor, the per-lane kernel 0.4 ms for all of them.)  MULTI always; one hit point per group: the DEFER instance, which the
    // host launches for scenes with planes while frames keep finding such groups (rtx_api.cpp: coop_defer), with a power of two of
    // tries (a queue group then lies within one wave; its rows must take the same decision).  Uniform over the queue group.
    static_assert(DEFER || !MULTI, "k_shadow_coop<MULTI> always defers");
    bool wide = false;
    constexpr bool can_defer = DEFER;
    if constexpr (can_defer) {
        const float spread = fmaxf(fmaxf(o0max - o0min, o1max - o1min), o2max - o2min);   // NaN / -inf (no walker): not wide
        wide = !beam_ok || spread > RTX_COOP_WIDE * (float)sc.radius;
        if (!MULTI && tries > 16u) {   // the rows of one queue group agree
            const unsigned long long wm = __ballot(wide);
            const unsigned long long mine = tries >= 64u ? ~0ull : (((1ull << tries) - 1ull) << (lane & ~(tries - 1u)));
            wide = (wm & mine) != 0ull;
        }
    }
    if (degen && !wide) retry_append(sq, light, (uint32_t)gid, sample, counters);   // (a wide group is left to the per-lane kernel, which appends)
    if constexpr (can_defer) {   // the queue groups of a wide unit go on the list of the per-lane kernel that runs next: one atomic per wave
        uint32_t *wl = sq.wide + (size_t)light * (1u + qc);
        const bool app = valid && wide && k == 0u;
        const unsigned long long am = __ballot(app);
        if (am != 0ull) {
            const int leader = __ffsll((long long)am) - 1;
            uint32_t base = 0u;
            if ((int)lane == leader) {
                base = atomicAdd(wl, (uint32_t)__popcll(am));
                if (!MULTI) atomicAdd(&counters->coop_deferred, (uint32_t)__popcll(am));   // (read by the host after the frame: is the DEFER instance earning its keep?)
            }
            base = (uint32_t)__shfl((int)base, leader);
            if (app) wl[1u + base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = group;
        }
    }
    // 1 / |d0| from the f32 inverse direction (relative error < 2^-22.4): the slopes are widened by 4e-7 of their size, which
    // covers it, their rounding to f32 and the widening's own rounding
    const double inv0 = (double)fabsf(a0 == 0 ? ix : (a0 == 1 ? iy : iz));
    const float m1f = (float)(d1 * inv0), m2f = (float)(d2 * inv0);
    const float m1w = fabsf(m1f) * 4.0e-7f, m2w = fabsf(m2f) * 4.0e-7f;
    grp_min_max(walk ? m1f - m1w : INF, walk ? m1f + m1w : -INF, &m1L, &m1H);
    grp_min_max(walk ? m2f - m2w : INF, walk ? m2f + m2w : -INF, &m2L, &m2H);
    const float umax = grp_max(walk ? f32_above((distance_to_light - tsh) * fabs(d0)) : 0.0f);
    {
        float big = fmaxf(fmaxf(fmaxf(fabsf(o0min), fabsf(o0max)), fmaxf(fabsf(o1min), fabsf(o1max))), fmaxf(fabsf(o2min), fabsf(o2max))) +
                    (float)sc.radius;
        float delta = big * (1.0f / 1048576.0f) * 1.0001f;
        // u <= 2*big, so a product u*m carries at most 2*big*|m|*2^-24 of rounding, and the rounding of
        // u itself (<= big*2^-23) is scaled by |m|: fold both into the origin intervals, 4x over
        float mmax = fmaxf(fmaxf(fabsf(m1L), fabsf(m1H)), fmaxf(fabsf(m2L), fabsf(m2H)));
        float slack = delta + big * mmax * 1.0e-6f;
        o0min -= delta; o0max += delta;
        o1min -= slack; o1max += slack;
        o2min -= slack; o2max += slack;
    }
    float umax_w = umax * 1.000001f + (o0max - o0min);
    if (!beam_ok) {   // mixed directions: an all-embracing beam keeps every non-empty box (an empty one is (+inf, -inf): NaN, dropped)
        o0min = o1min = o2min = -INF;
        o0max = o1max = o2max = INF;
        m1L = m1H = m2L = m2H = 0.0f;
        umax_w = 1.0f;
    }
    const bool pos1L = m1L >= 0.0f, pos1H = m1H >= 0.0f, pos2L = m2L >= 0.0f, pos2H = m2H >= 0.0f;
    if (l16 == 0u) {
        float *bm = s_beam[sub];
        bm[0] = o0min; bm[1] = o0max; bm[2] = o1min; bm[3] = o1max; bm[4] = o2min; bm[5] = o2max;
        bm[6] = m1L; bm[7] = m1H; bm[8] = m2L; bm[9] = m2H; bm[10] = umax_w;
    }
    COOP_FENCE();
#define COOP_STK(i) s_stack[sub][(i)]   /* spilled levels: 2 words each (first node, kind16 << 16 | pending16) */
#define COOP_QUE(i) s_queue[sub][(i)]
    const uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t below = (1u << l16) - 1u;
    bool alive = valid && !degen;     // the feeler has met no occluder yet
    bool seek = walk && !wide;   // ... and can still meet a finite one (a wide group does not walk at all)
    uint32_t a16 = (uint32_t)(__ballot(seek) >> gbase) & 0xFFFFu;
    // Group state, replicated in every lane of the group (it is computed from ballots only).
    // A LEVEL is a visited node with node-children still to visit: (tfirst, tmask) is the innermost
    // one, older ones are spilled to LDS.  Two nodes are tested per step (nodeA, nodeB).
#ifdef RTX_ABL_NOWALK
    uint32_t nodeA = NONE, nodeB = NONE;   // ablation: prologue and epilogue only
#else
    uint32_t nodeA = a16 ? 0u : NONE, nodeB = NONE;
#endif
    uint32_t tfirst = 0u, tmask = 0u, sp = 0u, qn = 0u;
    WORK(uint32_t w_nodes = 0; uint32_t w_entries = 0; uint32_t w_prims = 0;)

    // beam-vs-box test of one child record (see the comment above the beam set-up): c0 = the box along the
    // group's dominant axis, c1 / c2 = along the other two
    auto beam_test = [&](float c0l, float c0h, float c1l, float c1h, float c2l, float c2h) -> bool {
        const float *bm = s_beam[sub];
        const float o0min = bm[0], o0max = bm[1], o1min = bm[2], o1max = bm[3], o2min = bm[4], o2max = bm[5];
        const float m1L = bm[6], m1H = bm[7], m2L = bm[8], m2H = bm[9], umax_w = bm[10];
        float u0 = neg0 ? (o0min - c0h) : (c0l - o0max);      // smallest u at which any ray can be inside
        float u1 = neg0 ? (o0max - c0l) : (c0h - o0min);      // largest
        u0 = fmaxf(u0, 0.0f);
        u1 = fminf(u1, umax_w);
        // min(u0 * m, u1 * m) over u0 <= u1 is taken at u0 for m >= 0 and at u1 otherwise (max: the other way round); the signs of
        // the four slopes are group constants (pos1L ..), so each bound is one select and one FMA (rounded once: closer to the exact
        // bound than the product and sum the slack was sized for)
        const float b1lo = fmaf(pos1L ? u0 : u1, m1L, o1min), b1hi = fmaf(pos1H ? u1 : u0, m1H, o1max);
        const float b2lo = fmaf(pos2L ? u0 : u1, m2L, o2min), b2hi = fmaf(pos2H ? u1 : u0, m2H, o2max);
        return u0 <= u1 && b1lo <= c1h && b1hi >= c1l && b2lo <= c2h && b2hi >= c2l;
    };
    // record = (lo.x, lo.y, lo.z, hi.x | hi.y, hi.z, ref, first_node); AX = dominant axis known at compile time,
    // -1 = the lane's own a0 (per-lane selects)
    auto keep_box = [&](auto AX, const float4 &qa, const float4 &qb) -> bool {
        constexpr int ax = decltype(AX)::value;
        if constexpr (ax == 0) return beam_test(qa.x, qa.w, qa.y, qb.x, qa.z, qb.y);
        else if constexpr (ax == 1) return beam_test(qa.y, qb.x, qa.x, qa.w, qa.z, qb.y);
        else if constexpr (ax == 2) return beam_test(qa.z, qb.y, qa.x, qa.w, qa.y, qb.x);
        else
            return beam_test(a0 == 0 ? qa.x : (a0 == 1 ? qa.y : qa.z), a0 == 0 ? qa.w : (a0 == 1 ? qb.x : qb.y),
                             a1 == 0 ? qa.x : qa.y, a1 == 0 ? qa.w : qb.x, a2 == 1 ? qa.y : qa.z, a2 == 1 ? qb.x : qb.y);
    };
    // the four groups of a wave are neighbouring hit points looking at the same light: their dominant axis is
    // almost always the same, and then a scalar branch picks a beam test without per-lane component selects
    const int a0_first = __builtin_amdgcn_readfirstlane(a0);
    const int a0_wave = __all(a0 == a0_first) ? a0_first : -1;
    // the next node of the group's depth-first walk (NONE when the walk is over)
    auto take_node = [&]() -> uint32_t {
        if ((tmask & 0xFFFFu) == 0u && sp > 0u) {   // innermost level exhausted: back to the spilled one (never empty)
            --sp;
            COOP_FENCE();
            tfirst = COOP_STK(2u * sp);
            tmask = COOP_STK(2u * sp + 1u);
        }
        const uint32_t pend = tmask & 0xFFFFu;
        if (pend == 0u) return NONE;
        const uint32_t slot = (uint32_t)__ffs((int)pend) - 1u;
        tmask &= ~(1u << slot);
        return tfirst + (uint32_t)__popc((tmask >> 16) & ((1u << slot) - 1u));
    };

    // Exact tests of up to 64 waiting pairs, one per lane (octree.rs:170 + renderer.rs:212 on the owner's ray,
    // read back from LDS).  Must be called by all 64 lanes.
    const uint32_t wid = threadIdx.x >> 6;
    uint32_t pair_n = 0u;           // wave-uniform
    s_hit[threadIdx.x] = (walk && !wide) ? 0u : 2u;   // bit 0: an occluder was found; bit 1: the lane's ray does not take part in the tests
    if (MULTI) { s_org[0][threadIdx.x] = r.ox; s_org[1][threadIdx.x] = r.oy; s_org[2][threadIdx.x] = r.oz; }
    else if (l16 == 0u) { s_org[0][sub] = r.ox; s_org[1][sub] = r.oy; s_org[2][sub] = r.oz; }
    s_ray[0][threadIdx.x] = r.dx; s_ray[1][threadIdx.x] = r.dy; s_ray[2][threadIdx.x] = r.dz;
    s_rayf[0][threadIdx.x] = make_float4(ix, iy, iz, tmax_own);
    s_rayf[1][threadIdx.x] = make_float4(cnx, cny, cnz, cfx);
    s_rayg[threadIdx.x] = make_float2(cfy, cfz);
    COOP_FENCE();
#ifdef RTX_COUNT_LOOPS
    uint32_t c_steps = 0, c_drain = 0, c_pairs = 0, c_pairlanes = 0, c_gsteps = 0, c_devents = 0;
#endif
    auto run_pairs = [&]() {
        const uint32_t n = pair_n < 64u ? pair_n : 64u;
#ifdef RTX_COUNT_LOOPS
        ++c_pairs; c_pairlanes += n;
#endif
        pair_n -= n;
        const bool mine = lane < n;
        const uint32_t pw = s_pair[wid][pair_n + (mine ? lane : 0u)];
        const uint2 pr = make_uint2((pw & RTX_WREF_KIND) | (pw & RTX_PAIR_INDEX), (pw >> 23) & 127u);
        const uint32_t owner = (threadIdx.x & ~63u) + (pr.y & 63u);
        const double q_dist = __shfl(distance_to_light, (int)(pr.y & 63u));
        if (mine && s_hit[owner] == 0u) {
            Ray q;
            const uint32_t oi = MULTI ? owner : (owner >> 4);
            q.ox = s_org[0][oi]; q.oy = s_org[1][oi]; q.oz = s_org[2][oi];
            q.dx = s_ray[0][owner]; q.dy = s_ray[1][owner]; q.dz = s_ray[2][owner];
            q.ix = q.iy = q.iz = 0.0; q.sx = q.sy = q.sz = true; q.flip = 0u;   // read by the entry-bbox tests only
            // the inverse direction is fetched only where the exact entry-bbox test runs (keeps the primitive
            // tests' register peak down)
            auto entry_ray = [&]() {
                Ray e = q;
                e.ix = 1.0 / q.dx; e.iy = 1.0 / q.dy; e.iz = 1.0 / q.dz;   // as make_ray (ray.rs:29-33)
                e.sx = e.ix > 0.0; e.sy = e.iy > 0.0; e.sz = e.iz > 0.0;
                return e;
            };
            const bool surely_in = (pr.y & 64u) != 0u;
            const uint32_t idx = pr.x & RTX_WREF_INDEX;
            bool occluded = false;
#if RTX_COOP_DEFER_ENTRY
            // DEFERRED ENTRY-BBOX TEST.  A candidate counts iff its own bbox passes (octree.rs:170) AND the primitive test passes
            // (renderer.rs:212): two pure predicates, so their order is free.  Every pair here has passed the widened f32 box
            // test, which the f64 test fails only within delta of the box: run in place it costs as much as the primitive test
            // (three divisions for the inverse direction, min/max of the vertices, the slab test) and practically never
            // rejects.  So the primitive test runs alone; a hit whose box was not "surely in" is TENTATIVE: it stops the
            // feeler like a hit and records its primitive, and the feeler's own lane runs the f64 bbox test once, after the
            // walk (confirm pass below).  A hit that fails it (seen only in the forced-failure test build) does not count, and
            // the feeler is answered by the all-f64 walk through the launch's retry list, like a degenerate one.
            (void)entry_ray;
            uint32_t kind_bit = 0u;
            if ((pr.x & RTX_WREF_KIND) == RTX_WREF_TRI) {
                int pid, mat;
                TriV tr = load_tri32(sc.tris, idx, &pid, &mat);   // (idx < 2^23)
                WORK(++w_prims;)
                double t, be, ga;
                occluded = tri_hit(tr, q, EPSILON, q_dist, &t, &be, &ga);
#ifdef RTX_COUNT_CLASS
                {   // experiment: how the exact tests end (0 hit; 1 barycentrics out by > 1e-3; 2 out by less; 3 |t| < 1e-9; 4 behind; 5 beyond the light)
                    double t2, b2, g2;
                    int cls;
                    if (occluded) cls = 0;
                    else {
                        const bool any = tri_hit(tr, q, -inf(), inf(), &t2, &b2, &g2);
                        if (!any) {
                            V3 dd = v3(q.dx, q.dy, q.dz), e1 = tr.v1 - tr.v0, e2 = tr.v2 - tr.v0, pp = cross(dd, e2), ss = v3(q.ox, q.oy, q.oz) - tr.v0;
                            const double det = dot(e1, pp), idt = 1.0 / det, bb = idt * dot(ss, pp), gg = idt * dot(dd, cross(ss, e1));
                            const double out = fmax(fmax(-bb, -gg), bb + gg - 1.0);
                            cls = out > 1.0e-3 ? 1 : 2;
                        } else cls = fabs(t2) < 1.0e-9 ? 3 : (t2 < 0.0 ? 4 : 5);
                    }
                    atomicAdd(&counters->work[cls / 3][cls % 3], 1ull);
                }
#endif
            } else {
                const DSphere &sp_ = sc.spheres[idx];
                WORK(++w_prims;)
                double t;
                occluded = sphere_hit(v3(sp_.c[0], sp_.c[1], sp_.c[2]), sp_.r, q, EPSILON, q_dist, &t);
                kind_bit = 8u;
            }
            // bit 0: confirmed occluder; bit 2: tentative, bit 3: sphere, bits 4..: primitive index (< 2^23).  Several pairs
            // of one feeler may write in the same round: any of them will do (a failed confirmation goes to the exact walk)
            if (occluded) s_hit[owner] = surely_in ? 1u : (4u | kind_bit | (idx << 4));
#else
            if ((pr.x & RTX_WREF_KIND) == RTX_WREF_TRI) {
                int pid, mat;
                TriV tr = load_tri(sc.tris, (int)idx, &pid, &mat);
                if (surely_in || tri_bbox_hit(tr, entry_ray())) {
                    WORK(++w_prims;)
                    double t, be, ga;
                    occluded = tri_hit(tr, q, EPSILON, q_dist, &t, &be, &ga);
                }
            } else {
                const DSphere &sp_ = sc.spheres[idx];
                V3 cc = v3(sp_.c[0], sp_.c[1], sp_.c[2]);
                if (surely_in || sphere_bbox_hit(cc, sp_.r, entry_ray())) {
                    WORK(++w_prims;)
                    double t;
                    occluded = sphere_hit(cc, sp_.r, q, EPSILON, q_dist, &t);
                }
            }
            if (occluded) s_hit[owner] = 1u;
#endif
        }
        COOP_FENCE();
        if (s_hit[threadIdx.x] & 5u) { alive = false; seek = false; }
    };

    for (;;) {
        const bool g_on = nodeA != NONE;
        if (!__any(g_on)) break;
#ifdef RTX_COUNT_LOOPS
        ++c_steps;
        c_gsteps += (uint32_t)__popcll(__ballot(g_on)) >> 4;
#endif
        if (g_on) {
            const bool hasB = nodeB != NONE;
            // (32-bit byte offsets from the array's base, which sits in scalar registers: one add per record instead of a 64-bit shift and
            // add on a per-lane pointer that the allocator kept in scratch; the host keeps scenes whose box lists pass 4 GB off this kernel)
            const char *wbytes = reinterpret_cast<const char *>(sc.wide);
            const uint32_t l16off = l16 * (uint32_t)sizeof(DWideChild);
            const float4 *pa = reinterpret_cast<const float4 *>(wbytes + (nodeA * (uint32_t)(RTX_WIDE * sizeof(DWideChild)) + l16off));
            const float4 *pb = reinterpret_cast<const float4 *>(wbytes + ((hasB ? nodeB : nodeA) * (uint32_t)(RTX_WIDE * sizeof(DWideChild)) + l16off));
            const float4 a_lo = pa[0], a_hi = pa[1], b_lo = pb[0], b_hi = pb[1];
            WORK(w_nodes += hasB ? 2u : 1u;)
            const uint32_t refA = __float_as_uint(a_hi.z), refB = __float_as_uint(b_hi.z);
            const uint32_t kindA = refA & RTX_WREF_KIND, kindB = refB & RTX_WREF_KIND;
            bool hitA, hitB;                                    // an empty child's box is (+inf, -inf): never kept
            if (a0_wave == 0) {
                hitA = keep_box(std::integral_constant<int, 0>(), a_lo, a_hi);
                hitB = hasB && keep_box(std::integral_constant<int, 0>(), b_lo, b_hi);
            } else if (a0_wave == 1) {
                hitA = keep_box(std::integral_constant<int, 1>(), a_lo, a_hi);
                hitB = hasB && keep_box(std::integral_constant<int, 1>(), b_lo, b_hi);
            } else if (a0_wave == 2) {
                hitA = keep_box(std::integral_constant<int, 2>(), a_lo, a_hi);
                hitB = hasB && keep_box(std::integral_constant<int, 2>(), b_lo, b_hi);
            } else {
                hitA = keep_box(std::integral_constant<int, -1>(), a_lo, a_hi);
                hitB = hasB && keep_box(std::integral_constant<int, -1>(), b_lo, b_hi);
            }
            const bool isnA = kindA == RTX_WREF_NODE, isnB = kindB == RTX_WREF_NODE;
            const uint32_t hA = (uint32_t)(__ballot(hitA) >> gbase) & 0xFFFFu, kA = (uint32_t)(__ballot(isnA) >> gbase) & 0xFFFFu;
            const uint32_t hB = (uint32_t)(__ballot(hitB) >> gbase) & 0xFFFFu, kB = (uint32_t)(__ballot(isnB) >> gbase) & 0xFFFFu;
            const uint32_t nA = hA & kA, pA = hA & ~kA, nB = hB & kB, pB = hB & ~kB;
            // primitives the beam touches: queued for the per-ray tests
            if (hitA && !isnA) COOP_QUE(qn + __popc(pA & below)) = (nodeA << 4) | l16;
            qn += __popc(pA);
            if (hitB && !isnB) COOP_QUE(qn + __popc(pB & below)) = (nodeB << 4) | l16;
            qn += __popc(pB);
            // node children still to visit: a new innermost level per node (B's first, so that A's subtree is
            // walked first); the level it displaces is spilled if it still has children pending
            const bool spillB = nB != 0u && (tmask & 0xFFFFu) != 0u;
            if (spillB && l16 == 0u) { COOP_STK(2u * sp) = tfirst; COOP_STK(2u * sp + 1u) = tmask; }
            sp += spillB ? 1u : 0u;
            tfirst = nB ? __float_as_uint(b_hi.w) : tfirst;
            tmask = nB ? ((kB << 16) | nB) : tmask;
            const bool spillA = nA != 0u && (tmask & 0xFFFFu) != 0u;
            if (spillA && l16 == 0u) { COOP_STK(2u * sp) = tfirst; COOP_STK(2u * sp + 1u) = tmask; }
            sp += spillA ? 1u : 0u;
            tfirst = nA ? __float_as_uint(a_hi.w) : tfirst;
            tmask = nA ? ((kA << 16) | nA) : tmask;
            COOP_FENCE();
        }
        // ---- drain the candidate queues: every lane tests the primitive's box against its OWN ray in f32; the
        // (ray, primitive) pairs that survive are appended to the wave's pair list, and the exact f64 tests then
        // run on up to 64 pairs at a time, one pair per lane whoever owns the ray.  Tested in place, a primitive
        // would keep only the lanes whose ray passes its box busy (measured: 17 of 64 on average).
        const bool walk_over = (tmask & 0xFFFFu) == 0u && sp == 0u;
        const bool want_x = g_on && (qn >= RTX_COOP_QT || (walk_over && qn > 0u));
        if (__any(want_x)) {
#ifdef RTX_COUNT_LOOPS
            ++c_devents;
#endif
            {
                // All four queues of the wave are drained together and the (primitive, group) items are dealt out evenly: every
                // round, each 16-lane row takes ONE item and tests the primitive's box against the 16 rays of the item's group
                // (their f32 constants come from LDS).  Row-bound draining (each row its own queue) ran max(queue length)
                // rounds: measured 14.2 rounds per wave for 6.0 box tests per ray.
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)qn, 0), q1 = (uint32_t)__builtin_amdgcn_readlane((int)qn, 16);
                const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)qn, 32), q3 = (uint32_t)__builtin_amdgcn_readlane((int)qn, 48);
#ifdef RTX_ABL_NODRAIN
                const uint32_t p1 = q0, p2 = q0 + q1, p3 = p2 + q2, total = 0u * (p3 + q3);   // ablation: the walk alone
#else
                const uint32_t p1 = q0, p2 = q0 + q1, p3 = p2 + q2, total = p3 + q3;
#endif
                COOP_FENCE();
                for (uint32_t base = 0u; base < total; base += 4u) {
#ifdef RTX_COUNT_LOOPS
                    ++c_drain;
#endif
                    const uint32_t item = base + (lane >> 4);
                    bool push = false;
                    uint32_t pref = 0u, powner = lane;
                    if (item < total) {
                        const uint32_t g = item < p1 ? 0u : (item < p2 ? 1u : (item < p3 ? 2u : 3u));
                        const uint32_t qi = item - (g == 0u ? 0u : (g == 1u ? p1 : (g == 2u ? p2 : p3)));
                        const uint32_t c = s_queue[(wid << 2) + g][qi];
                        const uint32_t ol = (g << 4) | l16;               // the ray's lane in the wave
                        const uint32_t ot = (threadIdx.x & ~63u) + ol;    // ... and in the workgroup
                        if (s_hit[ot] == 0u) {
                            const float4 ra = s_rayf[0][ot], rn = s_rayf[1][ot];
                            const float2 rg = s_rayg[ot];
                            const float4 rf = make_float4(rn.w, rg.x, rg.y, 0.0f);
                            const float4 *pc = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.wide) + c * (uint32_t)sizeof(DWideChild));   // c = node * 16 + slot
                            const float4 c_lo = pc[0], c_hi = pc[1];
                            const float lx = c_lo.x, ly = c_lo.y, lz = c_lo.z, ux = c_lo.w, uy = c_hi.x, uz = c_hi.y;
                            WORK(++w_entries;)
                            const bool osx = ra.x > 0.0f, osy = ra.y > 0.0f, osz = ra.z > 0.0f;   // = Ray.signs: |1/d| >= 1 survives the rounding to f32
                            const float enx = osx ? lx : ux, efx = osx ? ux : lx;
                            const float eny = osy ? ly : uy, efy = osy ? uy : ly;
                            const float enz = osz ? lz : uz, efz = osz ? uz : lz;
                            const float lo = fmaxf(fmaxf(fmaxf(fmaf(enx, ra.x, rn.x), 0.0f), fmaf(eny, ra.y, rn.y)), fmaf(enz, ra.z, rn.z));
                            const float hi = fminf(fminf(fminf(fmaf(efx, ra.x, rf.x), ra.w), fmaf(efy, ra.y, rf.y)), fmaf(efz, ra.z, rf.z));
                            if (lo <= hi) {
                                // the box shrunk by delta instead of widened: a ray through it passes the exact f64
                                // entry test (octree.rs:170) with a margin far above f64 rounding
                                bool surely_in = false;
                                if (fabsf(ra.x) < 3.0e38f && fabsf(ra.y) < 3.0e38f && fabsf(ra.z) < 3.0e38f) {
                                    const float li = fmaxf(fmaxf(fmaf(enx, ra.x, rf.x), fmaf(eny, ra.y, rf.y)), fmaf(enz, ra.z, rf.z));
                                    const float hi2 = fminf(fminf(fmaf(efx, ra.x, rn.x), fmaf(efy, ra.y, rn.y)), fmaf(efz, ra.z, rn.z));
                                    surely_in = li <= hi2 && hi2 > 0.0f;
                                }
                                push = true;
                                pref = __float_as_uint(c_hi.z);
                                powner = ol | (surely_in ? 64u : 0u);
                            }
                        }
                    }
                    const unsigned long long pm = __ballot(push);
                    if (pm != 0ull) {
                        if (push) s_pair[wid][pair_n + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = pair_pack(pref, powner);
                        pair_n += (uint32_t)__popcll(pm);
                        COOP_FENCE();
                        if (pair_n >= 64u) run_pairs();
                    }
                }
                qn = 0u;
            }
            if (pair_n >= RTX_COOP_PT) run_pairs();
            a16 = (uint32_t)(__ballot(seek) >> gbase) & 0xFFFFu;
            if (!a16) { sp = 0u; tmask = 0u; }   // every feeler of the group is occluded
        }
        nodeA = NONE;
        nodeB = NONE;
        if (g_on) {
            nodeA = take_node();
            nodeB = take_node();
            if (nodeA == NONE) { nodeA = nodeB; nodeB = NONE; }
        }
    }

    while (pair_n > 0u) run_pairs();   // what the last drains left waiting (RTX_COOP_PT > 1 only)
#if RTX_COOP_DEFER_ENTRY
    {   // confirm pass: every feeler stopped by a tentative hit runs the f64 entry-bbox test of that primitive, once, on its own lane
        COOP_FENCE();
        const uint32_t hw = s_hit[threadIdx.x];
        if (hw & 4u) {
            const uint32_t oi = MULTI ? threadIdx.x : sub;
            const Ray e = make_ray(v3(s_org[0][oi], s_org[1][oi], s_org[2][oi]),
                                   v3(s_ray[0][threadIdx.x], s_ray[1][threadIdx.x], s_ray[2][threadIdx.x]));   // ray.rs:23-42
            const uint32_t idx = hw >> 4;
            bool ok;
            if (hw & 8u) {
                const DSphere &sp_ = sc.spheres[idx];
                ok = sphere_bbox_hit(v3(sp_.c[0], sp_.c[1], sp_.c[2]), sp_.r, e);
            } else {
                int pid, mat;
                const TriV tr = load_tri(sc.tris, (int)idx, &pid, &mat);
                ok = tri_bbox_hit(tr, e);
            }
#ifdef RTX_DEBUG_FAIL_CONFIRM
            if ((idx + threadIdx.x) % RTX_DEBUG_FAIL_CONFIRM == 0u) ok = false;   // test build: exercise the hand-off below
#endif
            if (!ok) {   // the recorded hit does not count: the feeler stays out of this kernel's sum and takes the all-f64 walk
                const uint64_t g_ = (uint64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
                retry_append(sq, light, (uint32_t)g_, sq.sample[(size_t)light * qc + (uint32_t)(g_ / tries)], counters);
            }
        }
    }
#endif
    // infinite prims (planes), octree.rs:163-165: a feeler that is still clear tests them in order
    if (alive && !wide) {
        Ray q;      // the lane's own feeler again (kept in LDS, not in registers, during the walk)
        const uint32_t oi = MULTI ? threadIdx.x : sub;
        q.ox = s_org[0][oi]; q.oy = s_org[1][oi]; q.oz = s_org[2][oi];
        q.dx = s_ray[0][threadIdx.x]; q.dy = s_ray[1][threadIdx.x]; q.dz = s_ray[2][threadIdx.x];
        for (int p = 0; p < sc.n_planes; ++p) {
            double t;
            if (plane_hit(sc.planes[p], q, EPSILON, distance_to_light, &t)) { alive = false; break; }
        }
    }
#ifdef RTX_COUNT_LOOPS
    if (lane == 0u) {   // per wave: loop steps, drain rounds, run_pairs calls; [0][..]: lanes busy in them, group-steps, steps with a drain
        atomicAdd(&counters->work[1][0], (unsigned long long)c_steps);
        atomicAdd(&counters->work[1][1], (unsigned long long)c_drain);
        atomicAdd(&counters->work[1][2], (unsigned long long)c_pairs);
        atomicAdd(&counters->work[0][0], (unsigned long long)c_pairlanes);
        atomicAdd(&counters->work[0][1], (unsigned long long)c_gsteps);
        atomicAdd(&counters->work[0][2], (unsigned long long)c_devents);
    }
#elif defined(RTX_COUNT_WORK)
    if (valid) {
        atomicAdd(&counters->work[1][0], (unsigned long long)w_nodes);
        atomicAdd(&counters->work[1][1], (unsigned long long)w_entries);
    }
    atomicAdd(&counters->work[1][2], (unsigned long long)w_prims);   // exact tests are run by whichever lane took the pair
#endif
    V3 v = (valid && alive) ? v3(1.0, 1.0, 1.0) : v3(0.0, 0.0, 0.0);
    double inv_tries = 1.0 / (double)tries;
    // the group's weight (light colour x material sample x path throughput) is fetched only now, by the lanes
    // that add to the accumulator: held in registers across the walk it would be spilled to scratch memory
    if (pow2_tries) {
        // the sum of the group's 0 / 1 answers is the number of its clear feelers: a ballot and a population count give the same
        // (exact) f64 value as the butterfly of additions
        const unsigned long long clear = __ballot(valid && alive);
        const unsigned long long mine = tries >= 64u ? ~0ull : (((1ull << tries) - 1ull) << (lane & ~(tries - 1u)));
        const double n_clear = (double)__popcll(clear & mine);
        v = v3(n_clear, n_clear, n_clear);
    }
    uint32_t t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));   // group / try index and loads afresh, not the prologue's values kept alive
    const uint64_t gid2 = (uint64_t)blockIdx.x * RTX_BLOCK + t2;
    if (pow2_tries && tries >= 4u) {
        // the three components of a group's result belong to its first three lanes: one store / atomic instruction per wave instead of three
        const uint32_t kk = (uint32_t)gid2 & (tries - 1u);
#ifdef RTX_COUNT_OCCL   // experiment (tools/exp/occl_count.py): groups with every feeler occluded / every feeler clear / mixed, in rtx_stats.work[0]
        if (valid && !wide && kk == 0u) atomicAdd(&counters->work[0][v.x == 0.0 ? 0 : (v.x == (double)tries ? 1 : 2)], 1ull);
#endif
        if (valid && !wide && kk < 3u) {
            const uint32_t g2 = (uint32_t)(gid2 >> tries_shift);
            if (sq.res) {   // deferred: k_accumulate multiplies with the weight and adds (shadow_result)
                sq.res[((size_t)light * 3 + kk) * qc + g2] = v.x * inv_tries;
            } else {
                const double wk = sq.f[(size_t)light * 6 * qc + (size_t)(3u + kk) * qc + g2];
                const double add = wk * (v.x * inv_tries);                                                  // renderer.rs:220,118
                // (a zero leaves the sum as it is: no atomic for a fully occluded group; a NaN or infinite weight times zero is a NaN, as in the reference, and is added)
                if (add != 0.0) unsafeAtomicAdd(acc + (size_t)sq.sample[(size_t)light * qc + g2] * 3 + kk, add);
            }
        }
        if (MULTI) {   // done: the per-lane pass that follows skips the group (after its three lanes have read the sample)
            __builtin_amdgcn_wave_barrier();
            if (valid && !wide && kk == 0u) sq.sample[(size_t)light * qc + (uint32_t)(gid2 >> tries_shift)] = SHADOW_GROUP_EMPTY;
        }
    } else
    if (valid && !wide && (!pow2_tries || ((uint32_t)gid2 & (tries - 1u)) == 0u)) {
        const uint32_t g2 = pow2_tries ? (uint32_t)(gid2 >> tries_shift) : (uint32_t)(gid2 / tries);
        const double *f = sq.f + (size_t)light * 6 * qc + g2;
        acc_add(acc, sq.sample[(size_t)light * qc + g2], v3(f[3 * qc], f[4 * qc], f[5 * qc]) * scale(v, inv_tries));   // renderer.rs:220,118
        if (MULTI) sq.sample[(size_t)light * qc + g2] = SHADOW_GROUP_EMPTY;   // done: the per-lane pass that follows skips it
    }
}

// The local lighting of a step's rays, added to their samples once every light's shadow launch has stored its result (shadow_result):
// light.color() * material.sample() * shadow, summed over the lights in their order (the fold of renderer.rs:114-119; the weight carries
// the path's throughput).  `lights`: bit li = light li's launch has deferred its results (the others have added theirs themselves).
__global__ void __launch_bounds__(RTX_BLOCK) k_accumulate(DShadowQ sq, uint32_t n_lights, uint32_t lights, double *acc) {
    const uint32_t g = blockIdx.x * RTX_BLOCK + threadIdx.x;
    const uint32_t n = *sq.count;
    if (blockIdx.x * RTX_BLOCK >= n) return;   // (the whole workgroup)
    const bool has = g < n;
    const size_t qc = sq.cap;
    V3 c = v3(0.0, 0.0, 0.0);
    uint32_t sample = 0u;
    if (has) {
        sample = sq.sample0[g];
        for (uint32_t li = 0; li < n_lights; ++li) {
            if (!((lights >> li) & 1u)) continue;
            const double *f = sq.f + (size_t)li * 6 * qc + g, *r = sq.res + (size_t)li * 3 * qc + g;
            c = c + v3(f[3 * qc], f[4 * qc], f[5 * qc]) * v3(r[0], r[qc], r[2 * qc]);
        }
    }
    acc_add_runs(acc, has, sample, c);
}

// K6 resolve: per-sample clamp + average (renderer.rs:95) and ColorRGBA::new_rgb_clamped
// (compositor/colorrgba.rs:68-76); row 0 = top (Surface::merge, surface.rs:83-94)
__global__ void __launch_bounds__(RTX_BLOCK) k_resolve(DFrame fr, const double *acc, uint8_t *rgba_out) {
    uint32_t gid = blockIdx.x * RTX_BLOCK + threadIdx.x;
    uint32_t n_pix = fr.n_blocks_local * fr.tiles_x * 64u;
    if (gid >= n_pix) return;
    uint32_t tile = gid >> 6, lane = gid & 63u;
    uint32_t px, row, sub;
    uint32_t s0 = (tile * fr.spp) * 64u + lane;
    if (!decode_sample(fr, s0, &px, &row, &sub)) return;
    V3 color = v3(0.0, 0.0, 0.0);
    for (uint32_t sb = 0; sb < fr.spp; ++sb) {
        const double *p = acc + (size_t)(s0 + sb * 64u) * 3;
        color = color + scale(v3(clamp01(p[0]), clamp01(p[1]), clamp01(p[2])), fr.inv_spp);
    }
    double c[3] = {color.x, color.y, color.z};
    uchar4 out;
    uint8_t q[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        double rv = round(c[ch] * 255.0);      // f64::round: half away from zero
        int iv = (rv != rv) ? 0 : (rv <= -2147483648.0 ? (int)0x80000000 : (rv >= 2147483647.0 ? 0x7FFFFFFF : (int)rv));
        iv = iv > 255 ? 255 : (iv < 0 ? 0 : iv);
        q[ch] = (uint8_t)iv;
    }
    out.x = q[0]; out.y = q[1]; out.z = q[2]; out.w = 255;
    *reinterpret_cast<uchar4 *>(rgba_out + ((size_t)out_row(fr, row) * fr.W + px) * 4) = out;
}

// Probe kernels behind rtx_nearest_hits / rtx_shadow_products
template <int MODE>
__global__ void __launch_bounds__(RTX_BLOCK) k_nearest(DScene sc, int64_t n, const double *origin, const double *direction,
                                                        int32_t *prim_id, double *t_out) {
    // (BVH: the node stack of traverse_bvh, then the level stack of traverse_exact for the rays handed over)
    constexpr bool BVH = MODE == MODE_CLOSEST_BVH || MODE == MODE_CLOSEST_BVH_DEALT;
    __shared__ uint32_t s_stk[BVH ? (RTX_BVH_STACK > RTX_MAX_LEVELS ? RTX_BVH_STACK : RTX_MAX_LEVELS) * RTX_BLOCK : RTX_LDS_WORDS];
    __shared__ uint32_t s_cand[MODE == MODE_CLOSEST ? RTX_CAND_WORDS : (BVH ? RTX_BVH_CAND * RTX_BLOCK : 1)];
    __shared__ __align__(8) uint32_t s_aux[MODE == MODE_CLOSEST_BVH_DEALT ? 4 * RTX_BLOCK : 2];
    if constexpr (!BVH) init_lds(s_stk);
    int64_t gid = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    // MODE_CLOSEST_BVH_DEALT: every lane of the wave takes part in the walk (its exact tests are dealt out over the wave's 64 lanes); in
    // the other modes lanes that leave here simply drop out of the wave-wide votes
    if (MODE != MODE_CLOSEST_BVH_DEALT && gid >= n) return;
    const bool act = gid < n;
    const int64_t g = act ? gid : 0;
    V3 o = v3(origin[g * 3], origin[g * 3 + 1], origin[g * 3 + 2]);
    V3 d = v3(direction[g * 3], direction[g * 3 + 1], direction[g * 3 + 2]);
    Ray r = make_ray(o, d);
    Hit hit;
    V3 unused;
    if constexpr (MODE == MODE_CLOSEST) {   // what k_closest + k_closest_retry do
        const bool degen = degenerate_dir(r.dx, r.dy, r.dz);
        bool tie;
        traverse_ww<WW_CLOSEST>(sc, r, !degen, s_stk, s_cand, T_MIN, inf(), &hit, nullptr, nullptr, &tie);
        if (degen || tie) traverse_exact<MODE_CLOSEST_REF>(sc, r, s_stk + threadIdx.x, T_MIN, inf(), &hit, &unused);
    } else if constexpr (MODE == MODE_CLOSEST_BVH) {   // what k_closest<MODE_CLOSEST_BVH> + k_closest_retry do
        const bool degen = degenerate_dir(r.dx, r.dy, r.dz);
        bool ovf;
        traverse_bvh<WW_CLOSEST>(sc, r, !degen, s_stk, s_cand, T_MIN, inf(), &hit, &ovf);
        if (degen || ovf) traverse_exact<MODE_CLOSEST_REF>(sc, r, s_stk + threadIdx.x, T_MIN, inf(), &hit, &unused);
    } else if constexpr (MODE == MODE_CLOSEST_BVH_DEALT) {   // what k_closest<MODE_CLOSEST_BVH_DEALT> + k_closest_retry do
        const bool degen = degenerate_dir(r.dx, r.dy, r.dz);
        bool ovf;
        traverse_bvh_dealt<WW_CLOSEST>(sc, r, act && !degen, s_stk, s_cand, s_aux, T_MIN, inf(), &hit, &ovf);
        if (act && (degen || ovf)) traverse_exact<MODE_CLOSEST_REF>(sc, r, s_stk + threadIdx.x, T_MIN, inf(), &hit, &unused);
        if (!act) return;
    } else {
        traverse<MODE>(sc, r, s_stk, T_MIN, inf(), &hit, &unused);
    }
    int pid = -1;
    if (hit.ref != REF_NONE) {
        if (hit.ref >= 0 && !(hit.ref & REF_PLANE)) pid = sc.tris[hit.ref].prim_id;
        else if (hit.ref >= 0) pid = sc.planes[hit.ref & ~REF_PLANE].prim_id;
        else pid = sc.spheres[~hit.ref].prim_id;
    }
    prim_id[gid] = pid;
    t_out[gid] = hit.ref != REF_NONE ? hit.t : inf();
}
template <int MODE>
__global__ void __launch_bounds__(RTX_BLOCK) k_shadow_products(DScene sc, int64_t n, const double *origin, const double *target,
                                                                double *out) {
    __shared__ uint32_t s_stk[MODE == MODE_PRODUCT_BVH ? (RTX_BVH_STACK > RTX_MAX_LEVELS ? RTX_BVH_STACK : RTX_MAX_LEVELS) * RTX_BLOCK : RTX_LDS_WORDS];
    __shared__ uint32_t s_cand[MODE == MODE_PRODUCT_FAST ? RTX_CAND_WORDS : (MODE == MODE_PRODUCT_BVH ? RTX_BVH_CAND * RTX_BLOCK : 1)];
    __shared__ uint32_t s_hits[MODE == MODE_PRODUCT_BVH ? RTX_BVH_HITS * RTX_BLOCK : 1];
    __shared__ uint32_t s_aux[MODE == MODE_PRODUCT_BVH ? RTX_BLOCK : 1];
    if constexpr (MODE != MODE_PRODUCT_BVH) init_lds(s_stk);
    int64_t gid = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (MODE != MODE_PRODUCT_FAST && MODE != MODE_PRODUCT_BVH && gid >= n) return;
    const bool act = gid < n;   // MODE_PRODUCT_FAST / _BVH: every lane of the wave takes part in the while-while walk
    const int64_t g = act ? gid : 0;
    V3 p = v3(origin[g * 3], origin[g * 3 + 1], origin[g * 3 + 2]);
    V3 s = v3(target[g * 3], target[g * 3 + 1], target[g * 3 + 2]);
    V3 l = unit(s - p);
    Ray r = make_ray(p, l);
    double dist = len(s - p);
    Hit hit;
    V3 v = v3(0, 0, 0);
    if constexpr (MODE == MODE_PRODUCT_BVH) {   // what k_shadow<MODE_PRODUCT_BVH> runs per feeler
        bool fb = false;
        traverse_bvh_dealt<WW_PRODUCT>(sc, r, act, s_stk, s_cand, s_aux, EPSILON, dist, &hit, &fb, s_hits, &v);
        if (act && fb) traverse_exact<MODE_PRODUCT>(sc, r, s_stk + threadIdx.x, EPSILON, dist, &hit, &v);
    } else if constexpr (MODE == MODE_PRODUCT_FAST) {   // what k_shadow<MODE_PRODUCT_FAST> runs per feeler
        bool fb = false;
        traverse_ww<WW_PRODUCT>(sc, r, act, s_stk, s_cand, EPSILON, dist, &hit, &v, &fb);
        if (act && fb) traverse_exact<MODE_PRODUCT>(sc, r, s_stk + threadIdx.x, EPSILON, dist, &hit, &v);
    } else {
        traverse<MODE>(sc, r, s_stk, EPSILON, dist, &hit, &v);
    }
    if (!act) return;
    out[gid * 3] = v.x; out[gid * 3 + 1] = v.y; out[gid * 3 + 2] = v.z;
}

// Probe behind rtx_shade_probe: the shading device functions of k_shade evaluated on caller-supplied arguments (the same functions,
// not copies), so that per-function golden vectors can be compared without going through a whole frame and its 8-bit quantisation.
__global__ void __launch_bounds__(RTX_BLOCK) k_shade_probe(DScene sc, int which, int64_t n, const double *in, double *out) {
    int64_t gid = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (gid >= n) return;
    const double *a = in + gid * 16;
    double *o = out + gid * 4;
    V3 r = v3(0, 0, 0);
    double w = 0.0;
    switch (which) {
    case RTX_PROBE_FRESNEL:    // ior, i, n, inside                       (renderer.rs:226-247)
        r.x = fresnel_reflect(a[0], v3(a[1], a[2], a[3]), v3(a[4], a[5], a[6]), a[7] != 0.0);
        break;
    case RTX_PROBE_REFRACT: {  // v, n, ior, inside -> direction, ok      (vec3.rs:77-92)
        V3 t = v3(0, 0, 0);
        w = refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6], a[7] != 0.0, &t) ? 1.0 : 0.0;
        r = t;
        break;
    }
    case RTX_PROBE_REFLECT:    // v, n                                    (vec3.rs:70-72)
        r = reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]));
        break;
    case RTX_PROBE_MATERIAL:   // material row, n, i, l, u, v             (Material::sample of the row's kind)
        r = material_sample(material_prepare(sc, sc.materials[(int)a[0]], a[10], a[11]), v3(a[1], a[2], a[3]), v3(a[4], a[5], a[6]), v3(a[7], a[8], a[9]));
        break;
    case RTX_PROBE_TEXTURE:    // texture row, u, v                       (Texture::color of the row's kind)
        r = texture_color(sc, (int)a[0], a[1], a[2]);
        break;
    case RTX_PROBE_SKYBOX:     // direction                               (cubemap.rs:45-84)
        r = cubemap_color(sc, v3(a[0], a[1], a[2]));
        break;
    case RTX_PROBE_BBOX: {     // bbox min, max, ray origin, direction -> intersects (bbox.rs:99-165)
        double tn, tf;
        r.x = slab(a[0], a[1], a[2], a[3], a[4], a[5], make_ray(v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11])), &tn, &tf) ? 1.0 : 0.0;
        break;
    }
    default:
        break;
    }
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = w;
}

// ================================================================================ launch wrappers
static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + RTX_BLOCK - 1) / RTX_BLOCK); }

void rtxk_generate(rtx_stream_t stream, const DFrame &fr, DPool pool, uint32_t pool_base, uint32_t s0, uint32_t count) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_generate, dim3(grid_for(count)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, fr, pool, pool_base, s0, count);
}
void rtxk_trace(rtx_stream_t stream, const DScene &sc, const DFrame &fr, DPool pool, uint32_t begin, uint32_t count,
                DShadowQ sq, DCounters *counters, double *acc, int32_t *prim_id_out, double *hit_t, int32_t *hit_ref,
                void *wait_before_shade, void *wait_before_shade2, void *record_after_closest, uint32_t *group_base, uint32_t child_begin,
                RtxTraceNames *names, const RtxSortBuf *sortbuf, bool sort_rays, uint32_t *host_step, uint32_t host_seq, uint32_t prim_s0) {
    if (count == 0) return;
    RtxTraceNames nm;
    static_assert(MODE_CLOSEST == 0 && MODE_CLOSEST_REF == 1 && MODE_CLOSEST_BVH == 5 && MODE_CLOSEST_BVH_DEALT == 8, "the kernel names below");
    nm.closest = (fr.flags & RTX_FLAG_REFERENCE_ORDER) ? "k_closest<1>"
                 : ((sc.bvh4 && !(fr.flags & RTX_FLAG_OCTREE_WALK)) ? (sc.all_opaque ? "k_closest<5>" : "k_closest<8>") : "k_closest<0>");
    nm.shade = (sc.shade_features || !(fr.flags & RTX_FLAG_INTERNAL_SMALL_SHADE)) ? "k_shade<true>" : "k_shade<false>";
    if (names) *names = nm;
    const bool want_groups = fr.shadow_samples != 0u && sc.n_lights > 0;   // the slots of the compact shadow queue
    const unsigned nb = grid_for(count);
    // group_base: four arrays of nb words: hits per workgroup, their exclusive scan, children per workgroup, their exclusive scan
    uint32_t *hit_count = want_groups ? group_base : nullptr, *hit_base = group_base + nb, *child_count = group_base + 2 * nb, *child_base = group_base + 3 * nb;
    const bool prim = prim_s0 != RTX_NOT_PRIMARY;
    const uint32_t *order = nullptr;
    if (sort_rays && sortbuf && pool.cell && sortbuf->ray_order) {
        rtxk_sort_rays(stream, pool, *sortbuf, begin, count);
        order = sortbuf->ray_order;
    }
    if (fr.flags & RTX_FLAG_REFERENCE_ORDER) {
        { if (prim) hipLaunchKernelGGL((k_closest<MODE_CLOSEST_REF, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                           count, counters, hit_t, hit_ref, order, fr, prim_s0);
        else hipLaunchKernelGGL((k_closest<MODE_CLOSEST_REF, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                           count, counters, hit_t, hit_ref, order, fr, prim_s0); }
        hipLaunchKernelGGL(k_group_count, dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin, count, hit_ref, fr.gloss_samples,
                           hit_count, child_count, prim_s0 != 0xFFFFFFFFu ? ((fr.reflect_depth << RAYF_REFLECT_SHIFT) | (fr.refract_depth << RAYF_REFRACT_SHIFT) | ((uint32_t)RAY_PRIMARY << RAYF_TYPE_SHIFT)) : 0u);
    } else
    {
        // (scenes with transmissive materials: rays run along the inside of thin shells and meet ten candidates each: the exact tests dealt
        // out over the wave, see traverse_bvh_dealt)
        if (sc.bvh4 && !(fr.flags & RTX_FLAG_OCTREE_WALK) && !sc.all_opaque)
            { if (prim) hipLaunchKernelGGL((k_closest<MODE_CLOSEST_BVH_DEALT, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                               count, counters, hit_t, hit_ref, order, fr, prim_s0);
        else hipLaunchKernelGGL((k_closest<MODE_CLOSEST_BVH_DEALT, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                               count, counters, hit_t, hit_ref, order, fr, prim_s0); }
        else if (sc.bvh4 && !(fr.flags & RTX_FLAG_OCTREE_WALK))
            { if (prim) hipLaunchKernelGGL((k_closest<MODE_CLOSEST_BVH, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                               count, counters, hit_t, hit_ref, order, fr, prim_s0);
        else hipLaunchKernelGGL((k_closest<MODE_CLOSEST_BVH, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin,
                               count, counters, hit_t, hit_ref, order, fr, prim_s0); }
        else
            { if (prim) hipLaunchKernelGGL((k_closest<MODE_CLOSEST, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin, count,
                               counters, hit_t, hit_ref, order, fr, prim_s0);
        else hipLaunchKernelGGL((k_closest<MODE_CLOSEST, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin, count,
                               counters, hit_t, hit_ref, order, fr, prim_s0); }
        hipLaunchKernelGGL(k_closest_retry, dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, pool, begin, count, hit_t,
                           hit_ref, fr.gloss_samples, hit_count, child_count, fr, prim_s0);
    }
    if (record_after_closest) (void)hipEventRecord((hipEvent_t)record_after_closest, (hipStream_t)stream);
    if (wait_before_shade) (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_before_shade, 0);
    if (wait_before_shade2) (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_before_shade2, 0);
    // (after the waits: it writes the queue's group count) also moves the pool's tail past the step's children
    hipLaunchKernelGGL(k_group_base, dim3(grid_for(nb)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, hit_count, child_count, nb, hit_base, child_base,
                       sq.count, child_begin, pool.cap, counters, host_step, host_seq);
    const bool sorted = want_groups && sortbuf && sq.cell && sq.order;
    if (!sorted) sq.cell = nullptr;
    if (sorted) (void)hipMemsetAsync(sq.cell, 0xFF, (size_t)count * sizeof(uint32_t), (hipStream_t)stream);   // (slots past the queue's count sort last)
    if (sc.shade_features || !(fr.flags & RTX_FLAG_INTERNAL_SMALL_SHADE))
        { if (prim) hipLaunchKernelGGL((k_shade<true, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, pool, begin, count, hit_t,
                           hit_ref, sq, counters, acc, prim_id_out, hit_base, child_base, child_begin, prim_s0);
        else hipLaunchKernelGGL((k_shade<true, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, pool, begin, count, hit_t,
                           hit_ref, sq, counters, acc, prim_id_out, hit_base, child_base, child_begin, prim_s0); }
    else
        { if (prim) hipLaunchKernelGGL((k_shade<false, true>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, pool, begin, count, hit_t,
                           hit_ref, sq, counters, acc, prim_id_out, hit_base, child_base, child_begin, prim_s0);
        else hipLaunchKernelGGL((k_shade<false, false>), dim3(nb), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, pool, begin, count, hit_t,
                           hit_ref, sq, counters, acc, prim_id_out, hit_base, child_base, child_begin, prim_s0); }
    if (sorted) rtxk_sort_groups(stream, sq, *sortbuf, count);
}
const char *rtxk_shadow(rtx_stream_t stream, const DScene &sc, const DFrame &fr, DShadowQ sq, uint32_t light, uint32_t n_groups,
                        uint32_t tries, DCounters *counters, double *acc, bool coop_defer) {
    if (n_groups == 0 || tries == 0) return "";
    uint32_t pow2 = (tries <= 64u && (tries & (tries - 1u)) == 0u) ? 1u : 0u;
    if (!rtx_shadow_defers(tries)) sq.res = nullptr;   // (a group's tries span waves: every lane adds its own share, shadow_kernel_body)
    uint64_t lanes = (uint64_t)n_groups * tries;
    // the cooperative walk pays off when the 16 rays of a group form a narrow beam: the feelers of ONE hit
    // point towards one light (tries a multiple of 16); groups mixing hit points can be arbitrarily
    // wide (a ground plane seen near the horizon) and walk most of the hierarchy
    const bool exact = (fr.flags & RTX_FLAG_INTERNAL_EXACT_SHADOWS) != 0u;   // re-render after a retry-list overflow
    const bool coop = sc.all_opaque && sc.wide && !(fr.flags & RTX_FLAG_NO_COOP) && !exact;
    const bool bvh = sc.bvh4 && !(fr.flags & RTX_FLAG_OCTREE_WALK);   // per-lane walks over the 4-wide hierarchy
    const bool product_bvh = !sc.all_opaque && !exact && !(fr.flags & RTX_FLAG_REFERENCE_ORDER) && bvh && sc.prim_leaves;
    // the any-hit kernels hand degenerate feelers to k_shadow_retry, the two-phase product those it cannot answer
    const bool any_hit = (sc.all_opaque != 0 && !exact) || product_bvh;
    // the per-lane kernel behind k_shadow_coop<false, true> works through a list that is usually short and whose length only the device
    // knows: a small grid, looping (k_shadow<.., LOOP>)
    const unsigned list_grid = std::min<unsigned>(grid_for(lanes), 2048u);
    if (any_hit) (void)hipMemsetAsync(sq.retry + (size_t)light * (2u + 2u * (size_t)sq.retry_cap), 0, sizeof(uint32_t), (hipStream_t)stream);
    // the name of the kernel the chain below picks for the launch's work (the same conditions, in the same order)
    static_assert(MODE_ANY == 2 && MODE_PRODUCT == 3 && MODE_PRODUCT_FAST == 4 && MODE_ANY_BVH == 6 && MODE_PRODUCT_BVH == 7, "the kernel names below");
    const char *name = exact ? "k_shadow<3>"
                       : (coop && tries % 16u == 0u) ? ((coop_defer && pow2 != 0u) ? "k_shadow_coop<false, true>" : "k_shadow_coop<false, false>")
                       : (coop && (tries == 4u || tries == 8u)) ? "k_shadow_coop<true, true>"
                       : (sc.all_opaque && bvh) ? "k_shadow<6>"
                       : sc.all_opaque ? "k_shadow<2>"
                       : (fr.flags & RTX_FLAG_REFERENCE_ORDER) ? "k_shadow<3>"
                       : product_bvh ? "k_shadow<7>" : "k_shadow<4>";
    if (exact)   // the all-f64 product walk answers every feeler, degenerate or not (all-opaque scenes: a product of zeros)
        hipLaunchKernelGGL(k_shadow<MODE_PRODUCT>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, nullptr);
    else if (coop && tries % 16u == 0u) {        // the 16 lanes of a group = feelers of one hit point
        // coop_defer (scenes with planes whose frames keep finding them): hit points outside the scene's cube, whose 16 feelers enter it
        // far apart, go to the per-lane kernel through the list sq.wide (see `wide` in the kernel)
        const bool defer = coop_defer && pow2 != 0u;
        if (defer) {
            (void)hipMemsetAsync(sq.wide + (size_t)light * (1u + (size_t)sq.cap), 0, sizeof(uint32_t), (hipStream_t)stream);
            hipLaunchKernelGGL((k_shadow_coop<false, true>), dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc);
        } else {
            hipLaunchKernelGGL((k_shadow_coop<false, false>), dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc);
        }
        if (defer && bvh)
            hipLaunchKernelGGL(k_shadow_list<MODE_ANY_BVH>, dim3(list_grid), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc, sq.wide + (size_t)light * (1u + (size_t)sq.cap));
        else if (defer)
            hipLaunchKernelGGL(k_shadow_list<MODE_ANY>, dim3(list_grid), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc, sq.wide + (size_t)light * (1u + (size_t)sq.cap));
    } else if (coop && (tries == 4u || tries == 8u)) {   // measured: with 1 try per hit point (16 origins per group) the per-lane walk wins
        // groups of 16 / tries neighbouring hit points: the cooperative kernel takes the compact ones and lists the others
        // (sq.wide) for the per-lane kernel, which works through that list in full waves
        (void)hipMemsetAsync(sq.wide + (size_t)light * (1u + (size_t)sq.cap), 0, sizeof(uint32_t), (hipStream_t)stream);
        hipLaunchKernelGGL((k_shadow_coop<true, true>), dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc);
        if (bvh)
            hipLaunchKernelGGL(k_shadow<MODE_ANY_BVH>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc, sq.wide + (size_t)light * (1u + (size_t)sq.cap));
        else
            hipLaunchKernelGGL(k_shadow<MODE_ANY>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                               n_groups, tries, pow2, counters, acc, sq.wide + (size_t)light * (1u + (size_t)sq.cap));
    } else if (sc.all_opaque && bvh)
        hipLaunchKernelGGL(k_shadow<MODE_ANY_BVH>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, sq.order);
    else if (sc.all_opaque)
        hipLaunchKernelGGL(k_shadow<MODE_ANY>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, sq.order);
    else if (fr.flags & RTX_FLAG_REFERENCE_ORDER)   // all-f64 restatement of the reference's walk
        hipLaunchKernelGGL(k_shadow<MODE_PRODUCT>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, sq.order);
    else if (product_bvh)   // the distinct hit prims over the 4-wide hierarchy, then their multiplicities
        hipLaunchKernelGGL(k_shadow<MODE_PRODUCT_BVH>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, sq.order);
    else
        hipLaunchKernelGGL(k_shadow<MODE_PRODUCT_FAST>, dim3(grid_for(lanes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light,
                           n_groups, tries, pow2, counters, acc, sq.order);
    if (any_hit)
        hipLaunchKernelGGL(k_shadow_retry, dim3(64), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, fr, sq, light, tries, acc);
    return name;
}
// Scene build on the device: DNode32 / DEntry32 (see rtx_device.h) from the f64 records.  v_cvt_f32_f64 with directed rounding
// (toward -inf for a box's lower corner, toward +inf for its upper corner) is the outward rounding the conservative tests rely on.
__global__ void __launch_bounds__(RTX_BLOCK) k_build_nodes32(const DNode *nodes, DNode32 *out, int64_t n, double cx, double cy, double cz) {
    const int64_t i = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const DNode nd = nodes[i];
    DNode32 o;
    o.bmin[0] = __double2float_rd(nd.bmin[0] - cx); o.bmin[1] = __double2float_rd(nd.bmin[1] - cy); o.bmin[2] = __double2float_rd(nd.bmin[2] - cz);
    o.bmax[0] = __double2float_ru(nd.bmax[0] - cx); o.bmax[1] = __double2float_ru(nd.bmax[1] - cy); o.bmax[2] = __double2float_ru(nd.bmax[2] - cz);
    if (nd.first_child >= 0) {
        o.link = ((uint32_t)nd.first_child << 8) | nd.child_mask;
        o.aux = 0xFFFFFFFFu;
    } else {
        o.link = (uint32_t)nd.entry_begin;
        o.aux = (uint32_t)nd.entry_count;
    }
    out[i] = o;
}
__global__ void __launch_bounds__(RTX_BLOCK) k_build_entries32(const int32_t *leaf_refs, const DTri *tris, const DSphere *spheres, DEntry32 *out,
                                                               int64_t n, double cx, double cy, double cz) {
    const int64_t i = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int ref = leaf_refs[i];
    double mn[3], mx[3];
    if (ref >= 0) {                                                                      // triangle.rs:102-106
        const DTri &t = tris[ref];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fmin(fmin(t.v[a], t.v[3 + a]), t.v[6 + a]);
            mx[a] = fmax(fmax(t.v[a], t.v[3 + a]), t.v[6 + a]);
        }
    } else {                                                                             // sphere.rs:18-33
        const DSphere &sp = spheres[~ref];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = sp.c[a] - sp.r; mx[a] = sp.c[a] + sp.r; }
    }
    DEntry32 o;
    o.bmin[0] = __double2float_rd(mn[0] - cx); o.bmin[1] = __double2float_rd(mn[1] - cy); o.bmin[2] = __double2float_rd(mn[2] - cz);
    o.bmax[0] = __double2float_ru(mx[0] - cx); o.bmax[1] = __double2float_ru(mx[1] - cy); o.bmax[2] = __double2float_ru(mx[2] - cz);
    o.ref = ref;
    o.pad = 0;
    out[i] = o;
}
void rtxk_build_twins(rtx_stream_t stream, const DNode *nodes, DNode32 *nodes32, int64_t n_nodes, const int32_t *leaf_refs,
                      const DTri *tris, const DSphere *spheres, DEntry32 *entries32, int64_t n_entries, const double centre[3]) {
    if (n_nodes > 0)
        hipLaunchKernelGGL(k_build_nodes32, dim3(grid_for((uint64_t)n_nodes)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, nodes, nodes32, n_nodes,
                           centre[0], centre[1], centre[2]);
    if (n_entries > 0)
        hipLaunchKernelGGL(k_build_entries32, dim3(grid_for((uint64_t)n_entries)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, leaf_refs, tris,
                           spheres, entries32, n_entries, centre[0], centre[1], centre[2]);
}
// prim_leaf_boxes[e] = nodes32[prim_leaves[e]] (phase 2 of the two-phase shadow product reads a prim's leaf boxes as one run)
__global__ void __launch_bounds__(RTX_BLOCK) k_build_leaf_boxes(const DNode32 *nodes32, const int32_t *prim_leaves, DNode32 *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (i < n) out[i] = nodes32[prim_leaves[i]];
}
void rtxk_build_leaf_boxes(rtx_stream_t stream, const DNode32 *nodes32, const int32_t *prim_leaves, DNode32 *out, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(k_build_leaf_boxes, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, nodes32, prim_leaves, out, n);
}
void rtxk_shade_probe(rtx_stream_t stream, const DScene &sc, int which, int64_t n, const double *in, double *out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_shade_probe, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, which, n, in, out);
}
__global__ void k_init_counters(DCounters *counters, unsigned long long (*ray_slots)[RTX_RAY_SLOT_WORDS]) {
    DCounters c = {};
    c.ray_slots = ray_slots;
    *counters = c;
}
void rtxk_init_counters(rtx_stream_t stream, DCounters *counters, unsigned long long (*ray_slots)[RTX_RAY_SLOT_WORDS]) {
    hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(1), 0, (hipStream_t)stream, counters, ray_slots);
}

// The workspace of rtx_render_device_noalloc (rtx.h): ray pool, shadow queue, hit records, counters and sample accumulators of
// ONE call at a time, carved out by the host (rtx_api.cpp).  Every GPU the library runs on holds its own copy of this array at its
// own address until the process ends; the name below is a host-side stand-in, never a device address: host code uses the pointer
// hipGetSymbolAddress returns for the CURRENT device, resolved once per device and kept.
static __device__ __align__(256) unsigned char scratch[RTX_STATIC_SCRATCH_BYTES];
static void *g_scratch_ptr[RTX_MAX_DEVICES];   // per device; NULL = not resolved yet

void *rtxk_static_scratch(int device, int *hip_error) {
    if (hip_error) *hip_error = (int)hipSuccess;
    if (device < 0 || device >= RTX_MAX_DEVICES) {
        if (hip_error) *hip_error = (int)hipErrorInvalidDevice;
        return nullptr;
    }
    if (g_scratch_ptr[device]) return g_scratch_ptr[device];
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != device) e = hipSetDevice(device);   // the address belongs to the device that is current
    void *ptr = nullptr;
    if (e == hipSuccess) e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess || !ptr) {
        if (hip_error) *hip_error = (int)(e != hipSuccess ? e : hipErrorInvalidSymbol);
        return nullptr;
    }
    g_scratch_ptr[device] = ptr;
    return ptr;
}

void rtxk_accumulate(rtx_stream_t stream, DShadowQ sq, uint32_t n_lights, uint32_t lights, uint32_t n_groups, double *acc) {
    if (n_groups == 0 || lights == 0u) return;
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(n_groups)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sq, n_lights, lights, acc);
}
void rtxk_resolve(rtx_stream_t stream, const DFrame &fr, const double *acc, uint8_t *rgba_out) {
    uint64_t n_pix = (uint64_t)fr.n_blocks_local * fr.tiles_x * 64u;
    if (n_pix == 0) return;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n_pix)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, fr, acc, rgba_out);
}
void rtxk_nearest(rtx_stream_t stream, const DScene &sc, int64_t n, const double *origin, const double *direction,
                  int32_t *prim_id, double *t_out, uint32_t flags) {
    if (n <= 0) return;
    if (flags & RTX_FLAG_REFERENCE_ORDER)
        hipLaunchKernelGGL(k_nearest<MODE_CLOSEST_REF>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, n,
                           origin, direction, prim_id, t_out);
    else if (sc.bvh4 && !(flags & RTX_FLAG_OCTREE_WALK) && !sc.all_opaque)   // (the walk the frames of this scene take, as in rtxk_trace)
        hipLaunchKernelGGL(k_nearest<MODE_CLOSEST_BVH_DEALT>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, n, origin,
                           direction, prim_id, t_out);
    else if (sc.bvh4 && !(flags & RTX_FLAG_OCTREE_WALK))
        hipLaunchKernelGGL(k_nearest<MODE_CLOSEST_BVH>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, n, origin,
                           direction, prim_id, t_out);
    else
        hipLaunchKernelGGL(k_nearest<MODE_CLOSEST>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, n, origin,
                           direction, prim_id, t_out);
}
void rtxk_shadow_products(rtx_stream_t stream, const DScene &sc, int64_t n, const double *origin, const double *target,
                          double *out, uint32_t flags) {
    if (n <= 0) return;
    // always the multiset product here: with all-opaque scenes it equals the any-hit answer
    if (flags & RTX_FLAG_REFERENCE_ORDER)
        hipLaunchKernelGGL(k_shadow_products<MODE_PRODUCT>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc, n,
                           origin, target, out);
    else if (sc.bvh4 && sc.prim_leaves && !(flags & RTX_FLAG_OCTREE_WALK))
        hipLaunchKernelGGL(k_shadow_products<MODE_PRODUCT_BVH>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc,
                           n, origin, target, out);
    else
        hipLaunchKernelGGL(k_shadow_products<MODE_PRODUCT_FAST>, dim3(grid_for((uint64_t)n)), dim3(RTX_BLOCK), 0, (hipStream_t)stream, sc,
                           n, origin, target, out);
}
