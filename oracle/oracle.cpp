// oracle.cpp — CPU restatement of gyng/rust-raytracer's per-pixel trace loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under rust-raytracer_amd/ may include, link, load or call
// this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
// only as the checker / the timed CPU baseline.  The product path has no CPU fallback.
//
// Parity status: the Rust reference cannot be compiled here (no rustc/cargo, crates not
// vendored: SURVEY.md §8c), so this restatement is pinned by (a) the known answers held by
// the reference's own #[test]s (tests/test_oracle_known_answers.py lists them file:line) and
// (b) literal transcription, statement by statement, in the reference's operation order,
// compiled with -ffp-contract=off (Rust never fuses a*b+c).  Functions with no reference
// test (trace recursion, shadow_intensity, fresnel, BRDFs, checker, cubemap, camera, octree
// structure, quantisation) are "restatement-pinned" only.  Every stochastic draw is
// "parity unpinned": the reference uses rand 0.5.4 thread_rng() (OS-seeded, not
// reproducible; crate source absent), so the RNG here is this build's own counter-based
// Philox4x32-10 (spec in DESIGN.md §RNG); RTX_RNG_CENTER replaces every draw by 0.5.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference/src/).
#include "../include/rtx.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double F64_EPSILON = 2.220446049250313e-16;   // std::f64::EPSILON
const double EPSILON = F64_EPSILON * 10000.0;        // raytracer/renderer.rs:12
const double PI = 3.14159265358979323846264338327950288; // std::f64::consts::PI

// ---------------------------------------------------------------- vec3.rs:6-235
struct Vec3 {
    double x, y, z;
};
inline Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // :129
inline Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // :153
inline Vec3 operator*(Vec3 a, Vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // :177
inline Vec3 operator*(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }       // :189
inline Vec3 operator-(Vec3 a) { return v3(-a.x, -a.y, -a.z); }                          // :225
inline double len(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }      // :30
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // :34
inline Vec3 cross(Vec3 a, Vec3 b) {                                                      // :38
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 unit(Vec3 a) {                                                               // :46
    double l = len(a);
    return v3(a.x / l, a.y / l, a.z / l);
}
inline Vec3 scale(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }            // :56
inline Vec3 reflect(Vec3 v, Vec3 n) { return scale(n, 2.0 * dot(n, v)) - v; }            // :70
// vec3.rs:77-92; returns false for total internal reflection (None)
inline bool refract(Vec3 v, Vec3 n, double ior, bool inside, Vec3 *out) {
    double n1, n2, n_dot_v;
    Vec3 nn;
    if (!inside) {
        n1 = 1.0; n2 = ior; n_dot_v = dot(n, v); nn = n;
    } else {
        n1 = ior; n2 = 1.0; n_dot_v = -dot(n, v); nn = -n;
    }
    double ratio = n1 / n2;
    double disc = 1.0 - ((ratio * ratio) * (1.0 - n_dot_v * n_dot_v));
    if (disc < 0.0) return false;
    *out = scale(v, -ratio) + scale(nn, ratio * n_dot_v - std::sqrt(disc));
    return true;
}
// f64::max / f64::min ignore a NaN operand, like fmax/fmin.   vec3.rs:102-108
inline double clamp01(double x) { return std::fmin(std::fmax(x, 0.0), 1.0); }
inline Vec3 clampv(Vec3 a) { return v3(clamp01(a.x), clamp01(a.y), clamp01(a.z)); }

// ---------------------------------------------------------------- RNG (this build's own)
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
const uint32_t TAG_ROOT = 0x52545831u;  // "RTX1"
const uint32_t TAG_CHILD = 0x4348494Cu; // "CHIL"
const uint32_t TAG_DRAW = 0x44524157u;  // "DRAW"
enum { STREAM_JITTER = 0, STREAM_LIGHT = 1, STREAM_GLOSS = 2 };
enum { CODE_MIRROR = 1, CODE_TRANSMIT = 2, CODE_GLOSS0 = 16 };

inline uint64_t root_key(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sub) {
    uint32_t ctr[4] = {pixel, sub, frame, TAG_ROOT};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    return (uint64_t)o[0] | ((uint64_t)o[1] << 32);
}
inline uint64_t child_key(uint64_t parent, uint32_t code) {
    uint32_t ctr[4] = {code, 0, 0, TAG_CHILD};
    uint32_t key[2] = {(uint32_t)parent, (uint32_t)(parent >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    return (uint64_t)o[0] | ((uint64_t)o[1] << 32);
}
struct Rng {
    uint32_t mode;
    // two uniform draws in [0,1) with 53 random bits each (rand's gen::<f64>() precision)
    void draw2(uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *a, double *b) const {
        if (mode == RTX_RNG_CENTER) { *a = 0.5; *b = 0.5; return; }
        uint32_t ctr[4] = {stream, i, j, TAG_DRAW};
        uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
        uint32_t o[4];
        philox4x32_10(ctr, k, o);
        uint64_t ua = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
        uint64_t ub = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
        *a = (double)(ua >> 11) * (1.0 / 9007199254740992.0);
        *b = (double)(ub >> 11) * (1.0 / 9007199254740992.0);
    }
    // three uniform draws in (0,1) from ONE Philox block, 32 random bits each: (word + 0.5) / 2^32
    // (exact in f64).  Used where the reference draws three numbers in a row (spherelight.rs:14-19,
    // vec3.rs:114-116).
    void draw3(uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *a, double *b, double *c) const {
        if (mode == RTX_RNG_CENTER) { *a = 0.5; *b = 0.5; *c = 0.5; return; }
        uint32_t ctr[4] = {stream, i, j, TAG_DRAW};
        uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
        uint32_t o[4];
        philox4x32_10(ctr, k, o);
        *a = ((double)o[0] + 0.5) * (1.0 / 4294967296.0);
        *b = ((double)o[1] + 0.5) * (1.0 / 4294967296.0);
        *c = ((double)o[2] + 0.5) * (1.0 / 4294967296.0);
    }
};

// ---------------------------------------------------------------- counters
struct Counters {
    uint64_t rays[5] = {0, 0, 0, 0, 0}; // primary, reflect, gloss, refract, shadow
    // traversal work split by ray class: [0] closest-hit rays (ray.rs:44-66), [1] shadow feelers
    uint64_t b_node[2] = {0, 0}, b_leaf[2] = {0, 0}, p_tri[2] = {0, 0}, p_sph[2] = {0, 0}, p_pln[2] = {0, 0};
    uint64_t hits = 0, mat_fetch = 0, texels = 0;
    int cls = 0;
    void add(const Counters &o) {
        for (int i = 0; i < 5; ++i) rays[i] += o.rays[i];
        for (int c = 0; c < 2; ++c) {
            b_node[c] += o.b_node[c]; b_leaf[c] += o.b_leaf[c]; p_tri[c] += o.p_tri[c]; p_sph[c] += o.p_sph[c]; p_pln[c] += o.p_pln[c];
        }
        hits += o.hits; mat_fetch += o.mat_fetch; texels += o.texels;
    }
};
enum { RAY_PRIMARY = 0, RAY_REFLECT = 1, RAY_GLOSS = 2, RAY_REFRACT = 3, RAY_SHADOW = 4 };

// ---------------------------------------------------------------- raytracer/ray.rs:15-42
struct Ray {
    Vec3 origin, direction, inverse_dir;
    bool signs[3];
};
inline Ray ray_new(Vec3 origin, Vec3 direction) {
    double inv_x = 1.0 / direction.x, inv_y = 1.0 / direction.y, inv_z = 1.0 / direction.z;
    Ray r;
    r.origin = origin; r.direction = direction; r.inverse_dir = v3(inv_x, inv_y, inv_z);
    r.signs[0] = inv_x > 0.0; r.signs[1] = inv_y > 0.0; r.signs[2] = inv_z > 0.0;
    return r;
}

// ---------------------------------------------------------------- geometry/bbox.rs
struct BBox {
    Vec3 min, max;
};
inline BBox union_point(const BBox &b, Vec3 p) {                   // bbox.rs:29-42
    return BBox{v3(std::fmin(b.min.x, p.x), std::fmin(b.min.y, p.y), std::fmin(b.min.z, p.z)),
                v3(std::fmax(b.max.x, p.x), std::fmax(b.max.y, p.y), std::fmax(b.max.z, p.z))};
}
inline BBox union_points(Vec3 p1, Vec3 p2) {                       // bbox.rs:45-58
    return BBox{v3(std::fmin(p1.x, p2.x), std::fmin(p1.y, p2.y), std::fmin(p1.z, p2.z)),
                v3(std::fmax(p1.x, p2.x), std::fmax(p1.y, p2.y), std::fmax(p1.z, p2.z))};
}
inline BBox union_bbox(const BBox &a, const BBox &b) {             // bbox.rs:62-75
    return BBox{v3(std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z)),
                v3(std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z))};
}
// bbox.rs:99-165 — comparisons transcribed literally (no fmin/fmax: a NaN keeps the old value)
inline bool bbox_intersects(const BBox &b, const Ray &ray) {
    Vec3 o = ray.origin;
    Vec3 min_bound = ray.signs[0] ? b.min : b.max;
    Vec3 max_bound = ray.signs[0] ? b.max : b.min;
    double t_min = (min_bound.x - o.x) * ray.inverse_dir.x;
    double t_max = (max_bound.x - o.x) * ray.inverse_dir.x;

    Vec3 min_y_bound = ray.signs[1] ? b.min : b.max;
    Vec3 max_y_bound = ray.signs[1] ? b.max : b.min;
    double ty_min = (min_y_bound.y - o.y) * ray.inverse_dir.y;
    double ty_max = (max_y_bound.y - o.y) * ray.inverse_dir.y;

    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;

    Vec3 min_z_bound = ray.signs[2] ? b.min : b.max;
    Vec3 max_z_bound = ray.signs[2] ? b.max : b.min;
    double tz_min = (min_z_bound.z - o.z) * ray.inverse_dir.z;
    double tz_max = (max_z_bound.z - o.z) * ray.inverse_dir.z;

    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;

    return t_min < INF && t_max > 0.0;
}
inline bool bbox_overlaps(const BBox &a, const BBox &o) {          // bbox.rs:167-173
    bool x = a.max.x >= o.min.x && a.min.x <= o.max.x;
    bool y = a.max.y >= o.min.y && a.min.y <= o.max.y;
    bool z = a.max.z >= o.min.z && a.min.z <= o.max.z;
    return x && y && z;
}

// ---------------------------------------------------------------- textures
struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;
};
struct Color4 {
    double r, g, b, a;
};
// compositor/colorrgba.rs:116-124 (u8 channels), :178-189 (Mul<f64>: alpha untouched), :138-148 (Add)
inline Color4 px_f64(const Image &im, size_t x, size_t y) {
    // The reference panics on an out-of-range index (surface.rs:102-109); that only happens
    // for 1-pixel-wide/high images.  Clamp instead so the oracle never reads out of bounds.
    if (x >= (size_t)im.w) x = (size_t)im.w - 1;
    if (y >= (size_t)im.h) y = (size_t)im.h - 1;
    const uint8_t *p = &im.rgba[(y * (size_t)im.w + x) * 4];
    return Color4{p[0] / 255.0, p[1] / 255.0, p[2] / 255.0, p[3] / 255.0};
}
inline Color4 cmul(Color4 c, double s) { return Color4{c.r * s, c.g * s, c.b * s, c.a}; }
inline Color4 cadd(Color4 a, Color4 b) { return Color4{a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
// `as usize` saturates: negative and NaN -> 0
inline size_t as_usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 1.8446744073709552e19) return (size_t)-1;
    return (size_t)v;
}
// material/textures/imagetexture.rs:26-47
inline Color4 image_color(const Image &im, double u, double v, Counters *c) {
    double s = std::fmod(u, 1.0) * ((double)im.w - 1.0);
    double t = std::fmod(v, 1.0) * ((double)im.h - 1.0);
    size_t x = as_usize(std::floor(s));
    size_t y = as_usize(std::floor(t));
    double u_ratio = s - (double)x;
    double v_ratio = t - (double)y;
    double u_opposite = 1.0 - u_ratio;
    double v_opposite = 1.0 - v_ratio;
    if (c) c->texels += 4;
    return cadd(cmul(cadd(cmul(px_f64(im, x, y), u_opposite), cmul(px_f64(im, x + 1, y), u_ratio)), v_opposite),
                cmul(cadd(cmul(px_f64(im, x, y + 1), u_opposite), cmul(px_f64(im, x + 1, y + 1), u_ratio)), v_ratio));
}
struct Texture {
    int kind = RTX_TEX_CHECKER;
    Vec3 color1{0, 0, 0}, color2{0, 0, 0};
    double scale = 1.0;
    const Image *image = nullptr;
};
// ColorRGBA<f64>::to_vec3 (colorrgba.rs:128-135) divides by max_value() = 1.0
inline Vec3 to_vec3_f64(double r, double g, double b) { return v3(r / 1.0, g / 1.0, b / 1.0); }
inline Vec3 texture_color(const Texture &tx, double u, double v, Counters *c) {
    switch (tx.kind) {
    case RTX_TEX_CHECKER: {                                        // checkertexture.rs:13-23
        double s = std::fabs(std::fmod(u, tx.scale));
        double t = std::fabs(std::fmod(v, tx.scale));
        double half = tx.scale / 2.0;
        if ((s > half && t < half) || (s < half && t > half)) return to_vec3_f64(tx.color1.x, tx.color1.y, tx.color1.z);
        return to_vec3_f64(tx.color2.x, tx.color2.y, tx.color2.z);
    }
    case RTX_TEX_UV:                                               // uvtexture.rs:10-14
        return to_vec3_f64(std::fmod(u, 1.0) + 0.0, std::fmod(v, 1.0) + 0.0, 0.0);
    default: {
        Color4 col = image_color(*tx.image, u, v, c);
        return to_vec3_f64(col.r, col.g, col.b);
    }
    }
}
// material/textures/cubemap.rs:45-84
inline Vec3 cubemap_color(const Image *const faces[6], Vec3 dir, Counters *c) {
    double x_mag = std::fabs(dir.x), y_mag = std::fabs(dir.y), z_mag = std::fabs(dir.z);
    int face = -1;
    double s = 0.0, t = 0.0;
    if (x_mag >= y_mag && x_mag >= z_mag) {
        face = dir.x <= 0.0 ? 0 : 1;
        double sc = dir.x < 0.0 ? 1.0 : -1.0;
        s = sc * dir.z / std::fabs(dir.x);
        t = dir.y / std::fabs(dir.x);
    } else if (y_mag >= x_mag && y_mag >= z_mag) {
        face = dir.y <= 0.0 ? 2 : 3;
        double sc = dir.y < 0.0 ? 1.0 : -1.0;
        s = sc * dir.x / std::fabs(dir.y);
        t = dir.z / std::fabs(dir.y);
    } else if (z_mag >= y_mag && z_mag >= x_mag) {
        face = dir.z <= 0.0 ? 4 : 5;
        double sc = dir.z < 0.0 ? -1.0 : 1.0;
        s = sc * dir.x / std::fabs(dir.z);
        t = dir.y / std::fabs(dir.z);
    }
    double seam_delta = 0.0001;
    s = std::fmin(std::fmax(1.0 - (s * 0.5 + 0.5), seam_delta), 1.0 - seam_delta);
    t = std::fmin(std::fmax(1.0 - (t * 0.5 + 0.5), seam_delta), 1.0 - seam_delta);
    if (face < 0) return v3(0, 0, 0); // the reference panics (cubemap.rs:79-81): NaN direction
    Color4 col = image_color(*faces[face], s, t, c);               // ImageTexture::sample, imagetexture.rs:20-22
    return to_vec3_f64(col.r, col.g, col.b);
}

// ---------------------------------------------------------------- materials
struct Material {
    rtx_material m;
    const Texture *tex = nullptr;
    // material.rs:4-14
    Vec3 sample(Vec3 n, Vec3 i, Vec3 l, double u, double v, Counters *c) const {
        Vec3 ambient_c = v3(m.ambient[0], m.ambient[1], m.ambient[2]);
        Vec3 diffuse_c = v3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
        Vec3 specular_c = v3(m.specular[0], m.specular[1], m.specular[2]);
        if (c) c->mat_fetch += 1;
        if (m.kind == RTX_MAT_FLAT) return diffuse_c;              // flatmaterial.rs:11-13
        Vec3 texc = tex ? texture_color(*tex, u, v, c) : to_vec3_f64(1.0, 1.0, 1.0); // ColorRGBA::white()
        if (m.kind == RTX_MAT_COOK_TORRANCE) {                     // cooktorrancematerial.rs:26-58
            Vec3 ambient = scale(ambient_c, m.k_a);
            Vec3 diffuse = scale(scale(diffuse_c, m.k_d), dot(n, l)) * texc;
            Vec3 h = unit(l + i);
            double n_dot_h = dot(n, h);
            double n_dot_l = dot(n, l);
            double v_dot_h = dot(i, h);
            double n_dot_v = dot(n, i);
            double n1 = 1.0, n2 = m.ior;
            double f0r = (n1 - n2) / (n1 + n2);
            double f0 = f0r * f0r;                                 // powf(2.0)
            double f = std::pow(1.0 - v_dot_h, 5.0) * (1.0 - f0) + f0;
            double alpha = std::acos(n_dot_h);
            double d = m.gauss_constant * std::exp(-alpha / std::sqrt(m.shine));
            double g1 = (2.0 * n_dot_h * n_dot_v) / v_dot_h;
            double g2 = (2.0 * n_dot_h * n_dot_l) / v_dot_h;
            double g = std::fmin(g1, g2);
            double brdf = f * d * g / (n_dot_v * n_dot_l * PI);
            return scale(specular_c, m.k_s * brdf) + diffuse + ambient;
        }
        // phongmaterial.rs:24-36
        Vec3 h = unit(l + i);
        Vec3 ambient = scale(ambient_c, m.k_a);
        Vec3 diffuse = scale(scale(diffuse_c, m.k_d), dot(n, l)) * texc;
        Vec3 specular = scale(scale(specular_c, m.k_s), std::pow(dot(n, h), m.shine));
        return ambient + diffuse + specular;
    }
    bool is_reflective() const { return m.kind != RTX_MAT_FLAT && m.k_sg > 0.0; }
    bool is_refractive() const { return m.kind != RTX_MAT_FLAT && m.k_tg > 0.0; }
    Vec3 global_specular(Vec3 c) const { return m.kind == RTX_MAT_FLAT ? v3(0, 0, 0) : scale(c, m.k_sg); }
    Vec3 global_transmissive(Vec3 c) const { return m.kind == RTX_MAT_FLAT ? v3(0, 0, 0) : scale(c, m.k_tg); }
    Vec3 transmission() const {
        return m.kind == RTX_MAT_FLAT ? v3(0, 0, 0) : v3(m.transmission[0], m.transmission[1], m.transmission[2]);
    }
    double ior() const { return m.kind == RTX_MAT_FLAT ? 1.0 : m.ior; }
    bool is_glossy() const { return m.kind != RTX_MAT_FLAT && m.glossiness > F64_EPSILON; }
    double glossiness() const { return m.kind == RTX_MAT_FLAT ? 0.0 : m.glossiness; }
};

// ---------------------------------------------------------------- raytracer/intersection.rs:4-11
struct Intersection {
    Vec3 n;
    double t, u, v;
    Vec3 position;
    const Material *material;
    int prim_id; // not in the reference: index in Octree.prims, planes = n_finite + j
};

// ---------------------------------------------------------------- prims
struct Prim {
    int kind;
    int prim_id = -1;
    const Material *material = nullptr;
    // sphere
    Vec3 center{0, 0, 0};
    double radius = 0;
    // plane
    double a = 0, b = 0, c = 0, d = 0;
    // triangle
    Vec3 vertices[3], normals[3];
    double tu[3], tv[3];

    bool partial_bounding_box(BBox *out) const {
        switch (kind) {
        case RTX_PRIM_SPHERE:                                      // sphere.rs:18-33
            *out = BBox{v3(center.x - radius, center.y - radius, center.z - radius),
                        v3(center.x + radius, center.y + radius, center.z + radius)};
            return true;
        case RTX_PRIM_TRIANGLE:                                    // triangle.rs:102-106
            *out = union_point(union_points(vertices[0], vertices[1]), vertices[2]);
            return true;
        default:                                                   // plane.rs:19-23
            return false;
        }
    }

    bool intersects(const Ray &ray, double t_min, double t_max, Intersection *out, Counters *cn) const {
        switch (kind) {
        case RTX_PRIM_SPHERE: {                                    // sphere.rs:36-73
            if (cn) cn->p_sph[cn->cls] += 1;
            Vec3 i = ray.origin - center;
            double a_ = 1.0;
            double b_ = 2.0 * dot(ray.direction, i);
            double c_ = dot(i, i) - radius * radius;
            double discriminant = b_ * b_ - 4.0 * a_ * c_;
            if (discriminant <= 0.0) return false;
            double disc_sqrt = std::sqrt(discriminant);
            double t1 = (-b_ + disc_sqrt) / 2.0 * a_;
            double t2 = (-b_ - disc_sqrt) / 2.0 * a_;
            if ((t1 >= t_min && t1 <= t_max) || (t2 >= t_min && t2 <= t_max)) {
                double t = std::fabs(t1) < std::fabs(t2) ? t1 : t2;
                Vec3 p = ray.origin + scale(ray.direction, t);
                Vec3 n = unit(p - center);
                double u = 0.5 + std::atan2(n.z, n.x) / (PI * 2.0);
                double v = 0.5 - std::asin(n.y) / PI;
                *out = Intersection{n, t, u, v, p, material, prim_id};
                return true;
            }
            return false;
        }
        case RTX_PRIM_PLANE: {                                     // plane.rs:26-50
            if (cn) cn->p_pln[cn->cls] += 1;
            Vec3 n = v3(a, b, c);
            double nrd = dot(n, ray.direction);
            double nro = dot(n, ray.origin);
            double t = (-d - nro) / nrd;
            if (t < t_min || t > t_max) return false;
            Vec3 p = ray.origin + scale(ray.direction, t);
            Vec3 u_axis = v3(n.y, n.z, -n.x);
            Vec3 v_axis = cross(u_axis, n);
            double u = dot(p, u_axis);
            double v = dot(p, v_axis);
            *out = Intersection{n, t, u, v, p, material, prim_id};
            return true;
        }
        default: {                                                 // triangle.rs:111-156
            if (cn) cn->p_tri[cn->cls] += 1;
            Vec3 e1 = vertices[1] - vertices[0];
            Vec3 e2 = vertices[2] - vertices[0];
            Vec3 p = cross(ray.direction, e2);
            double det = dot(e1, p);
            if (det > -F64_EPSILON && det < F64_EPSILON) return false;
            double inv_det = 1.0 / det;
            Vec3 s = ray.origin - vertices[0];
            double beta = inv_det * dot(s, p);
            if (beta < 0.0 || beta > 1.0) return false;
            Vec3 q = cross(s, e1);
            double gamma = inv_det * dot(ray.direction, q);
            if (gamma < 0.0 || beta + gamma > 1.0) return false;
            double t = inv_det * dot(e2, q);
            if (t < t_min || t > t_max) return false;
            Vec3 ip = ray.origin + scale(ray.direction, t);
            double alpha = 1.0 - beta - gamma;
            Vec3 n = scale(normals[0], alpha) + scale(normals[1], beta) + scale(normals[2], gamma);
            double u = tu[0] * alpha + tu[1] * beta + tu[2] * gamma;
            double v = tv[0] * alpha + tv[1] * beta + tv[2] * gamma;
            *out = Intersection{n, t, u, v, ip, material, prim_id};
            return true;
        }
        }
    }
};

// ---------------------------------------------------------------- raytracer/octree.rs
struct OctreeData {                                                // octree.rs:58-62
    BBox bbox;
    size_t index;
};
struct OctreeNode {                                                // octree.rs:51-56
    BBox bbox;
    int depth;
    std::vector<OctreeNode> children;
    std::vector<OctreeData> leaf_data;

    bool is_leaf() const { return children.empty(); }              // :130-132
    void subdivide() {                                             // :75-98
        children.reserve(8);
        for (unsigned x = 0; x < 2; ++x)
            for (unsigned y = 0; y < 2; ++y)
                for (unsigned z = 0; z < 2; ++z) {
                    Vec3 l = bbox.max - bbox.min;                  // BBox::len, bbox.rs:244
                    BBox cb;
                    cb.min = v3(bbox.min.x + (double)x * l.x / 2.0,
                                bbox.min.y + (double)y * l.y / 2.0,
                                bbox.min.z + (double)z * l.z / 2.0);
                    cb.max = v3(bbox.max.x - (double)(1 - x) * l.x / 2.0,
                                bbox.max.y - (double)(1 - y) * l.y / 2.0,
                                bbox.max.z - (double)(1 - z) * l.z / 2.0);
                    OctreeNode ch;
                    ch.bbox = cb;
                    ch.depth = depth - 1;
                    children.push_back(std::move(ch));
                }
    }
    void insert(size_t index, const BBox &object_bbox) {           // :101-128
        if (depth <= 0) {
            leaf_data.push_back(OctreeData{object_bbox, index});
            return;
        }
        if (is_leaf() && leaf_data.size() == 0) {
            leaf_data.push_back(OctreeData{object_bbox, index});
            return;
        }
        if (is_leaf() && leaf_data.size() == 1) {
            subdivide();
            OctreeData old = leaf_data[0];
            leaf_data.erase(leaf_data.begin());
            insert(old.index, old.bbox);
        }
        for (auto &child : children)
            if (bbox_overlaps(child.bbox, object_bbox)) child.insert(index, object_bbox);
    }
};
struct Octree {                                                    // octree.rs:9-13
    std::vector<const Prim *> prims, infinites;
    OctreeNode root;
    int depth = 0;
};
// `x as i32` saturates and maps NaN to 0
inline int as_i32(double v) {
    if (v != v) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int)v;
}
// Octree::from_iter, octree.rs:15-43
void octree_from_iter(Octree &oct, const std::vector<const Prim *> &all) {
    BBox bb;
    for (const Prim *p : all) (p->partial_bounding_box(&bb) ? oct.prims : oct.infinites).push_back(p);
    bool have = false;
    BBox bounds{v3(0, 0, 0), v3(0, 0, 0)};                          // BBox::zero() if there is nothing
    for (const Prim *p : oct.prims) {                              // BBox::from_union, bbox.rs:85-97
        p->partial_bounding_box(&bb);
        bounds = have ? union_bbox(bounds, bb) : bb;
        have = true;
    }
    // f64::log(self, base) = self.ln() / base.ln()
    int depth = as_i32(std::round(1.2 * (std::log((double)oct.prims.size()) / std::log(8.0))));
    oct.depth = depth;
    oct.root.bbox = bounds;
    oct.root.depth = depth;
    for (size_t i = 0; i < oct.prims.size(); ++i) {
        oct.prims[i]->partial_bounding_box(&bb);
        oct.root.insert(i, bb);
    }
}
// OctreeIterator::next, octree.rs:159-189: unordered DFS, children pushed 0..7 (popped 7..0),
// the popped node's leaf entries filtered by their own bbox, infinites last.
template <class F>
inline void for_each_candidate(const Octree &oct, const Ray &ray, Counters *cn, F &&f) {
    std::vector<const OctreeNode *> stack;                         // vec![&octree.root], :149
    stack.push_back(&oct.root);
    while (!stack.empty()) {
        const OctreeNode *node = stack.back();
        stack.pop_back();
        for (const auto &child : node->children) {
            if (cn) cn->b_node[cn->cls] += 1;
            if (bbox_intersects(child.bbox, ray)) stack.push_back(&child);
        }
        for (const auto &e : node->leaf_data) {
            if (cn) cn->b_leaf[cn->cls] += 1;
            if (bbox_intersects(e.bbox, ray)) f(oct.prims[e.index]);
        }
    }
    for (const Prim *p : oct.infinites) f(p);
}

// ---------------------------------------------------------------- lights
struct Light {
    rtx_light l;
    bool is_point() const { return l.kind == RTX_LIGHT_POINT; }
    Vec3 center() const { return v3(l.position[0], l.position[1], l.position[2]); }
    Vec3 color() const { return v3(l.color[0], l.color[1], l.color[2]); }
    // pointlight.rs:11-13 / spherelight.rs:13-23 (a CUBE of side `radius`; jitters even for 1 try)
    Vec3 position(const Rng &rng, uint64_t key, uint32_t light_index, uint32_t k) const {
        if (is_point()) return center();
        double ux, uy, uz;
        rng.draw3(key, STREAM_LIGHT, light_index, k, &ux, &uy, &uz);
        Vec3 jitter = v3(l.radius * (ux - 0.5), l.radius * (uy - 0.5), l.radius * (uz - 0.5));
        return center() + jitter;
    }
};

// ---------------------------------------------------------------- scene/camera.rs
struct Camera {
    Vec3 position, look_at, up, eye, right;
    double fov_deg, half_width, half_height, pixel_width, pixel_height;
    uint32_t image_width, image_height;
};
Camera camera_new(const rtx_camera &c) {                           // camera.rs:26-50, 88-105
    Camera cam;
    cam.position = v3(c.position[0], c.position[1], c.position[2]);
    cam.look_at = v3(c.look_at[0], c.look_at[1], c.look_at[2]);
    cam.up = v3(c.up[0], c.up[1], c.up[2]);
    cam.fov_deg = c.fov_deg;
    cam.image_width = c.image_width;
    cam.image_height = c.image_height;
    cam.eye = unit(cam.look_at - cam.position);
    cam.right = cross(cam.eye, cam.up);
    double fov_rad = cam.fov_deg * (PI / 180.0);                   // f64::to_radians
    double ratio = (double)cam.image_height / (double)cam.image_width;
    cam.half_width = std::tan(fov_rad);
    cam.half_height = cam.half_width * ratio;
    double camera_width = cam.half_width * 2.0;
    double camera_height = cam.half_height * 2.0;
    cam.pixel_width = camera_width / (double)(cam.image_width - 1);
    cam.pixel_height = camera_height / (double)(cam.image_height - 1);
    return cam;
}
inline Ray camera_get_ray(const Camera &c, double x, double y) {   // camera.rs:62-68
    return ray_new(c.position, unit(c.eye + scale(c.right, x * c.pixel_width - c.half_width) +
                                    scale(c.up, y * c.pixel_height - c.half_height)));
}

// ---------------------------------------------------------------- scene/scene.rs:7-12
struct Scene {
    std::vector<Image> images;
    std::vector<Texture> textures;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<Prim> prim_store;
    Octree octree;
    Vec3 background;
    bool has_skybox = false;
    const Image *skybox[6];
};

struct Options {                                                   // renderer.rs:14-21
    uint32_t reflect_depth, refract_depth, shadow_samples, gloss_samples, pixel_samples;
};

struct Ctx {
    const Scene *scene;
    Rng rng;
    Counters cn;
};

// Ray::get_nearest_hit, ray.rs:44-66
bool get_nearest_hit(const Scene &scene, const Ray &ray, Intersection *nearest_hit, Counters *cn) {
    double t_min = 0.000001;
    bool have = false;
    double nearest_t = INF;
    for_each_candidate(scene.octree, ray, cn, [&](const Prim *prim) {
        Intersection isect;
        if (prim->intersects(ray, t_min, nearest_t, &isect, cn)) {
            if (isect.t > t_min && isect.t < nearest_t) {
                nearest_t = isect.t;
                *nearest_hit = isect;
                have = true;
            }
        }
    });
    return have;
}

// Ray::perturb (ray.rs:68-81) with Vec3::random (vec3.rs:112-126) drawn from the child's key
Ray ray_perturb(const Ray &r, double magnitude, const Rng &rng, uint64_t key) {
    double u_phi, u_cos, u_r;
    rng.draw3(key, STREAM_GLOSS, 0, 0, &u_phi, &u_cos, &u_r);
    double phi = 0.0 + (2.0 * PI - 0.0) * u_phi;                   // gen_range(0, 2pi)
    double costheta = -1.0 + (1.0 - -1.0) * u_cos;                 // gen_range(-1, 1)
    double u = 0.0 + (1.0 - 0.0) * u_r;                            // gen_range(0, 1)
    double theta = std::acos(costheta);
    double rr = std::pow(u, 1.0 / 3.0);
    Vec3 rand_vec = v3(rr * std::sin(theta) * std::cos(phi), rr * std::sin(theta) * std::sin(phi),
                       rr * std::cos(theta)) * magnitude;
    Vec3 corrected = dot(rand_vec, r.direction) < 0.0 ? rand_vec * -1.0 : rand_vec;
    Vec3 direction = unit(corrected + r.direction);
    return ray_new(r.origin, direction);
}

// Renderer::fresnel_reflect, renderer.rs:226-247
double fresnel_reflect(double ior, Vec3 i, Vec3 n, bool inside) {
    double n1 = inside ? ior : 1.0, n2 = inside ? 1.0 : ior;
    Vec3 actual_n = inside ? -n : n;
    double r0_sqrt = (n1 - n2) / (n1 + n2);
    double r0 = r0_sqrt * r0_sqrt;
    double cos_angle;
    if (n1 <= n2) {
        cos_angle = dot(i, actual_n);
    } else {
        Vec3 t;
        if (!refract(i, -actual_n, ior, inside, &t)) return 1.0;
        cos_angle = -dot(actual_n, t);                             // `-actual_n.dot(&t)`: unary minus binds
    }                                                              //  after the method call
    double cos_term = 1.0 - cos_angle;
    return std::fmin(std::fmax(r0 + ((1.0 - r0) * cos_term * cos_term * cos_term * cos_term * cos_term), 0.0), 1.0);
}

// Renderer::shadow_intensity, renderer.rs:188-221
Vec3 shadow_intensity(Ctx &cx, const Intersection &hit, const Light &light, uint32_t light_index,
                      uint32_t shadow_samples, uint64_t key) {
    if (shadow_samples <= 0) return v3(1, 1, 1);
    uint32_t tries = light.is_point() ? 1 : shadow_samples;
    Vec3 shadow = v3(0, 0, 0);
    for (uint32_t k = 0; k < tries; ++k) {
        Vec3 sampled = light.position(cx.rng, key, light_index, k);
        Vec3 shadow_l = unit(sampled - hit.position);
        Ray shadow_ray = ray_new(hit.position, shadow_l);
        double distance_to_light = len(sampled - hit.position);
        cx.cn.rays[RAY_SHADOW] += 1;
        cx.cn.cls = 1;
        Vec3 acc = v3(1, 1, 1);
        for_each_candidate(cx.scene->octree, shadow_ray, &cx.cn, [&](const Prim *prim) {
            Intersection occ;
            if (prim->intersects(shadow_ray, EPSILON, distance_to_light, &occ, &cx.cn))
                acc = acc * occ.material->transmission();
        });
        cx.cn.cls = 0;
        shadow = shadow + acc;
    }
    return scale(shadow, 1.0 / (double)tries);
}

Vec3 trace(Ctx &cx, const Ray &ray, Options options, bool inside, uint64_t key, int ray_type, int *prim_id_out);

// Renderer::global_reflection, renderer.rs:148-168
Vec3 global_reflection(Ctx &cx, const Intersection &hit, Options options, bool inside, Vec3 i, Vec3 n,
                       double reflect_fresnel, uint64_t key) {
    Vec3 r = reflect(i, n);
    Ray reflect_ray = ray_new(hit.position, r);
    Options next = options;
    next.reflect_depth = options.reflect_depth - 1;
    Vec3 reflection;
    if (hit.material->is_glossy()) {
        Vec3 acc = v3(0, 0, 0);
        for (uint32_t g = 0; g < options.gloss_samples; ++g) {
            uint64_t ck = child_key(key, CODE_GLOSS0 + g);
            Ray gloss_ray = ray_perturb(reflect_ray, hit.material->glossiness(), cx.rng, ck);
            acc = acc + trace(cx, gloss_ray, next, inside, ck, RAY_GLOSS, nullptr);
        }
        reflection = scale(acc, 1.0 / (double)options.gloss_samples);
    } else {
        reflection = trace(cx, reflect_ray, next, inside, child_key(key, CODE_MIRROR), RAY_REFLECT, nullptr);
    }
    return scale(hit.material->global_specular(reflection), reflect_fresnel);
}

// Renderer::global_transmission, renderer.rs:170-186
Vec3 global_transmission(Ctx &cx, const Intersection &hit, Options options, bool inside, Vec3 i, Vec3 n,
                         double refract_fresnel, uint64_t key) {
    Vec3 t;
    double actual = refract_fresnel;
    if (!refract(i, n, hit.material->ior(), inside, &t)) {
        t = reflect(i, n);
        actual = 1.0;
    }
    Ray refract_ray = ray_new(hit.position + scale(t, EPSILON), t);
    Options next = options;
    next.refract_depth = options.refract_depth - 1;
    Vec3 refraction = trace(cx, refract_ray, next, !inside, child_key(key, CODE_TRANSMIT), RAY_REFRACT, nullptr);
    return scale(hit.material->global_transmissive(refraction), actual);
}

// Renderer::trace, renderer.rs:105-146
Vec3 trace(Ctx &cx, const Ray &ray, Options options, bool inside, uint64_t key, int ray_type, int *prim_id_out) {
    if (options.reflect_depth <= 0 || options.refract_depth <= 0) return v3(0, 0, 0);
    const Scene &scene = *cx.scene;
    cx.cn.rays[ray_type] += 1;
    Intersection hit;
    if (get_nearest_hit(scene, ray, &hit, &cx.cn)) {
        cx.cn.hits += 1;
        if (prim_id_out) *prim_id_out = hit.prim_id;
        Vec3 n = unit(hit.n);
        Vec3 i = unit(-ray.direction);
        Vec3 result = v3(0, 0, 0);
        for (size_t li = 0; li < scene.lights.size(); ++li) {
            const Light &light = scene.lights[li];
            Vec3 shadow = shadow_intensity(cx, hit, light, (uint32_t)li, options.shadow_samples, key);
            Vec3 l = unit(light.center() - hit.position);
            result = result + light.color() * hit.material->sample(n, i, l, hit.u, hit.v, &cx.cn) * shadow;
        }
        if (hit.material->is_reflective() || hit.material->is_refractive()) {
            double reflect_fresnel = fresnel_reflect(hit.material->ior(), i, n, inside);
            double refract_fresnel = 1.0 - reflect_fresnel;
            if (hit.material->is_reflective())
                result = result + global_reflection(cx, hit, options, inside, i, n, reflect_fresnel, key);
            if (hit.material->is_refractive())
                result = result + global_transmission(cx, hit, options, inside, i, n, refract_fresnel, key);
        }
        return result;
    }
    if (prim_id_out) *prim_id_out = -1;
    if (scene.has_skybox) return cubemap_color(scene.skybox, ray.direction, &cx.cn);
    return scene.background;
}

// ColorRGBA::new_rgb_clamped, compositor/colorrgba.rs:61-76: round half away from zero,
// saturating `as i32`, clamp to [0,255]
inline uint8_t quantize(double c) {
    int v = as_i32(std::round(c * 255.0));
    if (v > 255) v = 255;
    if (v < 0) v = 0;
    return (uint8_t)v;
}

// One pixel of Renderer::render_tile, renderer.rs:74-99
inline void render_pixel(Ctx &cx, const Camera &camera, Options options, uint32_t frame, uint64_t seed,
                         size_t abs_x, size_t abs_y, uint8_t rgba[4], int32_t *prim_id) {
    uint32_t pixel_samples = options.pixel_samples;
    double pixel_width = 1.0 / (double)pixel_samples;
    Vec3 color = v3(0, 0, 0);
    uint32_t pixel_index = (uint32_t)(abs_y * camera.image_width + abs_x);
    for (uint32_t y_subpixel = 0; y_subpixel < pixel_samples; ++y_subpixel) {
        for (uint32_t x_subpixel = 0; x_subpixel < pixel_samples; ++x_subpixel) {
            uint32_t sub = y_subpixel * pixel_samples + x_subpixel;
            uint64_t key = root_key(seed, frame, pixel_index, sub);
            double j_x = 0.0, j_y = 0.0;
            if (pixel_samples > 1) {
                double ua, ub;
                cx.rng.draw2(key, STREAM_JITTER, 0, 0, &ua, &ub);
                j_x = (double)x_subpixel * pixel_width + ua * pixel_width;
                j_y = (double)y_subpixel * pixel_width + ub * pixel_width;
            }
            Ray ray = camera_get_ray(camera, (double)abs_x + j_x, (double)abs_y + j_y);
            int pid = -1;
            Vec3 result = trace(cx, ray, options, false, key, RAY_PRIMARY, &pid);
            if (sub == 0 && prim_id) *prim_id = pid;
            color = color + scale(clampv(result), 1.0 / (double)(pixel_samples * pixel_samples));
        }
    }
    rgba[0] = quantize(color.x);
    rgba[1] = quantize(color.y);
    rgba[2] = quantize(color.z);
    rgba[3] = 255;
}

} // namespace

// ================================================================= C interface (ctypes)
extern "C" {

struct orc_counters {
    uint64_t rays[5];
    uint64_t b_node[2], b_leaf[2], p_tri[2], p_sph[2], p_pln[2], hits, mat_fetch, texels;
    double seconds;
    uint32_t threads;
    uint32_t pad;
};

struct orc_scene {
    Scene s;
};

int orc_scene_create(const rtx_scene_desc *d, orc_scene **out) {
    if (!d || !out) return 1;
    std::unique_ptr<orc_scene> h(new orc_scene());
    Scene &s = h->s;
    s.images.resize((size_t)d->n_images);
    for (int i = 0; i < d->n_images; ++i) {
        s.images[i].w = d->images[i].width;
        s.images[i].h = d->images[i].height;
        size_t nb = (size_t)d->images[i].width * (size_t)d->images[i].height * 4;
        s.images[i].rgba.assign(d->images[i].rgba, d->images[i].rgba + nb);
    }
    s.textures.resize((size_t)d->n_textures);
    for (int i = 0; i < d->n_textures; ++i) {
        const rtx_texture &t = d->textures[i];
        s.textures[i].kind = t.kind;
        s.textures[i].color1 = v3(t.color1[0], t.color1[1], t.color1[2]);
        s.textures[i].color2 = v3(t.color2[0], t.color2[1], t.color2[2]);
        s.textures[i].scale = t.scale;
        if (t.kind == RTX_TEX_IMAGE) {
            if (t.image < 0 || t.image >= d->n_images) return 1;
            s.textures[i].image = &s.images[(size_t)t.image];
        }
    }
    s.materials.resize((size_t)d->n_materials);
    for (int i = 0; i < d->n_materials; ++i) {
        s.materials[i].m = d->materials[i];
        int tx = d->materials[i].texture;
        if (tx >= d->n_textures) return 1;
        s.materials[i].tex = tx >= 0 ? &s.textures[(size_t)tx] : nullptr;
    }
    s.lights.resize((size_t)d->n_lights);
    for (int i = 0; i < d->n_lights; ++i) s.lights[i].l = d->lights[i];
    s.background = v3(d->background[0], d->background[1], d->background[2]);
    s.has_skybox = d->skybox_image[0] >= 0;
    for (int f = 0; f < 6; ++f) {
        s.skybox[f] = nullptr;
        if (s.has_skybox) {
            if (d->skybox_image[f] < 0 || d->skybox_image[f] >= d->n_images) return 1;
            s.skybox[f] = &s.images[(size_t)d->skybox_image[f]];
        }
    }
    int n_total = d->n_spheres + d->n_planes + d->n_triangles;
    if (d->order_kind && d->n_prims != n_total) return 1;
    s.prim_store.resize((size_t)n_total);
    std::vector<const Prim *> all;
    all.reserve((size_t)n_total);
    auto mat_at = [&](int m) -> const Material * { return (m >= 0 && m < d->n_materials) ? &s.materials[(size_t)m] : nullptr; };
    for (int k = 0; k < n_total; ++k) {
        int kind, idx;
        if (d->order_kind) { kind = d->order_kind[k]; idx = d->order_index[k]; }
        else if (k < d->n_spheres) { kind = RTX_PRIM_SPHERE; idx = k; }
        else if (k < d->n_spheres + d->n_triangles) { kind = RTX_PRIM_TRIANGLE; idx = k - d->n_spheres; }
        else { kind = RTX_PRIM_PLANE; idx = k - d->n_spheres - d->n_triangles; }
        Prim &p = s.prim_store[(size_t)k];
        p.kind = kind;
        if (kind == RTX_PRIM_SPHERE) {
            if (idx < 0 || idx >= d->n_spheres) return 1;
            p.center = v3(d->sphere_center[idx * 3], d->sphere_center[idx * 3 + 1], d->sphere_center[idx * 3 + 2]);
            p.radius = d->sphere_radius[idx];
            p.material = mat_at(d->sphere_material[idx]);
        } else if (kind == RTX_PRIM_PLANE) {
            if (idx < 0 || idx >= d->n_planes) return 1;
            p.a = d->plane_abcd[idx * 4]; p.b = d->plane_abcd[idx * 4 + 1];
            p.c = d->plane_abcd[idx * 4 + 2]; p.d = d->plane_abcd[idx * 4 + 3];
            p.material = mat_at(d->plane_material[idx]);
        } else if (kind == RTX_PRIM_TRIANGLE) {
            if (idx < 0 || idx >= d->n_triangles) return 1;
            for (int v = 0; v < 3; ++v) {
                const double *pv = d->tri_vertices + (size_t)idx * 9 + v * 3;
                const double *pn = d->tri_normals + (size_t)idx * 9 + v * 3;
                p.vertices[v] = v3(pv[0], pv[1], pv[2]);
                p.normals[v] = v3(pn[0], pn[1], pn[2]);
                p.tu[v] = d->tri_uv[(size_t)idx * 6 + v * 2];
                p.tv[v] = d->tri_uv[(size_t)idx * 6 + v * 2 + 1];
            }
            p.material = mat_at(d->tri_material[idx]);
        } else {
            return 1;
        }
        if (!p.material) return 1;
        all.push_back(&p);
    }
    octree_from_iter(s.octree, all);
    for (size_t i = 0; i < s.octree.prims.size(); ++i) const_cast<Prim *>(s.octree.prims[i])->prim_id = (int)i;
    for (size_t j = 0; j < s.octree.infinites.size(); ++j)
        const_cast<Prim *>(s.octree.infinites[j])->prim_id = (int)(s.octree.prims.size() + j);
    *out = h.release();
    return 0;
}

void orc_scene_destroy(orc_scene *h) { delete h; }

static void count_nodes(const OctreeNode &n, int64_t *nodes, int64_t *interior, int64_t *entries, int64_t *max_entries) {
    *nodes += 1;
    if (!n.children.empty()) *interior += 1;
    *entries += (int64_t)n.leaf_data.size();
    if ((int64_t)n.leaf_data.size() > *max_entries) *max_entries = (int64_t)n.leaf_data.size();
    for (const auto &c : n.children) count_nodes(c, nodes, interior, entries, max_entries);
}
// out[0..5] = depth (clamped at i32), n_finite, n_infinite, n_nodes, n_interior, n_leaf_entries, [6] = max entries
void orc_octree_info(const orc_scene *h, int64_t *out) {
    int64_t nodes = 0, interior = 0, entries = 0, mx = 0;
    count_nodes(h->s.octree.root, &nodes, &interior, &entries, &mx);
    out[0] = h->s.octree.depth; out[1] = (int64_t)h->s.octree.prims.size(); out[2] = (int64_t)h->s.octree.infinites.size();
    out[3] = nodes; out[4] = interior; out[5] = entries; out[6] = mx;
}
// DFS pre-order dump of the octree (children 0..7), for comparing the product's linearised
// tree with the reference-order structure: per node 6 doubles bbox + (n_children, n_entries),
// then the entries' prim indices.  Returns counts via the in/out sizes.
void orc_octree_dump(const orc_scene *h, double *bbox_out, int32_t *shape_out, int32_t *entries_out) {
    struct W {
        static void go(const OctreeNode &n, double *&b, int32_t *&s, int32_t *&e) {
            *b++ = n.bbox.min.x; *b++ = n.bbox.min.y; *b++ = n.bbox.min.z;
            *b++ = n.bbox.max.x; *b++ = n.bbox.max.y; *b++ = n.bbox.max.z;
            *s++ = (int32_t)n.children.size(); *s++ = (int32_t)n.leaf_data.size();
            for (const auto &d : n.leaf_data) *e++ = (int32_t)d.index;
            for (const auto &c : n.children) go(c, b, s, e);
        }
    };
    W::go(h->s.octree.root, bbox_out, shape_out, entries_out);
}

// Renderer::render, renderer.rs:30-65: full 128x8 tiles row-major (edge tiles over-render,
// merge clips: surface.rs:83-94), a pool of `threads` workers.  rows [row_begin,row_end) x
// columns [col_begin,col_end) restrict the tiles rendered (whole frame: 0,H,0,W) so a bounded
// sample of a big frame can be timed; shard_index/count pick 8-row blocks like the product.
int orc_render(const orc_scene *h, const rtx_camera *cam, const rtx_options *opt, uint32_t threads,
               uint32_t row_begin, uint32_t row_end, uint32_t col_begin, uint32_t col_end,
               uint8_t *rgba_out, int32_t *prim_id_out, orc_counters *counters) {
    if (!h || !cam || !opt || !rgba_out) return 1;
    if (cam->image_height % 8 != 0) return 1; // renderer.rs:73 underflows otherwise
    const Scene &scene = h->s;
    Camera camera = camera_new(*cam);
    Options options{opt->reflect_depth, opt->refract_depth, opt->shadow_samples, opt->gloss_samples, opt->pixel_samples};
    size_t W = cam->image_width, H = cam->image_height;
    if (row_end > H) row_end = (uint32_t)H;
    if (col_end > W) col_end = (uint32_t)W;
    uint32_t shard_count = opt->shard_count ? opt->shard_count : 1;
    struct Tile { size_t x_off, y_off; };
    std::vector<Tile> tiles;                                       // SubsurfaceIterator, surface.rs:146-179
    for (size_t y_off = 0; y_off < H; y_off += 8)
        for (size_t x_off = 0; x_off < W; x_off += 128) {
            if (y_off + 8 <= row_begin || y_off >= row_end) continue;
            if (x_off + 128 <= col_begin || x_off >= col_end) continue;
            if ((y_off / 8) % shard_count != opt->shard_index) continue;
            tiles.push_back(Tile{x_off, y_off});
        }
    if (threads == 0) threads = std::thread::hardware_concurrency();   // num_cpus::get(), main.rs:138
    if (threads == 0) threads = 1;
    std::atomic<size_t> next(0);
    std::vector<Counters> per_thread(threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](uint32_t tid) {
        Ctx cx{&scene, Rng{opt->rng_mode}, Counters()};
        for (;;) {
            size_t ti = next.fetch_add(1);
            if (ti >= tiles.size()) break;
            const Tile &tile = tiles[ti];
            for (size_t rel_y = 0; rel_y < 8; ++rel_y) {           // render_tile, renderer.rs:72-100
                size_t abs_y = H - (tile.y_off + rel_y) - 1;
                for (size_t rel_x = 0; rel_x < 128; ++rel_x) {
                    size_t abs_x = tile.x_off + rel_x;
                    uint8_t px[4];
                    int32_t pid = -1;
                    // over-rendered pixels past the right edge are traced and dropped by merge
                    // in the reference; tracing them has no observable effect, so skip them.
                    if (abs_x >= W) continue;
                    render_pixel(cx, camera, options, opt->frame, opt->seed, abs_x, abs_y, px, &pid);
                    size_t dst = (tile.y_off + rel_y) * W + abs_x; // Surface::merge, surface.rs:83-94
                    std::memcpy(rgba_out + dst * 4, px, 4);
                    if (prim_id_out) prim_id_out[dst] = pid;
                }
            }
        }
        per_thread[tid] = cx.cn;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (counters) {
        Counters total;
        for (auto &c : per_thread) total.add(c);
        std::memcpy(counters->rays, total.rays, sizeof(total.rays));
        for (int c = 0; c < 2; ++c) {
            counters->b_node[c] = total.b_node[c]; counters->b_leaf[c] = total.b_leaf[c];
            counters->p_tri[c] = total.p_tri[c]; counters->p_sph[c] = total.p_sph[c]; counters->p_pln[c] = total.p_pln[c];
        }
        counters->hits = total.hits; counters->mat_fetch = total.mat_fetch; counters->texels = total.texels;
        counters->seconds = std::chrono::duration<double>(t1 - t0).count();
        counters->threads = threads;
        counters->pad = 0;
    }
    return 0;
}

// Linear (float) per-sample colours before clamp/quantise, for RMSE-style comparisons:
// out[pixel][3] = sum over samples of clamp(result)/spp^2 (renderer.rs:95), row 0 = top.
int orc_render_linear(const orc_scene *h, const rtx_camera *cam, const rtx_options *opt, uint32_t threads, double *out) {
    if (!h || !cam || !opt || !out) return 1;
    const Scene &scene = h->s;
    Camera camera = camera_new(*cam);
    Options options{opt->reflect_depth, opt->refract_depth, opt->shadow_samples, opt->gloss_samples, opt->pixel_samples};
    size_t W = cam->image_width, H = cam->image_height;
    if (threads == 0) threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        Ctx cx{&scene, Rng{opt->rng_mode}, Counters()};
        for (;;) {
            size_t row = next.fetch_add(1);
            if (row >= H) break;
            size_t abs_y = H - row - 1;
            for (size_t abs_x = 0; abs_x < W; ++abs_x) {
                uint32_t ps = options.pixel_samples;
                double pw = 1.0 / (double)ps;
                Vec3 color = v3(0, 0, 0);
                uint32_t pixel_index = (uint32_t)(abs_y * W + abs_x);
                for (uint32_t ys = 0; ys < ps; ++ys)
                    for (uint32_t xs = 0; xs < ps; ++xs) {
                        uint32_t sub = ys * ps + xs;
                        uint64_t key = root_key(opt->seed, opt->frame, pixel_index, sub);
                        double jx = 0, jy = 0;
                        if (ps > 1) {
                            double ua, ub;
                            cx.rng.draw2(key, STREAM_JITTER, 0, 0, &ua, &ub);
                            jx = (double)xs * pw + ua * pw;
                            jy = (double)ys * pw + ub * pw;
                        }
                        Ray ray = camera_get_ray(camera, (double)abs_x + jx, (double)abs_y + jy);
                        Vec3 r = trace(cx, ray, options, false, key, RAY_PRIMARY, nullptr);
                        color = color + scale(clampv(r), 1.0 / (double)(ps * ps));
                    }
                out[(row * W + abs_x) * 3 + 0] = color.x;
                out[(row * W + abs_x) * 3 + 1] = color.y;
                out[(row * W + abs_x) * 3 + 2] = color.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}

// ---- probes for known-answer tests ------------------------------------------------------
static Ray mk_ray(const double *o, const double *d) { return ray_new(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])); }

int orc_bbox_intersects(const double *bmin, const double *bmax, const double *o, const double *d) {
    BBox b{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    return bbox_intersects(b, mk_ray(o, d)) ? 1 : 0;
}
int orc_bbox_overlaps(const double *amin, const double *amax, const double *bmin, const double *bmax) {
    BBox a{v3(amin[0], amin[1], amin[2]), v3(amax[0], amax[1], amax[2])};
    BBox b{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    return bbox_overlaps(a, b) ? 1 : 0;
}
// out = n[3], t, u, v, position[3]  (9 doubles); returns 1 for Some, 0 for None
static int emit(bool ok, const Intersection &is, double *out) {
    if (!ok) return 0;
    out[0] = is.n.x; out[1] = is.n.y; out[2] = is.n.z; out[3] = is.t; out[4] = is.u; out[5] = is.v;
    out[6] = is.position.x; out[7] = is.position.y; out[8] = is.position.z;
    return 1;
}
int orc_triangle_intersects(const double *v, const double *n, const double *uv, const double *o, const double *d,
                            double t_min, double t_max, double *out) {
    Prim p;
    p.kind = RTX_PRIM_TRIANGLE;
    for (int k = 0; k < 3; ++k) {
        p.vertices[k] = v3(v[k * 3], v[k * 3 + 1], v[k * 3 + 2]);
        p.normals[k] = v3(n[k * 3], n[k * 3 + 1], n[k * 3 + 2]);
        p.tu[k] = uv[k * 2]; p.tv[k] = uv[k * 2 + 1];
    }
    Intersection is;
    return emit(p.intersects(mk_ray(o, d), t_min, t_max, &is, nullptr), is, out);
}
void orc_triangle_auto_normal(const double *v, double *out) {     // triangle.rs:39-42
    Vec3 n = cross(v3(v[3], v[4], v[5]) - v3(v[0], v[1], v[2]), v3(v[6], v[7], v[8]) - v3(v[0], v[1], v[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
int orc_sphere_intersects(const double *center, double radius, const double *o, const double *d, double t_min, double t_max, double *out) {
    Prim p;
    p.kind = RTX_PRIM_SPHERE; p.center = v3(center[0], center[1], center[2]); p.radius = radius;
    Intersection is;
    return emit(p.intersects(mk_ray(o, d), t_min, t_max, &is, nullptr), is, out);
}
int orc_plane_intersects(const double *abcd, const double *o, const double *d, double t_min, double t_max, double *out) {
    Prim p;
    p.kind = RTX_PRIM_PLANE; p.a = abcd[0]; p.b = abcd[1]; p.c = abcd[2]; p.d = abcd[3];
    Intersection is;
    return emit(p.intersects(mk_ray(o, d), t_min, t_max, &is, nullptr), is, out);
}
// Ray::get_nearest_hit for a batch; prim_id -1 / t = +inf for None; hit_out (nullable) 9 doubles per ray
void orc_nearest_hits(const orc_scene *h, int64_t n, const double *o, const double *d, int32_t *prim_id, double *t_out, double *hit_out) {
    for (int64_t i = 0; i < n; ++i) {
        Intersection is;
        bool ok = get_nearest_hit(h->s, mk_ray(o + i * 3, d + i * 3), &is, nullptr);
        prim_id[i] = ok ? is.prim_id : -1;
        t_out[i] = ok ? is.t : INF;
        if (hit_out && ok) emit(true, is, hit_out + i * 9);
    }
}
// the candidate fold of shadow_intensity for a batch of (origin, target) pairs
void orc_shadow_products(const orc_scene *h, int64_t n, const double *origin, const double *target, double *out) {
    for (int64_t i = 0; i < n; ++i) {
        Vec3 p = v3(origin[i * 3], origin[i * 3 + 1], origin[i * 3 + 2]);
        Vec3 s = v3(target[i * 3], target[i * 3 + 1], target[i * 3 + 2]);
        Vec3 l = unit(s - p);
        Ray r = ray_new(p, l);
        double dist = len(s - p);
        Vec3 acc = v3(1, 1, 1);
        for_each_candidate(h->s.octree, r, nullptr, [&](const Prim *prim) {
            Intersection occ;
            if (prim->intersects(r, EPSILON, dist, &occ, nullptr)) acc = acc * occ.material->transmission();
        });
        out[i * 3] = acc.x; out[i * 3 + 1] = acc.y; out[i * 3 + 2] = acc.z;
    }
}
double orc_fresnel_reflect(double ior, const double *i, const double *n, int inside) {
    return fresnel_reflect(ior, v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), inside != 0);
}
int orc_refract(const double *v, const double *n, double ior, int inside, double *out) {
    Vec3 t;
    if (!refract(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), ior, inside != 0, &t)) return 0;
    out[0] = t.x; out[1] = t.y; out[2] = t.z;
    return 1;
}
void orc_reflect(const double *v, const double *n, double *out) {
    Vec3 r = reflect(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_material_sample(const orc_scene *h, int material, const double *n, const double *i, const double *l, double u, double v, double *out) {
    Vec3 c = h->s.materials[(size_t)material].sample(v3(n[0], n[1], n[2]), v3(i[0], i[1], i[2]), v3(l[0], l[1], l[2]), u, v, nullptr);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void orc_texture_color(const orc_scene *h, int texture, double u, double v, double *out) {
    Vec3 c = texture_color(h->s.textures[(size_t)texture], u, v, nullptr);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void orc_image_color(const uint8_t *rgba, int w, int hgt, double u, double v, double *out4) {
    Image im;
    im.w = w; im.h = hgt; im.rgba.assign(rgba, rgba + (size_t)w * hgt * 4);
    Color4 c = image_color(im, u, v, nullptr);
    out4[0] = c.r; out4[1] = c.g; out4[2] = c.b; out4[3] = c.a;
}
int orc_skybox_color(const orc_scene *h, const double *dir, double *out) {
    if (!h->s.has_skybox) return 0;
    Vec3 c = cubemap_color(h->s.skybox, v3(dir[0], dir[1], dir[2]), nullptr);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    return 1;
}
// out = origin[3], direction[3], inverse_dir[3], signs[3] as 0/1 doubles
void orc_camera_get_ray(const rtx_camera *cam, double x, double y, double *out) {
    Camera c = camera_new(*cam);
    Ray r = camera_get_ray(c, x, y);
    out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
    out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
    out[6] = r.inverse_dir.x; out[7] = r.inverse_dir.y; out[8] = r.inverse_dir.z;
    out[9] = r.signs[0]; out[10] = r.signs[1]; out[11] = r.signs[2];
}
// out = eye[3], right[3], half_width, half_height, pixel_width, pixel_height
void orc_camera_basis(const rtx_camera *cam, double *out) {
    Camera c = camera_new(*cam);
    out[0] = c.eye.x; out[1] = c.eye.y; out[2] = c.eye.z; out[3] = c.right.x; out[4] = c.right.y; out[5] = c.right.z;
    out[6] = c.half_width; out[7] = c.half_height; out[8] = c.pixel_width; out[9] = c.pixel_height;
}
int orc_quantize(double c) { return quantize(c); }
void orc_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }
uint64_t orc_root_key(uint64_t seed, uint32_t frame, uint32_t pixel, uint32_t sub) { return root_key(seed, frame, pixel, sub); }
uint64_t orc_child_key(uint64_t parent, uint32_t code) { return child_key(parent, code); }
void orc_draw2(uint32_t mode, uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *out2) {
    Rng r{mode};
    r.draw2(key, stream, i, j, &out2[0], &out2[1]);
}
void orc_draw3(uint32_t mode, uint64_t key, uint32_t stream, uint32_t i, uint32_t j, double *out3) {
    Rng r{mode};
    r.draw3(key, stream, i, j, &out3[0], &out3[1], &out3[2]);
}
double orc_epsilon(void) { return EPSILON; }

} // extern "C"
