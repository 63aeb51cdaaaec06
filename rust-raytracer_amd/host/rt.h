// rt.h — host-side mirror (C++) of the types gyng/rust-raytracer scenes are written against:
// Vec3, Mat4/Transform, Prim (Sphere, Plane, Triangle via TriangleOptions, Mesh), Material
// (CookTorrance, Phong, Flat), Texture (Checker, UV, Image, CubeMap), Light (Point, Sphere),
// Camera (+ keyframes, easing), Scene, RenderOptions, Renderer, Animator, Surface.
// Same names, same argument meaning, same defaults; the tracing itself happens behind the
// C ABI of include/rtx.h (Renderer::render flattens the scene and calls rtx_render).
//
// The Rust toolchain is absent in this pipeline, so this C++ layer stands where the Rust crate
// would keep its own types and bind the C ABI through `extern "C"`.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtx.h"

namespace rt {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double F64_EPSILON = 2.220446049250313e-16;

// ------------------------------------------------------------------ src/vec3.rs
struct Vec3 {
    double x = 0, y = 0, z = 0;
    static Vec3 zero() { return {0, 0, 0}; }
    static Vec3 one() { return {1, 1, 1}; }
    double len() const { return std::sqrt(x * x + y * y + z * z); }
    double dot(const Vec3 &o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3 &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    Vec3 unit() const { double l = len(); return {x / l, y / l, z / l}; }
    Vec3 scale(double s) const { return {x * s, y * s, z * s}; }
    static Vec3 lerp(const Vec3 &a, const Vec3 &b, double alpha) {            // vec3.rs:94-100
        return {a.x + (b.x - a.x) * alpha, a.y + (b.y - a.y) * alpha, a.z + (b.z - a.z) * alpha};
    }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// ------------------------------------------------------------------ src/mat4.rs (row-major)
struct Mat4 {
    double m[4][4];
    static Mat4 identity();
    static Mat4 zero();
    static Mat4 translate_matrix(const Vec3 &v);
    static Mat4 scale_matrix(const Vec3 &v);
    static Mat4 rotate_x_deg_matrix(double angle);
    static Mat4 rotate_y_deg_matrix(double angle);
    static Mat4 rotate_z_deg_matrix(double angle);
    static Mat4 rotate_axis_deg_matrix(double angle, const Vec3 &axis);
    static Mat4 mult_m(const Mat4 &a, const Mat4 &b);
    static Vec3 mult_v(const Mat4 &m, const Vec3 &v);
    static Vec3 mult_p(const Mat4 &m, const Vec3 &p);
    static Vec3 transform_normal(const Vec3 &n, const Mat4 &transform);
    Mat4 inverse() const;
    Mat4 transpose() const;
    bool has_scale() const;
    double scale() const { return m[0][0]; }
};
inline Mat4 operator*(const Mat4 &a, const Mat4 &b) { return Mat4::mult_m(a, b); }
struct Transform {                                                       // mat4.rs:20-32
    Mat4 m, inv;
    explicit Transform(const Mat4 &mat) : m(mat), inv(mat.inverse()) {}
};

// ------------------------------------------------------------------ compositor: Surface (RGBA8)
struct ColorRGBA8 {
    uint8_t r = 0, g = 0, b = 0, a = 255;
};
struct Surface {                                                         // compositor/surface.rs:18-25
    size_t width = 0, height = 0;
    std::vector<ColorRGBA8> buffer;
    Surface() = default;
    Surface(size_t w, size_t h, ColorRGBA8 bg = {}) : width(w), height(h), buffer(w * h, bg) {}
    ColorRGBA8 &at(size_t x, size_t y) { return buffer[width * y + x]; }
    const ColorRGBA8 &at(size_t x, size_t y) const { return buffer[width * y + x]; }
};

// ------------------------------------------------------------------ src/material/textures
struct ColorF {
    double r = 0, g = 0, b = 0, a = 1;
    static ColorF white() { return {1, 1, 1, 1}; }
    static ColorF black() { return {0, 0, 0, 1}; }
    static ColorF new_rgb(double r, double g, double b) { return {r, g, b, 1}; }
};
struct Texture {
    enum Kind { Checker, UV, Image } kind = Checker;
    ColorF color1, color2;                                               // CheckerTexture
    double scale = 1.0;
    std::shared_ptr<const Surface> image;                                // ImageTexture
};
using TexturePtr = std::shared_ptr<const Texture>;
TexturePtr CheckerTexture(ColorF color1, ColorF color2, double scale);
TexturePtr UVTexture();
TexturePtr ImageTexture(std::shared_ptr<const Surface> image);
struct CubeMap {                                                         // cubemap.rs:7-9
    std::shared_ptr<const Surface> faces[6];                             // left,right,down,up,front,back
    // CubeMap::load(x, x_neg, y, y_neg, z, z_neg): PNG files (cubemap.rs:14-42)
    static std::shared_ptr<CubeMap> load(const std::string &x, const std::string &x_neg, const std::string &y,
                                         const std::string &y_neg, const std::string &z, const std::string &z_neg);
};

// ------------------------------------------------------------------ src/material/materials
struct Material {
    enum Kind { CookTorrance = RTX_MAT_COOK_TORRANCE, Phong = RTX_MAT_PHONG, Flat = RTX_MAT_FLAT } kind = CookTorrance;
    double k_a = 0.0, k_d = 1.0, k_s = 1.0, k_sg = 0.0, k_tg = 0.0;
    Vec3 ambient = Vec3::one(), diffuse{0.5, 0.5, 0.5}, transmission = Vec3::zero(), specular = Vec3::one();
    double roughness = 0.15;      // CookTorrance
    double shininess = 10.0;      // Phong
    double glossiness = 0.0;
    double gauss_constant = 1.0;  // CookTorrance
    double ior = 1.5;
    TexturePtr diffuse_texture;
    Vec3 color{0.5, 0.5, 0.5};    // Flat
};
using MaterialPtr = std::shared_ptr<const Material>;
Material CookTorranceMaterial();   // Default: cooktorrancematerial.rs:93-111
Material PhongMaterial();          // Default: phongmaterial.rs:71-88 (ior 1.0)
Material FlatMaterial(Vec3 color = {0.5, 0.5, 0.5});
inline MaterialPtr boxed(const Material &m) { return std::make_shared<const Material>(m); }

// ------------------------------------------------------------------ src/geometry
struct Prim {
    enum Kind { Sphere = RTX_PRIM_SPHERE, Plane = RTX_PRIM_PLANE, Triangle = RTX_PRIM_TRIANGLE } kind;
    MaterialPtr material;
    Vec3 center; double radius = 0;                                      // Sphere (sphere.rs:12-16)
    double a = 0, b = 0, c = 0, d = 0;                                   // Plane (plane.rs:11-17)
    Vec3 vertices[3], normals[3]; double tu[3] = {0.5, 0.0, 1.0}, tv[3] = {1.0, 0.0, 0.0};   // Triangle
    void mut_transform(const Transform &t);                              // prim.rs:9, per-kind impls
};
Prim Sphere(Vec3 center, double radius, MaterialPtr material);
Prim Plane(double a, double b, double c, double d, MaterialPtr material);
struct TriangleOptions {                                                 // triangle.rs:32-87
    Vec3 v[3];
    bool has_normals = false, has_texinfo = false;
    Vec3 n[3];
    double tu[3], tv[3];
    MaterialPtr mat;
    TriangleOptions(Vec3 v0, Vec3 v1, Vec3 v2) : v{v0, v1, v2} {}
    TriangleOptions &normals(Vec3 n0, Vec3 n1, Vec3 n2) { n[0] = n0; n[1] = n1; n[2] = n2; has_normals = true; return *this; }
    TriangleOptions &texinfo(double u0, double v0, double u1, double v1, double u2, double v2) {
        tu[0] = u0; tv[0] = v0; tu[1] = u1; tv[1] = v1; tu[2] = u2; tv[2] = v2; has_texinfo = true; return *this;
    }
    TriangleOptions &material(MaterialPtr m) { mat = std::move(m); return *this; }
    Prim build() const;
};
struct Mesh {                                                            // mesh.rs:5-15
    std::vector<Prim> triangles;
    void mut_transform(const Transform &t) { for (auto &tr : triangles) tr.mut_transform(t); }
};

// ------------------------------------------------------------------ src/light
struct Light {
    enum Kind { Point = RTX_LIGHT_POINT, Sphere = RTX_LIGHT_SPHERE } kind;
    Vec3 position, color;
    double radius = 0;
};
inline Light PointLight(Vec3 position, Vec3 color) { return {Light::Point, position, color, 0.0}; }
inline Light SphereLight(Vec3 position, Vec3 color, double radius) { return {Light::Sphere, position, color, radius}; }

// ------------------------------------------------------------------ raytracer/animator
struct Easing {                                                          // easing.rs
    double a = 0.0, b = 1.0 / 3.0, c = 2.0 / 3.0, d = 1.0;
    static Easing linear() { return Easing(); }
    double t(double t) const;
};
struct CameraKeyframe {                                                  // camerakeyframe.rs
    double time;
    Vec3 position, look_at, up;
    Easing easing;
};

// ------------------------------------------------------------------ src/scene
struct Camera {                                                          // camera.rs:7-23
    Vec3 position, look_at, up;
    double fov_deg = 45;
    uint32_t image_width = 0, image_height = 0;
    Vec3 eye, right;
    double half_width = 0, half_height = 0, pixel_width = 0, pixel_height = 0;
    std::vector<CameraKeyframe> keyframes;
    bool has_keyframes = false;
    static Camera create(Vec3 position, Vec3 look_at, Vec3 up, double fov_deg, uint32_t w, uint32_t h);   // Camera::new
    static Camera new_with_keyframes(Vec3 position, Vec3 look_at, Vec3 up, double fov_deg, uint32_t w, uint32_t h,
                                     std::vector<CameraKeyframe> keyframes);
    void insert_keyframes(std::vector<CameraKeyframe> additional);
    rtx_camera abi() const;
};

struct DeviceScene;   // owns the rtx_scene handle
struct Scene {                                                           // scene.rs:7-12
    std::vector<Light> lights;
    std::vector<Prim> prims;           // insertion order; Octree::from_iter happens inside rtx_scene_create
    Vec3 background;
    std::shared_ptr<const CubeMap> skybox;
    mutable std::shared_ptr<DeviceScene> device;   // uploaded on first render (the Arc<Scene> is immutable)
};

// The flattened scene: owns every array an rtx_scene_desc points into.
struct FlatScene {
    std::vector<double> sphere_center, sphere_radius, plane_abcd, tri_vertices, tri_normals, tri_uv;
    std::vector<int32_t> sphere_material, plane_material, tri_material, order_kind, order_index;
    std::vector<rtx_material> materials;
    std::vector<rtx_texture> textures;
    std::vector<rtx_image> images;
    std::vector<std::shared_ptr<const Surface>> image_refs;
    std::vector<rtx_light> lights;
    rtx_scene_desc desc{};
};
std::unique_ptr<FlatScene> flatten(const Scene &scene);

struct RenderOptions {                                                   // renderer.rs:14-21
    uint32_t reflect_depth = 1, refract_depth = 1, shadow_samples = 1, gloss_samples = 1, pixel_samples = 1;
};
struct RenderStats {
    rtx_stats last{};
    uint64_t rays_total = 0;
    double ms_device_total = 0;
};
struct Renderer {                                                        // renderer.rs:24-27
    size_t tasks = 0;                 // kept for API parity; the GPU path has no thread pool
    RenderOptions options;
    // what the reference leaves to thread_rng()/num_cpus: optional, default to the reference's behaviour
    uint32_t rng_mode = RTX_RNG_PHILOX;
    uint64_t seed = 0;
    int device = 0;
    uint32_t shard_index = 0, shard_count = 1;
    mutable RenderStats stats;
    // Renderer::render(&self, camera, shared_scene) -> Surface   (renderer.rs:30). Throws std::runtime_error
    // with the rtx error text where the reference would panic.
    Surface render(const Camera &camera, const std::shared_ptr<const Scene> &scene, uint32_t frame = 0) const;
};
struct Animator {                                                        // animator.rs:9-15
    double fps = 25, animate_from = 0, animate_to = 0;
    uint32_t starting_frame_number = 0;
    Renderer renderer;
    uint32_t frame_shard_index = 0, frame_shard_count = 1;   // frame k is rendered iff k % count == index
    void animate(const Camera &camera, const std::shared_ptr<const Scene> &scene, const std::string &filename) const;
    static void get_neighbour_keyframes(const std::vector<CameraKeyframe> &keyframes, double time, CameraKeyframe *first,
                                        CameraKeyframe *second, double *alpha);
    static Camera lerp_camera(const Camera &camera, double time);
    uint32_t total_frames() const { return (uint32_t)std::floor(fps * (animate_to - animate_from)); }
    double frame_time(uint32_t frame_number) const { return animate_from + (double)frame_number / fps; }
};

// ------------------------------------------------------------------ src/util
namespace util {
// util/import.rs:12-128 — v / vt / vn / f (first three vertices of a face), CookTorrance only
bool from_obj(const Material &material, bool flip_normals, const std::string &filename, Mesh *out, std::string *err);
// util/import.rs:130-145 — 8-bit PNG (gray, gray+alpha, RGB, RGBA, palette), non-interlaced -> RGBA8
bool from_image(const std::string &path, Surface *out, std::string *err);
// util/export.rs:5-18 — "P3 W H 255\n" then "r g b " per pixel
bool to_ppm(const Surface &surface, const std::string &filename, std::string *err);
std::string ppm_bytes(const Surface &surface);
} // namespace util

// ------------------------------------------------------------------ src/my_scene
struct SceneConfig {                                                     // my_scene/mod.rs:17-25
    virtual ~SceneConfig() {}
    virtual Camera get_camera(uint32_t w, uint32_t h, double fov) const = 0;
    virtual Camera get_animation_camera(uint32_t w, uint32_t h, double fov) const { return get_camera(w, h, fov); }
    virtual std::shared_ptr<Scene> get_scene() const = 0;
};
// my_scene/mod.rs:27-45; nullptr for an unknown name.  asset_root is where ./docs/assets of the
// reference would be; when an OBJ / skybox PNG is missing a seeded procedural stand-in of
// BASELINE.json's size is generated instead (and *used_standin set).
std::unique_ptr<SceneConfig> scene_by_name(const std::string &name, const std::string &asset_root = ".", double detail = 1.0);

// ------------------------------------------------------------------ src/main.rs
struct JobConfig {                                                       // SceneConfig in main.rs:34-49 (13 mandatory fields)
    std::string name, output_file;
    uint32_t size[2] = {0, 0};
    double fov = 0, fps = 0, time_slice[2] = {0, 0};
    uint32_t reflect_depth = 0, refract_depth = 0, shadow_samples = 0, gloss_samples = 0, pixel_samples = 0;
    bool animating = false;
    uint32_t starting_frame_number = 0;
    // optional extras of this build (all default so sample-config.json runs unmodified)
    std::string rng_mode = "philox";
    uint64_t seed = 0;
    int device = 0;
    std::string asset_root = ".";
};
// returns "" on success, else the reference's message ("parse failure, missing field ``x''")
std::string parse_job_config(const std::string &json_text, JobConfig *out);

} // namespace rt
