// rt.cpp — implementation of the host-side mirror declared in rt.h.
#include "rt.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <map>
#include <sstream>
#include <stdexcept>

namespace rt {

// ====================================================================== Mat4 (src/mat4.rs)
static Mat4 mat4_new(double t00, double t01, double t02, double t03, double t10, double t11, double t12, double t13,
                     double t20, double t21, double t22, double t23, double t30, double t31, double t32, double t33) {
    Mat4 r;
    double v[4][4] = {{t00, t01, t02, t03}, {t10, t11, t12, t13}, {t20, t21, t22, t23}, {t30, t31, t32, t33}};
    std::memcpy(r.m, v, sizeof(v));
    return r;
}
Mat4 Mat4::identity() { return mat4_new(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); }
Mat4 Mat4::zero() { return mat4_new(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); }
Mat4 Mat4::translate_matrix(const Vec3 &v) { return mat4_new(1, 0, 0, v.x, 0, 1, 0, v.y, 0, 0, 1, v.z, 0, 0, 0, 1); }   // :100
Mat4 Mat4::scale_matrix(const Vec3 &v) { return mat4_new(v.x, 0, 0, 0, 0, v.y, 0, 0, 0, 0, v.z, 0, 0, 0, 0, 1); }       // :109
static double deg_to_rad(double deg) { return deg * PI / 180.0; }                                                       // :342
Mat4 Mat4::rotate_x_deg_matrix(double angle) {                                                                          // :128
    double s = std::sin(deg_to_rad(angle)), c = std::cos(deg_to_rad(angle));
    return mat4_new(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
}
Mat4 Mat4::rotate_y_deg_matrix(double angle) {                                                                          // :140
    double s = std::sin(deg_to_rad(angle)), c = std::cos(deg_to_rad(angle));
    return mat4_new(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
}
Mat4 Mat4::rotate_z_deg_matrix(double angle) {                                                                          // :152
    double s = std::sin(deg_to_rad(angle)), c = std::cos(deg_to_rad(angle));
    return mat4_new(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
}
Mat4 Mat4::rotate_axis_deg_matrix(double angle, const Vec3 &axis) {                                                     // :164
    Vec3 a = axis.unit();
    double s = std::sin(deg_to_rad(angle)), c = std::cos(deg_to_rad(angle));
    return mat4_new(a.x * a.x + (1.0 - a.x * a.x) * c, a.x * a.y * (1.0 - c) - a.z * s, a.x * a.z * (1.0 - c) + a.y * s, 0.0,
                    a.x * a.y * (1.0 - c) + a.z * s, a.y * a.y + (1.0 - a.y * a.y) * c, a.y * a.z * (1.0 - c) - a.x * s, 0.0,
                    a.x * a.z * (1.0 - c) - a.y * s, a.y * a.z * (1.0 - c) + a.x * s, a.z * a.z + (1.0 - a.z * a.z) * c, 0.0,
                    0.0, 0.0, 0.0, 1.0);
}
Mat4 Mat4::transpose() const {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
    return r;
}
bool Mat4::has_scale() const {                                                                                          // :118
    auto approx_eq = [](double f1, double f2) { return std::fabs(f1 - f2) < F64_EPSILON; };
    return approx_eq(m[0][0], m[1][1]) && approx_eq(m[0][0], m[2][2]);
}
Mat4 Mat4::inverse() const {                                                                                            // :239-284
    const double(*a)[4] = m;
    double s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1];
    double s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
    double s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
    double s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
    double s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3];
    double s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
    double c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
    double c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    double c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    double c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    double c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
    double c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    double invdet = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    Mat4 r;
    r.m[0][0] = (a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * invdet;
    r.m[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * invdet;
    r.m[0][2] = (a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * invdet;
    r.m[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * invdet;
    r.m[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * invdet;
    r.m[1][1] = (a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * invdet;
    r.m[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * invdet;
    r.m[1][3] = (a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * invdet;
    r.m[2][0] = (a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * invdet;
    r.m[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * invdet;
    r.m[2][2] = (a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * invdet;
    r.m[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * invdet;
    r.m[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * invdet;
    r.m[3][1] = (a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * invdet;
    r.m[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * invdet;
    r.m[3][3] = (a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * invdet;
    return r;
}
Mat4 Mat4::mult_m(const Mat4 &a, const Mat4 &b) {                                                                       // :288-305
    Mat4 out = Mat4::zero();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k) out.m[i][j] += a.m[i][k] * b.m[k][j];
    return out;
}
Vec3 Mat4::mult_v(const Mat4 &m, const Vec3 &v) {                                                                       // :307
    return {m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
            m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z};
}
static bool are_equal_rel(double a, double b) {                                                                         // :34-40
    double mx = a;
    if (mx < b) mx = b;
    return std::fabs(a - b) <= F64_EPSILON * mx;
}
Vec3 Mat4::mult_p(const Mat4 &m, const Vec3 &p) {                                                                       // :315-336
    double xp = m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3];
    double yp = m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3];
    double zp = m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3];
    double wp = m.m[3][0] * p.x + m.m[3][1] * p.y + m.m[3][2] * p.z + m.m[3][3];
    if (are_equal_rel(wp, 1.0)) return {xp, yp, zp};
    return {xp / wp, yp / wp, zp / wp};
}
Vec3 Mat4::transform_normal(const Vec3 &n, const Mat4 &transform) {                                                     // :221-229
    Mat4 inv = transform.inverse();
    return {inv.m[0][0] * n.x + inv.m[1][0] * n.y + inv.m[2][0] * n.z, inv.m[0][1] * n.x + inv.m[1][1] * n.y + inv.m[2][1] * n.z,
            inv.m[0][2] * n.x + inv.m[1][2] * n.y + inv.m[2][2] * n.z};
}

// ====================================================================== textures / materials / prims
TexturePtr CheckerTexture(ColorF color1, ColorF color2, double scale) {
    auto t = std::make_shared<Texture>();
    t->kind = Texture::Checker; t->color1 = color1; t->color2 = color2; t->scale = scale;
    return t;
}
TexturePtr UVTexture() {
    auto t = std::make_shared<Texture>();
    t->kind = Texture::UV;
    return t;
}
TexturePtr ImageTexture(std::shared_ptr<const Surface> image) {
    auto t = std::make_shared<Texture>();
    t->kind = Texture::Image; t->image = std::move(image);
    return t;
}
std::shared_ptr<CubeMap> CubeMap::load(const std::string &x, const std::string &x_neg, const std::string &y,
                                       const std::string &y_neg, const std::string &z, const std::string &z_neg) {
    const std::string *names[6] = {&x, &x_neg, &y, &y_neg, &z, &z_neg};
    auto cm = std::make_shared<CubeMap>();
    for (int i = 0; i < 6; ++i) {
        auto s = std::make_shared<Surface>();
        std::string err;
        if (!util::from_image(*names[i], s.get(), &err)) return nullptr;   // ImageTexture::load unwraps: the reference panics
        cm->faces[i] = s;
    }
    return cm;
}
Material CookTorranceMaterial() { return Material(); }
Material PhongMaterial() {
    Material m;
    m.kind = Material::Phong;
    m.ior = 1.0;
    return m;
}
Material FlatMaterial(Vec3 color) {
    Material m;
    m.kind = Material::Flat;
    m.color = color;
    return m;
}

Prim Sphere(Vec3 center, double radius, MaterialPtr material) {
    Prim p;
    p.kind = Prim::Sphere; p.center = center; p.radius = radius; p.material = std::move(material);
    return p;
}
Prim Plane(double a, double b, double c, double d, MaterialPtr material) {
    Prim p;
    p.kind = Prim::Plane; p.a = a; p.b = b; p.c = c; p.d = d; p.material = std::move(material);
    return p;
}
Prim TriangleOptions::build() const {                                                                                   // triangle.rs:75-87
    Prim p;
    p.kind = Prim::Triangle;
    for (int k = 0; k < 3; ++k) p.vertices[k] = v[k];
    if (has_normals) {
        for (int k = 0; k < 3; ++k) p.normals[k] = n[k];
    } else {                                                                                                            // get_auto_normals :39-42
        Vec3 fn = (v[1] - v[0]).cross(v[2] - v[0]);
        p.normals[0] = p.normals[1] = p.normals[2] = fn;
    }
    if (has_texinfo)
        for (int k = 0; k < 3; ++k) { p.tu[k] = tu[k]; p.tv[k] = tv[k]; }
    p.material = mat ? mat : boxed(FlatMaterial(Vec3::one()));                                                          // :78
    return p;
}
void Prim::mut_transform(const Transform &t) {
    if (kind == Triangle) {                                                                                             // triangle.rs:158-174
        Vec3 v0 = Mat4::mult_p(t.m, vertices[0]), v1 = Mat4::mult_p(t.m, vertices[1]), v2 = Mat4::mult_p(t.m, vertices[2]);
        Vec3 n0 = Mat4::transform_normal(normals[0], t.m), n1 = Mat4::transform_normal(normals[1], t.m),
             n2 = Mat4::transform_normal(normals[2], t.m);
        vertices[0] = v0; vertices[1] = v1; vertices[2] = v2;
        normals[0] = n0; normals[1] = n1; normals[2] = n2;
    } else if (kind == Sphere) {                                                                                        // sphere.rs:75-86
        Vec3 nc = Mat4::mult_p(t.m, center);
        double nr = t.m.has_scale() ? radius * t.m.scale() : radius;
        center = nc; radius = nr;
    } else {                                                                                                            // plane.rs:53-72 ("entirely ad-hoc")
        Vec3 nv = Mat4::mult_v(t.m, Vec3{a, b, c});
        a = nv.x; b = nv.y; c = nv.z;
        Vec3 trans{t.m.m[0][3], t.m.m[1][3], t.m.m[2][3]};
        double t_x = std::pow(t.m.m[0][3], 2.0) * (trans.x < 0.0 ? -1.0 : 1.0);
        double t_y = std::pow(t.m.m[1][3], 2.0) * (trans.y < 0.0 ? -1.0 : 1.0);
        double t_z = std::pow(t.m.m[2][3], 2.0) * (trans.z < 0.0 ? -1.0 : 1.0);
        double add_sub = (t_x + t_y + t_z < 0.0) ? -1.0 : 1.0;
        d += trans.len() * add_sub;
    }
}

// ====================================================================== camera / animation
double Easing::t(double t) const {                                                                                     // easing.rs:21-41
    auto lin = [](double a, double b, double t) { return (1.0 - t) * a + t * b; };
    auto quad = [&](double a, double b, double c, double t) { return lin(lin(a, b, t), lin(b, c, t), t); };
    return lin(quad(a, b, c, t), quad(b, c, d, t), t);
}
Camera Camera::create(Vec3 position, Vec3 look_at, Vec3 up, double fov_deg, uint32_t w, uint32_t h) {                   // camera.rs:26-50
    Camera c;
    c.position = position; c.look_at = look_at; c.up = up; c.fov_deg = fov_deg; c.image_width = w; c.image_height = h;
    c.eye = (look_at - position).unit();                                                                                // :88-91
    c.right = c.eye.cross(up);
    double fov_rad = fov_deg * (PI / 180.0);                                                                            // :93-105
    double ratio = (double)h / (double)w;
    c.half_width = std::tan(fov_rad);
    c.half_height = c.half_width * ratio;
    c.pixel_width = c.half_width * 2.0 / (double)(w - 1);
    c.pixel_height = c.half_height * 2.0 / (double)(h - 1);
    return c;
}
Camera Camera::new_with_keyframes(Vec3 position, Vec3 look_at, Vec3 up, double fov_deg, uint32_t w, uint32_t h,
                                  std::vector<CameraKeyframe> keyframes) {
    Camera c = Camera::create(position, look_at, up, fov_deg, w, h);
    c.insert_keyframes(std::move(keyframes));
    return c;
}
void Camera::insert_keyframes(std::vector<CameraKeyframe> additional) {                                                 // camera.rs:73-86
    std::vector<CameraKeyframe> k;
    k.push_back(CameraKeyframe{0.0, position, look_at, up, Easing::linear()});
    for (auto &kf : additional) k.push_back(kf);
    keyframes = std::move(k);
    has_keyframes = true;
}
rtx_camera Camera::abi() const {
    rtx_camera c{};
    c.position[0] = position.x; c.position[1] = position.y; c.position[2] = position.z;
    c.look_at[0] = look_at.x; c.look_at[1] = look_at.y; c.look_at[2] = look_at.z;
    c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
    c.fov_deg = fov_deg; c.image_width = image_width; c.image_height = image_height;
    return c;
}
void Animator::get_neighbour_keyframes(const std::vector<CameraKeyframe> &keyframes, double time, CameraKeyframe *first_out,
                                       CameraKeyframe *second_out, double *alpha) {                                    // animator.rs:58-89
    if (keyframes.size() <= 1) throw std::runtime_error("Not enough keyframes to interpolate: got: " +
                                                         std::to_string(keyframes.size()) + " expected: >= 2");
    const CameraKeyframe *first = &keyframes[0], *second = &keyframes[1];
    for (const CameraKeyframe &keyframe : keyframes) {
        if (keyframe.time <= time && time - keyframe.time >= first->time - time) first = &keyframe;
        if (keyframe.time > time && (keyframe.time - time < second->time - time || second->time < time)) second = &keyframe;
    }
    double keyframe_length = second->time - first->time;
    *alpha = keyframe_length == 0.0 ? 0.0 : second->easing.t((time - first->time) / keyframe_length);
    *first_out = *first;
    *second_out = *second;
}
Camera Animator::lerp_camera(const Camera &camera, double time) {                                                       // animator.rs:91-114
    if (!camera.has_keyframes) throw std::runtime_error("Cannot lerp a camera with no keyframes!");
    CameraKeyframe first, second;
    double alpha;
    get_neighbour_keyframes(camera.keyframes, time, &first, &second, &alpha);
    Camera c = Camera::create(Vec3::lerp(first.position, second.position, alpha), Vec3::lerp(first.look_at, second.look_at, alpha),
                              Vec3::lerp(first.up, second.up, alpha), camera.fov_deg, camera.image_width, camera.image_height);
    c.keyframes = camera.keyframes;
    c.has_keyframes = true;
    return c;
}

// ====================================================================== flatten: Scene -> rtx_scene_desc
static bool same_material(const Material &a, const Material &b) {
    return a.kind == b.kind && a.k_a == b.k_a && a.k_d == b.k_d && a.k_s == b.k_s && a.k_sg == b.k_sg && a.k_tg == b.k_tg &&
           a.ambient == b.ambient && a.diffuse == b.diffuse && a.transmission == b.transmission && a.specular == b.specular &&
           a.roughness == b.roughness && a.shininess == b.shininess && a.glossiness == b.glossiness &&
           a.gauss_constant == b.gauss_constant && a.ior == b.ior && a.diffuse_texture == b.diffuse_texture && a.color == b.color;
}
std::unique_ptr<FlatScene> flatten(const Scene &scene) {
    std::unique_ptr<FlatScene> f(new FlatScene());
    std::map<const Material *, int> mat_by_ptr;
    std::vector<const Material *> mats;
    std::map<const Texture *, int> tex_index;
    std::map<const Surface *, int> img_index;
    auto image_id = [&](const std::shared_ptr<const Surface> &s) {
        auto it = img_index.find(s.get());
        if (it != img_index.end()) return it->second;
        rtx_image im{};
        im.width = (int32_t)s->width; im.height = (int32_t)s->height;
        im.rgba = reinterpret_cast<const uint8_t *>(s->buffer.data());
        f->images.push_back(im);
        f->image_refs.push_back(s);
        return img_index[s.get()] = (int)f->images.size() - 1;
    };
    auto texture_id = [&](const TexturePtr &t) {
        if (!t) return -1;
        auto it = tex_index.find(t.get());
        if (it != tex_index.end()) return it->second;
        rtx_texture x{};
        x.kind = t->kind == Texture::Checker ? RTX_TEX_CHECKER : t->kind == Texture::UV ? RTX_TEX_UV : RTX_TEX_IMAGE;
        x.image = t->kind == Texture::Image ? image_id(t->image) : -1;
        x.color1[0] = t->color1.r; x.color1[1] = t->color1.g; x.color1[2] = t->color1.b;
        x.color2[0] = t->color2.r; x.color2[1] = t->color2.g; x.color2[2] = t->color2.b;
        x.scale = t->scale;
        f->textures.push_back(x);
        return tex_index[t.get()] = (int)f->textures.size() - 1;
    };
    // the reference clones one material per triangle (util/import.rs:104): dedupe by pointer, then by value
    auto material_id = [&](const MaterialPtr &mp) {
        auto it = mat_by_ptr.find(mp.get());
        if (it != mat_by_ptr.end()) return it->second;
        for (size_t i = 0; i < mats.size(); ++i)
            if (same_material(*mats[i], *mp)) return mat_by_ptr[mp.get()] = (int)i;
        const Material &m = *mp;
        rtx_material r{};
        r.kind = (int32_t)m.kind;
        r.texture = texture_id(m.diffuse_texture);
        r.k_a = m.k_a; r.k_d = m.k_d; r.k_s = m.k_s; r.k_sg = m.k_sg; r.k_tg = m.k_tg;
        const Vec3 &dif = m.kind == Material::Flat ? m.color : m.diffuse;
        r.ambient[0] = m.ambient.x; r.ambient[1] = m.ambient.y; r.ambient[2] = m.ambient.z;
        r.diffuse[0] = dif.x; r.diffuse[1] = dif.y; r.diffuse[2] = dif.z;
        r.specular[0] = m.specular.x; r.specular[1] = m.specular.y; r.specular[2] = m.specular.z;
        r.transmission[0] = m.transmission.x; r.transmission[1] = m.transmission.y; r.transmission[2] = m.transmission.z;
        r.shine = m.kind == Material::Phong ? m.shininess : m.roughness;
        r.gauss_constant = m.gauss_constant;
        r.glossiness = m.glossiness;
        r.ior = m.kind == Material::Flat ? 1.0 : m.ior;
        f->materials.push_back(r);
        mats.push_back(mp.get());
        return mat_by_ptr[mp.get()] = (int)mats.size() - 1;
    };
    for (const Prim &p : scene.prims) {
        int mid = material_id(p.material);
        f->order_kind.push_back((int32_t)p.kind);
        if (p.kind == Prim::Sphere) {
            f->order_index.push_back((int32_t)f->sphere_radius.size());
            f->sphere_center.insert(f->sphere_center.end(), {p.center.x, p.center.y, p.center.z});
            f->sphere_radius.push_back(p.radius);
            f->sphere_material.push_back(mid);
        } else if (p.kind == Prim::Plane) {
            f->order_index.push_back((int32_t)f->plane_material.size());
            f->plane_abcd.insert(f->plane_abcd.end(), {p.a, p.b, p.c, p.d});
            f->plane_material.push_back(mid);
        } else {
            f->order_index.push_back((int32_t)f->tri_material.size());
            for (int k = 0; k < 3; ++k) f->tri_vertices.insert(f->tri_vertices.end(), {p.vertices[k].x, p.vertices[k].y, p.vertices[k].z});
            for (int k = 0; k < 3; ++k) f->tri_normals.insert(f->tri_normals.end(), {p.normals[k].x, p.normals[k].y, p.normals[k].z});
            for (int k = 0; k < 3; ++k) f->tri_uv.insert(f->tri_uv.end(), {p.tu[k], p.tv[k]});
            f->tri_material.push_back(mid);
        }
    }
    for (const Light &l : scene.lights) {
        rtx_light r{};
        r.kind = (int32_t)l.kind;
        r.position[0] = l.position.x; r.position[1] = l.position.y; r.position[2] = l.position.z;
        r.color[0] = l.color.x; r.color[1] = l.color.y; r.color[2] = l.color.z;
        r.radius = l.radius;
        f->lights.push_back(r);
    }
    rtx_scene_desc &d = f->desc;
    for (int i = 0; i < 6; ++i) d.skybox_image[i] = -1;
    if (scene.skybox)
        for (int i = 0; i < 6; ++i) d.skybox_image[i] = image_id(scene.skybox->faces[i]);
    d.n_spheres = (int32_t)f->sphere_radius.size();
    d.sphere_center = f->sphere_center.data(); d.sphere_radius = f->sphere_radius.data(); d.sphere_material = f->sphere_material.data();
    d.n_planes = (int32_t)f->plane_material.size();
    d.plane_abcd = f->plane_abcd.data(); d.plane_material = f->plane_material.data();
    d.n_triangles = (int32_t)f->tri_material.size();
    d.tri_vertices = f->tri_vertices.data(); d.tri_normals = f->tri_normals.data(); d.tri_uv = f->tri_uv.data();
    d.tri_material = f->tri_material.data();
    d.n_prims = (int32_t)f->order_kind.size();
    d.order_kind = f->order_kind.data(); d.order_index = f->order_index.data();
    d.n_materials = (int32_t)f->materials.size(); d.materials = f->materials.data();
    d.n_textures = (int32_t)f->textures.size(); d.textures = f->textures.data();
    d.n_images = (int32_t)f->images.size(); d.images = f->images.data();
    d.n_lights = (int32_t)f->lights.size(); d.lights = f->lights.data();
    d.background[0] = scene.background.x; d.background[1] = scene.background.y; d.background[2] = scene.background.z;
    return f;
}

// ====================================================================== Renderer / Animator
struct DeviceScene {
    rtx_scene *handle = nullptr;
    int device = 0;
    ~DeviceScene() { rtx_scene_destroy(handle); }
};
static void rtx_throw(rtx_status st, const char *where) {
    if (st != RTX_OK) throw std::runtime_error(std::string(where) + ": " + rtx_strerror(st) + ": " + rtx_last_error());
}
Surface Renderer::render(const Camera &camera, const std::shared_ptr<const Scene> &scene, uint32_t frame) const {
    if (!scene->device || scene->device->device != device) {
        auto flat = flatten(*scene);
        auto dev = std::make_shared<DeviceScene>();
        dev->device = device;
        rtx_throw(rtx_scene_create(&flat->desc, device, &dev->handle), "rtx_scene_create");
        scene->device = dev;
    }
    rtx_camera cam = camera.abi();
    rtx_options opt{};
    opt.reflect_depth = options.reflect_depth; opt.refract_depth = options.refract_depth;
    opt.shadow_samples = options.shadow_samples; opt.gloss_samples = options.gloss_samples;
    opt.pixel_samples = options.pixel_samples;
    opt.rng_mode = rng_mode; opt.seed = seed; opt.frame = frame;
    opt.shard_index = shard_index; opt.shard_count = shard_count;
    Surface surface(camera.image_width, camera.image_height, ColorRGBA8{0, 0, 0, 255});                                 // renderer.rs:32-34
    static_assert(sizeof(ColorRGBA8) == 4, "RGBA8");
    rtx_stats st{};
    rtx_throw(rtx_render(scene->device->handle, &cam, &opt, reinterpret_cast<uint8_t *>(surface.buffer.data()), nullptr, &st),
              "rtx_render");
    stats.last = st;
    stats.rays_total += st.rays_primary + st.rays_reflect + st.rays_gloss + st.rays_refract + st.rays_shadow;
    stats.ms_device_total += st.ms_device;
    return surface;
}
void Animator::animate(const Camera &camera, const std::shared_ptr<const Scene> &scene, const std::string &filename) const {
    uint32_t total = total_frames();                                                                                    // animator.rs:23-24
    std::future<void> pending;   // one frame is written while the next renders (animator.rs:26-42)
    for (uint32_t frame_number = 0; frame_number < total; ++frame_number) {
        if (frame_number % frame_shard_count != frame_shard_index) continue;
        double time = frame_time(frame_number);                                                                         // :45
        Camera lerped = lerp_camera(camera, time);
        Surface frame = renderer.render(lerped, scene, frame_number);
        if (pending.valid()) pending.get();
        char num[32];
        std::snprintf(num, sizeof(num), "%06u", starting_frame_number + frame_number);                                  // :35-37
        std::string name = filename + num + ".ppm";
        pending = std::async(std::launch::async, [frame = std::move(frame), name]() {
            std::string err;
            if (!util::to_ppm(frame, name, &err)) throw std::runtime_error("ppm write failure: " + err);
        });
    }
    if (pending.valid()) pending.get();
}

// ====================================================================== util: OBJ, PNG, PPM
namespace util {

static std::vector<std::string> split_ws(const std::string &line) {
    std::vector<std::string> out;
    std::istringstream is(line);
    std::string tok;
    while (is >> tok) out.push_back(tok);
    return out;
}
static bool parse_f64(const std::string &s, double *out) {
    char *end = nullptr;
    *out = std::strtod(s.c_str(), &end);
    return end && *end == '\0' && end != s.c_str();
}
bool from_obj(const Material &material, bool flip_normals, const std::string &filename, Mesh *out, std::string *err) {
    std::ifstream in(filename);
    if (!in) { if (err) *err = "No such file or directory: " + filename; return false; }
    std::vector<Vec3> vertices, normals;
    std::vector<std::pair<double, double>> tex_coords;
    double normal_scale = flip_normals ? -1.0 : 1.0;
    MaterialPtr shared = boxed(material);   // one shared row instead of one clone per triangle (import.rs:104)
    std::string line;
    const size_t NONE = (size_t)-1;
    while (std::getline(in, line)) {
        auto tokens = split_ws(line);
        if (tokens.empty()) continue;
        auto need = [&](size_t n) { return tokens.size() > n; };
        if (tokens[0] == "v" || tokens[0] == "vn") {
            double x, y, z;
            if (!need(3) || !parse_f64(tokens[1], &x) || !parse_f64(tokens[2], &y) || !parse_f64(tokens[3], &z)) {
                if (err) *err = "malformed line: " + line;
                return false;
            }
            if (tokens[0] == "v") vertices.push_back({x, y, z});
            else normals.push_back({x * normal_scale, y * normal_scale, z * normal_scale});
        } else if (tokens[0] == "vt") {
            double u, v;
            if (!need(2) || !parse_f64(tokens[1], &u) || !parse_f64(tokens[2], &v)) { if (err) *err = "malformed line: " + line; return false; }
            tex_coords.push_back({u, v});
        } else if (tokens[0] == "f") {
            // ["f", "1/2/3", "2/2/2", "12//4"] => [[0,1,2],[1,1,1],[11,!0,3]]; only the first three vertices are used
            if (!need(3)) { if (err) *err = "Face syntax of OBJ not supported or malformed"; return false; }
            size_t pairs[3][3];
            for (int k = 0; k < 3; ++k) {
                std::vector<std::string> parts;
                std::string cur;
                for (char ch : tokens[(size_t)k + 1]) {
                    if (ch == '/') { parts.push_back(cur); cur.clear(); } else cur.push_back(ch);
                }
                parts.push_back(cur);
                if (parts.size() < 3) { if (err) *err = "Face syntax of OBJ not supported or malformed"; return false; }   // the reference indexes [1] and [2]
                for (int j = 0; j < 3; ++j) {
                    char *end = nullptr;
                    unsigned long long v = std::strtoull(parts[(size_t)j].c_str(), &end, 10);
                    bool ok = !parts[(size_t)j].empty() && end && *end == '\0' && parts[(size_t)j][0] != '-';
                    pairs[k][j] = ok ? (size_t)v - 1 : NONE;
                }
            }
            double u[3] = {0, 0, 0}, v[3] = {0, 0, 0};
            if (pairs[0][1] != NONE) {
                for (int k = 0; k < 3; ++k) {
                    if (pairs[k][1] >= tex_coords.size()) { if (err) *err = "texture coordinate index out of range"; return false; }
                    u[k] = tex_coords[pairs[k][1]].first; v[k] = tex_coords[pairs[k][1]].second;
                }
            }
            for (int k = 0; k < 3; ++k)
                if (pairs[k][0] >= vertices.size() || pairs[k][2] >= normals.size()) { if (err) *err = "vertex/normal index out of range"; return false; }
            TriangleOptions t(vertices[pairs[0][0]], vertices[pairs[1][0]], vertices[pairs[2][0]]);
            t.material(shared);
            t.normals(normals[pairs[0][2]], normals[pairs[1][2]], normals[pairs[2][2]]);
            t.texinfo(u[0], v[0], u[1], v[1], u[2], v[2]);
            out->triangles.push_back(t.build());
        }
    }
    return true;
}

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
bool from_image(const std::string &path, Surface *out, std::string *err) {
    auto bad = [&](const std::string &m) { if (err) *err = m; return false; };
    std::ifstream in(path, std::ios::binary);
    if (!in) return bad("No such file or directory: " + path);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) return bad("not a PNG file (only PNG is supported here)");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    while (pos + 12 <= data.size()) {
        uint32_t len = be32(&data[pos]);
        std::string type((const char *)&data[pos + 4], 4);
        if (pos + 12 + len > data.size()) return bad("truncated PNG chunk");
        const uint8_t *body = &data[pos + 8];
        if (type == "IHDR") {
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "tRNS") trns.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (w == 0 || h == 0) return bad("PNG without IHDR");
    if (interlace != 0) return bad("interlaced PNG not supported");
    if (depth != 8 && depth != 16) return bad("only 8/16-bit PNG channels are supported");
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (ctype == 3 && depth != 8)) return bad("unsupported PNG colour type");
    size_t bpp = (size_t)channels * (size_t)(depth / 8), stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return bad("PNG inflate failed");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {                                   // PNG filter reconstruction
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t ft = src[0];
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int x = src[i + 1], r;
            switch (ft) {
            case 0: r = x; break;
            case 1: r = x + a; break;
            case 2: r = x + b; break;
            case 3: r = x + ((a + b) >> 1); break;
            case 4: {
                int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                r = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
                break;
            }
            default: return bad("bad PNG filter");
            }
            dst[i] = (uint8_t)r;
        }
    }
    *out = Surface(w, h, ColorRGBA8{0, 0, 0, 0});                       // ColorRGBA::transparent(), import.rs:136-138
    size_t step = (size_t)(depth / 8);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *p = &img[stride * y + (size_t)x * bpp];
            ColorRGBA8 c;
            switch (ctype) {                                             // DynamicImage::to_rgba
            case 0: c = {p[0], p[0], p[0], 255}; break;
            case 2: c = {p[0], p[step], p[2 * step], 255}; break;
            case 3: {
                size_t i = p[0];
                if (i * 3 + 2 >= plte.size()) return bad("palette index out of range");
                c = {plte[i * 3], plte[i * 3 + 1], plte[i * 3 + 2], i < trns.size() ? trns[i] : (uint8_t)255};
                break;
            }
            case 4: c = {p[0], p[0], p[0], p[step]}; break;
            default: c = {p[0], p[step], p[2 * step], p[3 * step]}; break;
            }
            out->at(x, y) = c;
        }
    return true;
}
std::string ppm_bytes(const Surface &surface) {                                                                         // export.rs:5-18
    std::string s = "P3 " + std::to_string(surface.width) + " " + std::to_string(surface.height) + " 255\n";
    s.reserve(s.size() + surface.buffer.size() * 12);
    char buf[16];
    for (const ColorRGBA8 &p : surface.buffer) {
        int n = std::snprintf(buf, sizeof(buf), "%u %u %u ", p.r, p.g, p.b);
        s.append(buf, (size_t)n);
    }
    return s;
}
bool to_ppm(const Surface &surface, const std::string &filename, std::string *err) {
    std::ofstream f(filename, std::ios::binary);
    if (!f) { if (err) *err = "cannot create " + filename; return false; }
    std::string s = ppm_bytes(surface);
    f.write(s.data(), (std::streamsize)s.size());
    return (bool)f;
}
} // namespace util

// ====================================================================== main.rs: JSON job config
namespace {
struct JVal {
    enum { Num, Str, Bool, Arr } kind = Num;
    double num = 0;
    std::string str;
    bool b = false;
    std::vector<double> arr;
};
struct JParser {
    const std::string &s;
    size_t i = 0;
    std::string err;
    explicit JParser(const std::string &t) : s(t) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    bool lit(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    bool str(std::string *out) {
        ws();
        if (i >= s.size() || s[i] != '"') return false;
        ++i;
        out->clear();
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) { ++i; char c = s[i]; out->push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c); }
            else out->push_back(s[i]);
            ++i;
        }
        if (i >= s.size()) return false;
        ++i;
        return true;
    }
    bool num(double *out) {
        ws();
        const char *b = s.c_str() + i;
        char *e = nullptr;
        *out = std::strtod(b, &e);
        if (e == b) return false;
        i += (size_t)(e - b);
        return true;
    }
    bool value(JVal *v) {
        ws();
        if (i >= s.size()) return false;
        if (s[i] == '"') { v->kind = JVal::Str; return str(&v->str); }
        if (s[i] == '[') {
            ++i; v->kind = JVal::Arr;
            if (lit(']')) return true;
            do { double d; if (!num(&d)) return false; v->arr.push_back(d); } while (lit(','));
            return lit(']');
        }
        if (s.compare(i, 4, "true") == 0) { i += 4; v->kind = JVal::Bool; v->b = true; return true; }
        if (s.compare(i, 5, "false") == 0) { i += 5; v->kind = JVal::Bool; v->b = false; return true; }
        v->kind = JVal::Num;
        return num(&v->num);
    }
    bool object(std::map<std::string, JVal> *out) {
        if (!lit('{')) return false;
        if (lit('}')) return true;
        do {
            std::string k;
            JVal v;
            if (!str(&k) || !lit(':') || !value(&v)) return false;
            (*out)[k] = v;
        } while (lit(','));
        return lit('}');
    }
};
} // namespace
std::string parse_job_config(const std::string &json_text, JobConfig *out) {
    JParser p(json_text);
    std::map<std::string, JVal> o;
    if (!p.object(&o)) return "parse failure: SyntaxError at byte " + std::to_string(p.i);
    auto missing = [](const char *f) { return std::string("parse failure, missing field ``") + f + "''\n"; };
    auto expected = [](const char *f, const char *what) { return std::string("parse failure: ExpectedError(\"") + what + "\", \"" + f + "\")"; };
    // field order as rustc_serialize decodes the struct (main.rs:35-48)
#define GET(name, KIND, what)                                         \
    auto it_##name = o.find(#name);                                   \
    if (it_##name == o.end()) return missing(#name);                  \
    if (it_##name->second.kind != JVal::KIND) return expected(#name, what);
    GET(name, Str, "String") out->name = it_name->second.str;
    GET(size, Arr, "Array") if (it_size->second.arr.size() != 2) return expected("size", "Array of 2");
    out->size[0] = (uint32_t)it_size->second.arr[0]; out->size[1] = (uint32_t)it_size->second.arr[1];
    GET(fov, Num, "Number") out->fov = it_fov->second.num;
    GET(reflect_depth, Num, "Number") out->reflect_depth = (uint32_t)it_reflect_depth->second.num;
    GET(refract_depth, Num, "Number") out->refract_depth = (uint32_t)it_refract_depth->second.num;
    GET(shadow_samples, Num, "Number") out->shadow_samples = (uint32_t)it_shadow_samples->second.num;
    GET(gloss_samples, Num, "Number") out->gloss_samples = (uint32_t)it_gloss_samples->second.num;
    GET(pixel_samples, Num, "Number") out->pixel_samples = (uint32_t)it_pixel_samples->second.num;
    GET(output_file, Str, "String") out->output_file = it_output_file->second.str;
    GET(animating, Bool, "Boolean") out->animating = it_animating->second.b;
    GET(fps, Num, "Number") out->fps = it_fps->second.num;
    GET(time_slice, Arr, "Array") if (it_time_slice->second.arr.size() != 2) return expected("time_slice", "Array of 2");
    out->time_slice[0] = it_time_slice->second.arr[0]; out->time_slice[1] = it_time_slice->second.arr[1];
    GET(starting_frame_number, Num, "Number") out->starting_frame_number = (uint32_t)it_starting_frame_number->second.num;
#undef GET
    auto opt = o.find("rng_mode");
    if (opt != o.end() && opt->second.kind == JVal::Str) out->rng_mode = opt->second.str;
    opt = o.find("seed");
    if (opt != o.end() && opt->second.kind == JVal::Num) out->seed = (uint64_t)opt->second.num;
    opt = o.find("device");
    if (opt != o.end() && opt->second.kind == JVal::Num) out->device = (int)opt->second.num;
    opt = o.find("asset_root");
    if (opt != o.end() && opt->second.kind == JVal::Str) out->asset_root = opt->second.str;
    return "";
}

} // namespace rt
