// capi.cpp — a small C interface over the C++ host mirror, for the Python harness (tests,
// bench.py) to obtain the registry's scenes as rtx_scene_desc data and to reach the
// loaders/writers.  Not part of include/rtx.h: the drop-in boundary is librtx.so.
#include <cstring>
#include <memory>
#include <string>

#include "rt.h"

using namespace rt;

struct rtxh_scene {
    std::shared_ptr<Scene> scene;
    std::unique_ptr<FlatScene> flat;
    std::unique_ptr<SceneConfig> config;
};

static thread_local std::string g_err;

extern "C" {

const char *rtxh_last_error(void) { return g_err.c_str(); }

// scene_by_name(name) + get_scene() + flatten.  NULL for an unknown scene name.
rtxh_scene *rtxh_scene_by_name(const char *name, const char *asset_root, double detail) {
    auto cfg = scene_by_name(name, asset_root ? asset_root : ".", detail);
    if (!cfg) { g_err = std::string("unknown scene ``") + name + "''"; return nullptr; }
    auto *h = new rtxh_scene();
    h->scene = cfg->get_scene();
    h->flat = flatten(*h->scene);
    h->config = std::move(cfg);
    return h;
}
void rtxh_scene_free(rtxh_scene *h) { delete h; }
const rtx_scene_desc *rtxh_scene_desc(const rtxh_scene *h) { return &h->flat->desc; }

// get_camera / get_animation_camera + Animator::lerp_camera at `time` (animated != 0)
int rtxh_scene_camera(const rtxh_scene *h, uint32_t w, uint32_t hgt, double fov, int animated, double time, rtx_camera *out) {
    try {
        Camera c = animated ? Animator::lerp_camera(h->config->get_animation_camera(w, hgt, fov), time)
                            : h->config->get_camera(w, hgt, fov);
        *out = c.abi();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// util::import::from_obj -> flattened triangles (vertices[n][9], normals[n][9], uv[n][6]); returns n or -1
struct rtxh_mesh { Mesh mesh; };
rtxh_mesh *rtxh_from_obj(const char *filename, int flip_normals) {
    auto *m = new rtxh_mesh();
    if (!util::from_obj(CookTorranceMaterial(), flip_normals != 0, filename, &m->mesh, &g_err)) { delete m; return nullptr; }
    return m;
}
int64_t rtxh_mesh_size(const rtxh_mesh *m) { return (int64_t)m->mesh.triangles.size(); }
void rtxh_mesh_copy(const rtxh_mesh *m, double *vertices, double *normals, double *uv) {
    size_t i = 0;
    for (const Prim &t : m->mesh.triangles) {
        for (int k = 0; k < 3; ++k) {
            vertices[i * 9 + k * 3 + 0] = t.vertices[k].x; vertices[i * 9 + k * 3 + 1] = t.vertices[k].y; vertices[i * 9 + k * 3 + 2] = t.vertices[k].z;
            normals[i * 9 + k * 3 + 0] = t.normals[k].x; normals[i * 9 + k * 3 + 1] = t.normals[k].y; normals[i * 9 + k * 3 + 2] = t.normals[k].z;
            uv[i * 6 + k * 2 + 0] = t.tu[k]; uv[i * 6 + k * 2 + 1] = t.tv[k];
        }
        ++i;
    }
}
void rtxh_mesh_transform(rtxh_mesh *m, const double *mat16) {
    Mat4 mm;
    std::memcpy(mm.m, mat16, sizeof(mm.m));
    m->mesh.mut_transform(Transform(mm));
}
void rtxh_mesh_free(rtxh_mesh *m) { delete m; }

// util::import::from_image: returns 0 and fills w/h; call again with rgba != NULL to copy
int rtxh_from_image(const char *path, int32_t *w, int32_t *h, uint8_t *rgba) {
    Surface s;
    if (!util::from_image(path, &s, &g_err)) return 1;
    *w = (int32_t)s.width; *h = (int32_t)s.height;
    if (rgba) std::memcpy(rgba, s.buffer.data(), s.buffer.size() * 4);
    return 0;
}
// util::export::to_ppm
int rtxh_to_ppm(const uint8_t *rgba, int32_t w, int32_t h, const char *filename) {
    Surface s((size_t)w, (size_t)h);
    std::memcpy(s.buffer.data(), rgba, (size_t)w * h * 4);
    return util::to_ppm(s, filename, &g_err) ? 0 : 1;
}
void rtxh_mat4_inverse(const double *in16, double *out16) {
    Mat4 m;
    std::memcpy(m.m, in16, sizeof(m.m));
    Mat4 r = m.inverse();
    std::memcpy(out16, r.m, sizeof(r.m));
}
void rtxh_mat4_mult_p(const double *m16, const double *p3, double *out3) {
    Mat4 m;
    std::memcpy(m.m, m16, sizeof(m.m));
    Vec3 r = Mat4::mult_p(m, Vec3{p3[0], p3[1], p3[2]});
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
double rtxh_easing_t(double a, double b, double c, double d, double t) {
    Easing e;
    e.a = a; e.b = b; e.c = c; e.d = d;
    return e.t(t);
}
// test_lerp_camera_position's camera (animator.rs:121-159) generalised: keyframes[n][10] = time, pos, look_at, up
int rtxh_lerp_camera(const rtx_camera *base, const double *keyframes, int n, double time, rtx_camera *out) {
    try {
        std::vector<CameraKeyframe> k;
        for (int i = 0; i < n; ++i) {
            const double *p = keyframes + i * 10;
            k.push_back(CameraKeyframe{p[0], {p[1], p[2], p[3]}, {p[4], p[5], p[6]}, {p[7], p[8], p[9]}, Easing::linear()});
        }
        Camera c = Camera::new_with_keyframes({base->position[0], base->position[1], base->position[2]},
                                              {base->look_at[0], base->look_at[1], base->look_at[2]},
                                              {base->up[0], base->up[1], base->up[2]}, base->fov_deg, base->image_width,
                                              base->image_height, k);
        *out = Animator::lerp_camera(c, time).abi();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}
// parse_job_config: returns 0 and fills the numeric fields, or 1 with rtxh_last_error() = the reference's message
int rtxh_parse_config(const char *json, char *name_out, int name_cap, uint32_t *u32_out /*[8]*/, double *f64_out /*[4]*/) {
    JobConfig c;
    g_err = parse_job_config(json, &c);
    if (!g_err.empty()) return 1;
    std::snprintf(name_out, (size_t)name_cap, "%s", c.name.c_str());
    u32_out[0] = c.size[0]; u32_out[1] = c.size[1]; u32_out[2] = c.reflect_depth; u32_out[3] = c.refract_depth;
    u32_out[4] = c.shadow_samples; u32_out[5] = c.gloss_samples; u32_out[6] = c.pixel_samples; u32_out[7] = c.starting_frame_number;
    f64_out[0] = c.fov; f64_out[1] = c.fps; f64_out[2] = c.time_slice[0]; f64_out[3] = c.time_slice[1];
    return 0;
}

} // extern "C"
