// my_scene.cpp — the scene registry (src/my_scene/mod.rs:27-45): all 14 names of the reference, constants
// transcribed as data.
//
// The reference's model and skybox assets (docs/assets git submodule) are NOT in this
// pipeline.  Each scene first tries the reference's own file paths under `asset_root`
// (so the real raytracer-assets run unchanged when present) and otherwise generates a seeded,
// deterministic procedural stand-in with BASELINE.json's triangle count.
#include <cstdio>
#include <functional>

#include "rt.h"

namespace rt {
namespace {

// ---------------------------------------------------------------------- procedural geometry
struct SurfPoint {
    Vec3 p, n;
};
using SurfFn = std::function<SurfPoint(double u, double v)>;   // u, v in [0, 1]

// A parametric grid: 2*nu*nv triangles with per-vertex (smooth, un-normalised) normals and UVs.
void add_grid(std::vector<Prim> &out, const MaterialPtr &mat, const SurfFn &f, int nu, int nv, bool skip_pole0 = false,
              bool skip_pole1 = false) {
    std::vector<SurfPoint> pts((size_t)(nu + 1) * (size_t)(nv + 1));
    for (int i = 0; i <= nu; ++i)
        for (int j = 0; j <= nv; ++j) pts[(size_t)i * (size_t)(nv + 1) + (size_t)j] = f((double)i / nu, (double)j / nv);
    auto at = [&](int i, int j) -> const SurfPoint & { return pts[(size_t)i * (size_t)(nv + 1) + (size_t)j]; };
    auto emit = [&](int i0, int j0, int i1, int j1, int i2, int j2) {
        const SurfPoint &a = at(i0, j0), &b = at(i1, j1), &c = at(i2, j2);
        TriangleOptions t(a.p, b.p, c.p);
        t.normals(a.n, b.n, c.n);
        t.texinfo((double)i0 / nu, (double)j0 / nv, (double)i1 / nu, (double)j1 / nv, (double)i2 / nu, (double)j2 / nv);
        t.material(mat);
        out.push_back(t.build());
    };
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nv; ++j) {
            if (!(skip_pole0 && j == 0)) emit(i, j, i + 1, j, i + 1, j + 1);
            if (!(skip_pole1 && j == nv - 1)) emit(i, j, i + 1, j + 1, i, j + 1);
        }
}
int scaled(int n, double detail) {
    int v = (int)std::lround(n * detail);
    return v < 2 ? 2 : v;
}
// numeric normal of a position function (central differences): deterministic, smooth
SurfFn with_numeric_normals(std::function<Vec3(double, double)> pos, double flip = 1.0) {
    return [pos, flip](double u, double v) {
        const double h = 1e-4;
        Vec3 du = pos(u + h, v) - pos(u - h, v), dv = pos(u, v + h) - pos(u, v - h);
        return SurfPoint{pos(u, v), du.cross(dv).scale(flip)};
    };
}
SurfFn blob(Vec3 c, double radius, double bump, double ph, Vec3 axes = {1, 1, 1}, bool z_up = false) {
    return with_numeric_normals([=](double u, double v) {
        double th = 2.0 * PI * u, phi = PI * v;
        double r = radius * (1.0 + bump * (std::sin(3.0 * th + ph) * std::sin(4.0 * phi + 2.0 * ph) +
                                          0.5 * std::sin(7.0 * th + 3.0 * ph) * std::sin(5.0 * phi + ph) +
                                          0.25 * std::sin(13.0 * th) * std::sin(11.0 * phi + ph)));
        Vec3 d{std::sin(phi) * std::cos(th), std::cos(phi), std::sin(phi) * std::sin(th)};
        if (z_up) d = Vec3{d.x, d.z, d.y};
        return Vec3{c.x + r * axes.x * d.x, c.y + r * axes.y * d.y, c.z + r * axes.z * d.z};
    }, z_up ? -1.0 : 1.0);   // du x dv points outward for the y-up parametrisation
}
SurfFn cylinder(Vec3 base, double radius, double height, double flute) {   // y-up column with fluting
    return with_numeric_normals([=](double u, double v) {
        double th = 2.0 * PI * u;
        double r = radius * (1.0 + flute * std::cos(12.0 * th)) * (1.0 + 0.15 * std::pow(2.0 * v - 1.0, 8.0));
        return Vec3{base.x + r * std::cos(th), base.y + height * v, base.z + r * std::sin(th)};
    }, -1.0);
}
SurfFn arch(Vec3 a, Vec3 b, double rise, double tube) {   // half torus spanning a -> b, bulging up
    Vec3 mid = (a + b) * 0.5, half = (b - a) * 0.5;
    double span = half.len();
    Vec3 ex = half.unit();
    return with_numeric_normals([=](double u, double v) {
        double al = PI * u, be = 2.0 * PI * v;
        Vec3 centre = mid - ex * (span * std::cos(al)) + Vec3{0, rise * std::sin(al), 0};
        Vec3 radial = (Vec3{0, rise * std::sin(al), 0} - ex * (span * std::cos(al))).unit();
        Vec3 side = ex.cross(Vec3{0, 1, 0}).unit();
        return centre + radial * (tube * std::cos(be)) + side * (tube * std::sin(be));
    }, -1.0);
}
SurfFn wall(Vec3 origin, Vec3 eu, Vec3 ev, double relief, double ph) {   // quad origin + u*eu + v*ev with relief
    Vec3 n = eu.cross(ev).unit();
    return with_numeric_normals([=](double u, double v) {
        double d = relief * (std::sin(40.0 * u + ph) * std::sin(24.0 * v) + 0.3 * std::sin(130.0 * u) * std::sin(90.0 * v + ph));
        return origin + eu * u + ev * v + n * d;
    });
}
SurfFn cloth(Vec3 top_left, Vec3 along, double drop, double wave, double ph) {
    Vec3 out = along.unit().cross(Vec3{0, 1, 0});
    return with_numeric_normals([=](double u, double v) {
        double w = wave * (std::sin(9.0 * u + ph) * (0.3 + v) + 0.4 * std::sin(23.0 * u + 5.0 * v + ph));
        return top_left + along * u + Vec3{0, -drop * v, 0} + out * w;
    });
}

// ---------------------------------------------------------------------- stand-in assets
std::shared_ptr<const CubeMap> procedural_skybox(int size, uint32_t seed) {
    auto cm = std::make_shared<CubeMap>();
    for (int f = 0; f < 6; ++f) {
        auto s = std::make_shared<Surface>((size_t)size, (size_t)size);
        for (int y = 0; y < size; ++y)
            for (int x = 0; x < size; ++x) {
                double u = (x + 0.5) / size, v = (y + 0.5) / size;
                // direction-free smooth pattern per face: gradient + low-frequency "cloud" bands
                double horizon = f == 3 ? 0.15 : f == 2 ? 0.95 : v;   // up face bright sky, down face ground
                double cloud = 0.5 + 0.5 * std::sin(6.0 * u + 1.7 * f + seed) * std::sin(5.0 * v + 0.9 * f) +
                               0.25 * std::sin(17.0 * u + 3.0 * v + f);
                double r = 0.35 + 0.30 * (1.0 - horizon) + 0.18 * cloud;
                double g = 0.45 + 0.30 * (1.0 - horizon) + 0.16 * cloud;
                double b = 0.60 + 0.35 * (1.0 - horizon) + 0.10 * cloud;
                if (f == 2) { r *= 0.5; g *= 0.45; b *= 0.35; }
                auto q = [](double c) { c = c < 0 ? 0 : c > 1 ? 1 : c; return (uint8_t)std::lround(c * 255.0); };
                s->at((size_t)x, (size_t)y) = ColorRGBA8{q(r), q(g), q(b), 255};
            }
        cm->faces[f] = s;
    }
    return cm;
}
std::shared_ptr<const CubeMap> skybox_or_standin(const std::string &root, const char *set, double detail) {
    std::string base = root + "/docs/assets/textures/skyboxes/" + set + "/";
    auto cm = CubeMap::load(base + "left.png", base + "right.png", base + "down.png", base + "up.png", base + "front.png",
                            base + "back.png");
    if (cm) return cm;
    int size = scaled(512, detail);
    return procedural_skybox(size < 8 ? 8 : size, 1);
}
bool obj_or_nothing(const Material &m, const std::string &root, const char *file, std::vector<Prim> &prims, bool flip_normals = false,
                    const Transform *xf = nullptr) {
    Mesh mesh;
    std::string err;
    if (!util::from_obj(m, flip_normals, root + "/docs/assets/models/" + file, &mesh, &err)) return false;
    if (xf) mesh.mut_transform(*xf);
    for (auto &t : mesh.triangles) prims.push_back(std::move(t));
    return true;
}
// A stand-in mesh goes through the same Mesh::mut_transform (mesh.rs:10-14) the reference applies to the loaded OBJ.
void add_grid_transformed(std::vector<Prim> &out, const MaterialPtr &mat, const SurfFn &f, int nu, int nv, const Transform &xf,
                          bool skip_pole0 = false, bool skip_pole1 = false) {
    Mesh mesh;
    add_grid(mesh.triangles, mat, f, nu, nv, skip_pole0, skip_pole1);
    mesh.mut_transform(xf);
    for (auto &t : mesh.triangles) out.push_back(std::move(t));
}
Material ct(double k_a, double k_d, double k_s, double k_sg, double k_tg, double gauss, double rough, double gloss, double ior,
            Vec3 ambient, Vec3 diffuse, Vec3 specular, Vec3 transmission, TexturePtr tex = nullptr) {
    Material m = CookTorranceMaterial();
    m.k_a = k_a; m.k_d = k_d; m.k_s = k_s; m.k_sg = k_sg; m.k_tg = k_tg; m.gauss_constant = gauss; m.roughness = rough;
    m.glossiness = gloss; m.ior = ior; m.ambient = ambient; m.diffuse = diffuse; m.specular = specular;
    m.transmission = transmission; m.diffuse_texture = std::move(tex);
    return m;
}
Material phong(double k_a, double k_d, double k_s, double k_sg, double k_tg, double shininess, double gloss, double ior,
               Vec3 ambient, Vec3 diffuse, Vec3 specular, Vec3 transmission) {
    Material m = PhongMaterial();
    m.k_a = k_a; m.k_d = k_d; m.k_s = k_s; m.k_sg = k_sg; m.k_tg = k_tg; m.shininess = shininess; m.glossiness = gloss;
    m.ior = ior; m.ambient = ambient; m.diffuse = diffuse; m.specular = specular; m.transmission = transmission;
    return m;
}

struct Base : SceneConfig {
    std::string root;
    double detail;
    Base(std::string r, double d) : root(std::move(r)), detail(d) {}
};

// ---------------------------------------------------------------------- my_scene/sphere.rs
struct SphereConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // sphere.rs:16-26
        return Camera::create({0, 0, 10}, {0, 0, 0}, {0, 1, 0}, fov, w, h);
    }
    Camera get_animation_camera(uint32_t w, uint32_t h, double fov) const override {      // sphere.rs:38-81
        return Camera::new_with_keyframes({0, 0, 10}, {0, 0, 0}, {0, 1, 0}, fov, w, h,
                                          {{2.5, {10, 0, 0}, {0, 0, 0}, {0, 1, 0}, Easing::linear()},
                                           {5.0, {0, 0, -10}, {0, 0, 0}, {0, 1, 0}, Easing::linear()},
                                           {7.5, {-10, 0, 0}, {0, 0, 0}, {0, 1, 0}, Easing::linear()},
                                           {10.0, {0, 0, 10}, {0, 0, 0}, {0, 1, 0}, Easing::linear()}});
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // sphere.rs:83-109
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({3, 10, 6}, Vec3::one(), 5.0));
        Material shiny = ct(0.0, 0.2, 1.0, 1.0, 0.0, 5.0, 0.01, 0.0, 0.05, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero());
        s->prims.push_back(Sphere(Vec3::zero(), 2.0, boxed(shiny)));
        s->background = {0.3, 0.5, 0.8};
        s->skybox = skybox_or_standin(root, "storm_y_up", detail);
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/bunny.rs
struct BunnyConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // bunny.rs:15-24
        return Camera::create({0, -150, 30}, {0, 60, 50}, {0, 0, 1}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // bunny.rs:26-59
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({200, -200, 100}, Vec3::one(), 40.0));
        s->lights.push_back(SphereLight({-95, 20, 170}, {0.5, 0.5, 0.3}, 15.0));
        Material red = ct(0.1, 0.4, 0.5, 0.5, 0.0, 5.0, 0.05, 0.0, 0.98, Vec3::one(), {1.0, 0.25, 0.1}, Vec3::one(), Vec3::zero());
        Material green = ct(0.0, 0.4, 0.6, 0.7, 0.0, 50.0, 0.3, 0.0, 1.5, Vec3::one(), {0.2, 0.7, 0.2}, Vec3::one(), Vec3::zero());
        Material shiny = ct(0.0, 0.2, 0.7, 1.0, 0.0, 25.0, 0.01, 0.0, 0.2, Vec3::one(), {0.9, 0.9, 0.1}, {0.9, 0.9, 0.1}, Vec3::zero());
        MaterialPtr shiny_p = boxed(shiny);
        s->prims.push_back(Plane(0.0, 0.0, 1.0, -10.0, boxed(green)));
        s->prims.push_back(Sphere({-75, 60, 50}, 40.0, shiny_p));
        s->prims.push_back(Sphere({-75, 60, 140}, 40.0, shiny_p));
        if (!obj_or_nothing(red, root, "bunny.obj", s->prims)) {
            // stand-in: displaced sphere, 2*187*(188-1) = 69 938 triangles, resting on the z = 10 plane
            int nu = scaled(187, detail), nv = scaled(188, detail);
            add_grid(s->prims, boxed(red), blob({10, 60, 52}, 42.0, 0.12, 0.7, {1, 1, 1}, true), nu, nv, true, true);
        }
        s->background = {0.3, 0.5, 0.8};
        s->skybox = skybox_or_standin(root, "storm_y_up", detail);
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/sponza.rs
struct SponzaConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // sponza.rs:16-25
        return Camera::create({800, 30, 90}, {-500, 1000, -100}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // sponza.rs:27-68
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({0, 3000, 1000}, {1.0, 0.8, 0.4}, 50.0));
        s->lights.push_back(SphereLight({300, 300, 60}, {0.38, 0.32, 0.28}, 20.0));
        TexturePtr checker = CheckerTexture(ColorF::white(), ColorF::new_rgb(0.15, 0.11, 0.1), 32.0);
        Material stone = ct(0.1, 0.8, 0.2, 0.2, 0.0, 50.0, 1.0, 0.0, 1.5, {0.88, 0.83, 0.77}, {0.88, 0.83, 0.77}, Vec3::one(), Vec3::zero());
        Material ground = ct(0.03, 0.9, 0.3, 0.5, 0.0, 25.0, 0.1, 0.0, 0.5, Vec3::one(), {0.38, 0.38, 0.5}, Vec3::one(), Vec3::zero(), checker);
        Material cloth_m = ct(0.03, 0.8, 0.1, 0.05, 0.0, 40.0, 0.8, 0.0, 1.3, Vec3::one(), {0.85, 0.05, 0.05}, Vec3::one(), Vec3::zero());
        Material shrubbery = ct(0.03, 0.8, 0.2, 0.05, 0.0, 50.0, 0.2, 0.0, 1.2, Vec3::one(), {0.16, 0.47, 0.11}, Vec3::one(), Vec3::zero());
        s->prims.push_back(Plane(0.0, 1.0, 0.0, 0.0, boxed(ground)));
        bool real = obj_or_nothing(stone, root, "sponza_other.obj", s->prims) &&
                    obj_or_nothing(shrubbery, root, "sponza_column_shrubbery.obj", s->prims) &&
                    obj_or_nothing(cloth_m, root, "sponza_cloth.obj", s->prims);
        if (!real) {
            s->prims.resize(1);
            build_atrium(s->prims, boxed(stone), boxed(shrubbery), boxed(cloth_m));
        }
        s->background = {0.84, 0.34, 0.0};
        s->skybox = skybox_or_standin(root, "storm_y_up", detail);
        return s;
    }
    // An open-roofed two-storey atrium, 262 144 triangles at detail 1, in the reference's three
    // material groups and insertion order (other, column_shrubbery, cloth).
    void build_atrium(std::vector<Prim> &prims, MaterialPtr stone, MaterialPtr shrub, MaterialPtr cloth_m) const {
        auto S = [&](int n) { return scaled(n, detail); };
        const double X = 1500, Z = 600, HY = 1200;
        // -- "other": outer walls (4 x 9216), gallery slabs (4 x 1024), end-wall relief (43 520),
        //             ground-storey columns (32 x 960) and arches (32 x 256)
        add_grid(prims, stone, wall({-X, 0, -Z}, {2 * X, 0, 0}, {0, HY, 0}, 6.0, 0.3), S(96), S(48));
        add_grid(prims, stone, wall({X, 0, Z}, {-2 * X, 0, 0}, {0, HY, 0}, 6.0, 1.1), S(96), S(48));
        add_grid(prims, stone, wall({-X, 0, Z}, {0, 0, -2 * Z}, {0, HY, 0}, 6.0, 2.3), S(96), S(48));
        add_grid(prims, stone, wall({X, 0, -Z}, {0, 0, 2 * Z}, {0, HY, 0}, 6.0, 0.9), S(96), S(48));
        add_grid(prims, stone, wall({-X, 470, -Z}, {2 * X, 0, 0}, {0, 0, 300}, 2.0, 0.2), S(64), S(8));
        add_grid(prims, stone, wall({-X, 470, Z - 300}, {2 * X, 0, 0}, {0, 0, 300}, 2.0, 0.5), S(64), S(8));
        add_grid(prims, stone, wall({-X, 470, -300}, {200, 0, 0}, {0, 0, 600}, 2.0, 0.8), S(64), S(8));
        add_grid(prims, stone, wall({X - 200, 470, -300}, {200, 0, 0}, {0, 0, 600}, 2.0, 1.4), S(64), S(8));
        add_grid(prims, stone, wall({-X + 8, 80, 420}, {0, 0, -840}, {0, 900, 0}, 22.0, 1.9), S(160), S(136));
        const int NC = 16;
        auto col_x = [&](int i) { return -1300.0 + 2600.0 * i / (NC - 1); };
        for (int side = 0; side < 2; ++side) {
            double z = side ? 300.0 : -300.0;
            for (int i = 0; i < NC; ++i) add_grid(prims, stone, cylinder({col_x(i), 0, z}, 34.0, 450.0, 0.05), S(24), S(20));
            for (int i = 0; i < NC; ++i) {
                double x0 = col_x(i), x1 = i + 1 < NC ? col_x(i + 1) : col_x(i) + 120.0;
                add_grid(prims, stone, arch({x0, 380, z}, {x1, 380, z}, 80.0, 14.0), S(16), S(8));
            }
        }
        // -- "column_shrubbery": upper-storey columns (32 x 960) + arches (32 x 256) + shrubs (16 x 1536)
        for (int side = 0; side < 2; ++side) {
            double z = side ? 300.0 : -300.0;
            for (int i = 0; i < NC; ++i) add_grid(prims, shrub, cylinder({col_x(i), 500, z}, 26.0, 420.0, 0.08), S(24), S(20));
            for (int i = 0; i < NC; ++i) {
                double x0 = col_x(i), x1 = i + 1 < NC ? col_x(i + 1) : col_x(i) + 120.0;
                add_grid(prims, shrub, arch({x0, 860, z}, {x1, 860, z}, 70.0, 11.0), S(16), S(8));
            }
        }
        for (int k = 0; k < 16; ++k) {
            double x = -1100.0 + 146.0 * k, z = (k & 1) ? 190.0 : -190.0;
            add_grid(prims, shrub, blob({x, 55.0, z}, 48.0, 0.22, 0.37 * k, {1.0, 1.1, 1.0}), S(32), S(25), true, true);
        }
        // -- "cloth": 12 hanging sheets (12 x 6272)
        for (int k = 0; k < 12; ++k) {
            double x = -1150.0 + 200.0 * k, z = (k % 3 == 0) ? -296.0 : (k % 3 == 1 ? 296.0 : 0.0);
            double top = (k % 3 == 2) ? 900.0 : 470.0;
            add_grid(prims, cloth_m, cloth({x, top, z}, {150, 0, (k % 3 == 2) ? 60.0 : 0.0}, 330.0, 16.0, 0.6 * k), S(56), S(56));
        }
    }
};

// ---------------------------------------------------------------------- my_scene/heptoroid.rs
struct HeptoroidConfig : Base {
    int material;   // 0 shiny, 1 refractive, 2 white
    HeptoroidConfig(std::string r, double d, int m) : Base(std::move(r), d), material(m) {}
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // heptoroid.rs:15-24
        return Camera::create({7, 2, -6}, {0, 0, 0}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // heptoroid.rs:26-63
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({2, 3, -2}, {1, 1, 1}, 1.0));
        Material m = material == 0
                         ? ct(0.0, 0.2, 1.0, 0.55, 0.0, 5.0, 0.01, 0.0, 0.25, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero())
                     : material == 1
                         ? ct(0.0, 0.0, 1.0, 1.0, 1.0, 5.0, 0.01, 0.0, 1.50, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, {0.8, 0.8, 0.8})
                         : ct(0.0, 0.9, 1.0, 0.15, 0.0, 5.0, 0.1, 0.0, 0.5, Vec3::one(), {0.9, 0.85, 0.7}, Vec3::one(), Vec3::zero());
        if (!obj_or_nothing(m, root, "heptoroid.obj", s->prims)) {
            // stand-in: a 7-lobed, twisted torus, 2*256*224 = 114 688 triangles
            SurfFn f = with_numeric_normals([](double u, double v) {
                double a = 2.0 * PI * u, b = 2.0 * PI * v;
                double R = 3.45 + 0.5 * std::cos(7.0 * a);
                double r = 1.1 + 0.27 * std::sin(7.0 * a + 2.0 * b);
                double y = r * std::sin(b) + 0.4 * std::sin(7.0 * a);
                return Vec3{(R + r * std::cos(b)) * std::cos(a), y, (R + r * std::cos(b)) * std::sin(a)};
            }, -1.0);
            add_grid(s->prims, boxed(m), f, scaled(256, detail), scaled(224, detail));
        }
        s->background = {0.84, 0.34, 0.0};
        s->skybox = skybox_or_standin(root, "miramar_y_up", detail);
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/sibenik.rs
struct SibenikConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // sibenik.rs:17-26
        return Camera::create({-16, -14.5, -2}, {8, -3, 2}, {0, 1, 0}, fov, w, h);
    }
    Camera get_animation_camera(uint32_t w, uint32_t h, double fov) const override {      // sibenik.rs:28-46 (7 s)
        return Camera::new_with_keyframes({-16, -14.5, -2}, {8, -3, 2}, {0, 1, 0}, fov, w, h,
                                          {{7.0, {8, -13.5, 0.2}, {8.5, 8.0, 2.0}, {-0.9, 0.0, -0.7}, Easing::linear()}});
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // sibenik.rs:48-76
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({8, 8, 0}, {1.0, 0.8, 0.4}, 0.5));
        s->lights.push_back(SphereLight({8, -5, 0}, {0.5, 0.4, 0.2}, 1.0));
        s->lights.push_back(PointLight({-16, -14.5, -2}, {0.15, 0.07, 0.05}));
        TexturePtr checker = CheckerTexture(ColorF::white(), ColorF::new_rgb(0.15, 0.11, 0.1), 1.0);
        Material stone = ct(0.1, 0.8, 0.2, 0.0, 0.0, 25.0, 1.0, 0.0, 1.5, {0.88, 0.83, 0.77}, {0.88, 0.83, 0.77}, Vec3::one(), Vec3::zero());
        Material ground = ct(0.03, 0.9, 0.3, 0.5, 0.0, 25.0, 0.1, 0.0, 0.5, Vec3::one(), {0.38, 0.38, 0.5}, Vec3::one(), Vec3::zero(), checker);
        s->prims.push_back(Plane(0.0, -1.0, 0.0, -14.9, boxed(ground)));
        if (!obj_or_nothing(stone, root, "sibenik.obj", s->prims)) {
            // stand-in nave, ~75k triangles: walls, barrel vault, two rows of columns with arches, a pulpit
            auto S = [&](int n) { return scaled(n, detail); };
            MaterialPtr st = boxed(stone);
            const double X = 20, Z = 10, Y0 = -15, Y1 = 6;
            add_grid(s->prims, st, wall({-X, Y0, -Z}, {2 * X, 0, 0}, {0, Y1 - Y0, 0}, 0.08, 0.3), S(128), S(48));
            add_grid(s->prims, st, wall({X, Y0, Z}, {-2 * X, 0, 0}, {0, Y1 - Y0, 0}, 0.08, 1.2), S(128), S(48));
            add_grid(s->prims, st, wall({-X, Y0, Z}, {0, 0, -2 * Z}, {0, Y1 - Y0 + 9, 0}, 0.10, 2.1), S(64), S(48));
            add_grid(s->prims, st, wall({X, Y0, -Z}, {0, 0, 2 * Z}, {0, Y1 - Y0 + 9, 0}, 0.10, 0.7), S(64), S(48));
            add_grid(s->prims, st, with_numeric_normals([=](double u, double v) {   // barrel vault
                         double a = PI * v;
                         return Vec3{-X + 2 * X * u, Y1 + 9.0 * std::sin(a) + 0.05 * std::sin(60.0 * u), -Z * std::cos(a)};
                     }), S(128), S(48));
            for (int side = 0; side < 2; ++side) {
                double z = side ? 5.0 : -5.0;
                for (int i = 0; i < 6; ++i) add_grid(s->prims, st, cylinder({-15.0 + 6.0 * i, Y0, z}, 0.7, 13.0, 0.04), S(32), S(24));
                for (int i = 0; i < 5; ++i)
                    add_grid(s->prims, st, arch({-15.0 + 6.0 * i, -2.5, z}, {-9.0 + 6.0 * i, -2.5, z}, 2.5, 0.35), S(24), S(12));
            }
            add_grid(s->prims, st, blob({12, -13.2, 0}, 1.6, 0.15, 0.9, {1.4, 1.0, 1.4}), S(32), S(25), true, true);
        }
        s->background = {0.5, 0.5, 0.5};
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/cornell.rs
struct CornellConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // cornell.rs:16-25
        return Camera::create({50, 25, 150}, {50, 50, 50}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // cornell.rs:27-75
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({50, 80, 50}, Vec3::one(), 10.0));
        TexturePtr checker = CheckerTexture(ColorF::white(), ColorF::new_rgb(0.8, 0.1, 0.1), 16.0);
        Material checker_grey = ct(0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.15, 0.0, 0.7, Vec3::one(), {0.6, 0.6, 0.6}, Vec3::one(), Vec3::zero(), checker);
        Material grey = CookTorranceMaterial();
        grey.diffuse = {0.6, 0.6, 0.6};
        Material blue = ct(0.0, 0.3, 0.7, 0.0, 0.0, 50.0, 0.1, 0.0, 1.3, Vec3::one(), {0.1, 0.1, 1.0}, Vec3::one(), Vec3::zero());
        Material red = phong(0.0, 0.6, 0.4, 0.8, 0.0, 10.0, 0.0, 0.5, Vec3::one(), {1.0, 0.0, 0.0}, Vec3::one(), Vec3::zero());
        Material green = phong(0.0, 0.9, 0.1, 0.5, 0.0, 10.0, 0.0, 0.7, Vec3::one(), {0.0, 1.0, 0.0}, Vec3::one(), Vec3::zero());
        Material shiny = ct(0.0, 0.2, 1.0, 0.8, 0.0, 5.0, 0.01, 0.0, 0.25, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero());
        Material shiny_glossy = ct(0.0, 0.7, 1.0, 0.4, 0.0, 5.0, 0.01, 0.2, 0.25, Vec3::one(), {0.3, 0.3, 1.0}, {0.3, 0.3, 1.0}, Vec3::zero());
        Material refract = ct(0.0, 0.0, 1.0, 1.0, 1.0, 5.0, 0.01, 0.0, 3.0, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, {0.8, 0.8, 0.8});
        MaterialPtr grey_p = boxed(grey), blue_p = boxed(blue);
        s->prims.push_back(Plane(0.0, 0.0, 1.0, 0.0, grey_p));                 // Ahead
        s->prims.push_back(Plane(0.0, 1.0, 0.0, 0.0, boxed(checker_grey)));    // Bottom
        s->prims.push_back(Plane(0.0, -1.0, 0.0, 100.0, grey_p));              // Top
        s->prims.push_back(Plane(1.0, 0.0, 0.0, 0.0, boxed(red)));             // Left
        s->prims.push_back(Plane(-1.0, 0.0, 0.0, 100.0, boxed(green)));        // Right
        s->prims.push_back(Sphere({30, 15, 20}, 15.0, boxed(shiny)));
        s->prims.push_back(Sphere({70, 17, 60}, 17.0, boxed(refract)));
        s->prims.push_back(Sphere({50, 50, 20}, 10.0, boxed(shiny_glossy)));
        s->prims.push_back(Sphere({20, 13, 90}, 13.0, blue_p));
        TriangleOptions t({20, 95, 20}, {15, 50, 40}, {35, 50, 35});
        t.texinfo(0.5, 1.0, 0.0, 0.0, 1.0, 0.0);
        t.material(blue_p);
        s->prims.push_back(t.build());
        s->background = Vec3::one();
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/cow.rs
struct CowConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // cow.rs:15-24
        return Camera::create({-2, 4, 10}, {0, 0, 0}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // cow.rs:26-48
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({3, 10, 6}, Vec3::one(), 5.0));
        Material red = ct(0.0, 0.6, 1.0, 0.2, 0.0, 30.0, 0.1, 0.0, 0.8, Vec3::one(), {1.0, 0.25, 0.1}, Vec3::one(), Vec3::zero());
        Material green = ct(0.0, 0.5, 0.4, 0.1, 0.0, 25.0, 0.4, 0.0, 0.95, Vec3::one(), {0.2, 0.7, 0.2}, Vec3::one(), Vec3::zero());
        s->prims.push_back(Plane(0.0, 1.0, 0.0, 3.6, boxed(green)));
        if (!obj_or_nothing(red, root, "cow.obj", s->prims, true)) {
            // stand-in ("5000 polys", cow.rs:14): an elongated blob standing on the y = -3.6 plane, 2*50*50 triangles
            add_grid(s->prims, boxed(red), blob({0, -0.6, 0}, 3.0, 0.18, 1.3, {1.5, 1.0, 0.8}), scaled(50, detail), scaled(50, detail));
        }
        s->background = {0.3, 0.5, 0.8};
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/easing.rs
struct EasingConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // easing.rs:18-27
        return Camera::create({0, 0, 150}, {0, 0, 0}, {0, 1, 0}, fov, w, h);
    }
    Camera get_animation_camera(uint32_t w, uint32_t h, double fov) const override {      // easing.rs:29-47
        return Camera::new_with_keyframes({0, 0, 150}, {0, 0, 0}, {0, 1, 0}, fov, w, h,
                                          {{10.0, {0, 1000, 150}, {0, 1000, 0}, {0, 1, 0}, Easing{0.0, 0.05, 0.1, 1.0}}});
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // easing.rs:49-105
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({0, 0, 150}, Vec3::one(), 10.0));
        s->lights.push_back(SphereLight({0, 1000, 150}, Vec3::one(), 10.0));
        TexturePtr checker = CheckerTexture(ColorF::white(), ColorF::new_rgb(0.1, 0.1, 0.1), 32.0);
        Material checker_mat = ct(0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.15, 0.0, 0.7, Vec3::one(), {0.6, 0.6, 0.6}, Vec3::one(), Vec3::zero(), checker);
        s->prims.push_back(Plane(0.0, 0.0, 1.0, 0.0, boxed(checker_mat)));
        s->background = {1, 1, 1};
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/fresnel.rs
struct FresnelConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // fresnel.rs:18-29
        return Camera::create({50, 50, 250}, {50, 50, 50}, {0, 1, 0}, fov, w, h);
    }
    Camera get_animation_camera(uint32_t w, uint32_t h, double fov) const override {      // fresnel.rs:31-72
        return Camera::new_with_keyframes({0, 1, 250}, {0, 1, 50}, {0, 1, 0}, fov, w, h,
                                          {{2.5, {50, 100, 250}, {0, 1, 50}, {0, 1, 0}, Easing::linear()},
                                           {5.0, {0, 200, 250}, {0, 1, 50}, {0, 1, 0}, Easing::linear()},
                                           {7.5, {-50, 100, 250}, {0, 1, 50}, {0, 1, 0}, Easing::linear()},
                                           {10.0, {0, 1, 250}, {0, 1, 50}, {0, 1, 0}, Easing::linear()}});
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // fresnel.rs:74-101
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({50, 80, 50}, Vec3::one(), 10.0));
        TexturePtr checker = CheckerTexture(ColorF::white(), ColorF::new_rgb(0.1, 0.1, 0.1), 32.0);
        Material checker_red = ct(0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.15, 0.0, 1.5, Vec3::one(), {0.6, 0.6, 0.6}, Vec3::one(), Vec3::zero(), checker);
        Material shiny = ct(0.0, 0.2, 1.0, 1.0, 0.0, 5.0, 0.01, 0.0, 0.15, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero());
        Material global_specular_only = ct(0.0, 0.0, 0.0, 1.0, 0.0, 5.0, 0.01, 0.0, 1.5, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero());
        Material refract = ct(0.0, 0.0, 1.0, 1.0, 1.0, 5.0, 0.01, 0.0, 3.0, Vec3::one(), {1, 1, 1}, {0.9, 0.9, 0.9}, Vec3::zero());
        s->prims.push_back(Plane(0.0, 0.0, 1.0, 0.0, boxed(checker_red)));            // Ahead
        s->prims.push_back(Plane(0.0, 1.0, 0.0, 0.0, boxed(global_specular_only)));   // Bottom
        s->prims.push_back(Sphere({30, 15, 20}, 15.0, boxed(shiny)));
        s->prims.push_back(Sphere({70, 17, 60}, 17.0, boxed(refract)));
        s->background = {1, 1, 1};
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/lucy.rs
struct LucyConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // lucy.rs:15-24
        return Camera::create({-1500, 300, 600}, {0, 400, -200}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // lucy.rs:26-53
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({-1400, 200, 100}, {1.0, 0.80, 0.40}, 50.0));
        Material grey = ct(0.0, 0.5, 0.8, 0.5, 0.0, 5.0, 0.1, 0.0, 0.4, Vec3::one(), {0.6, 0.6, 0.65}, Vec3::one(), Vec3::zero());
        if (!obj_or_nothing(grey, root, "lucy.obj", s->prims, true)) {
            // stand-in ("50000 polys, model not included!", lucy.rs:14): a tall statue-like blob, 2*158*158 = 49 928 triangles
            add_grid(s->prims, boxed(grey), blob({0, 420, -200}, 330.0, 0.2, 2.1, {0.55, 1.25, 0.55}), scaled(158, detail), scaled(158, detail));
        }
        s->background = {0.84, 0.34, 0.0};
        s->skybox = skybox_or_standin(root, "storm_y_up", detail);
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/tachikoma.rs
struct TachikomaConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // tachikoma.rs:14-23
        return Camera::create({100, 60, -150}, {0, 50, 0}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // tachikoma.rs:25-56
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({0, 100, 0}, {1, 1, 1}, 25.0));
        Material blue = ct(0.0, 0.9, 1.0, 0.4, 0.0, 5.0, 0.01, 0.0, 0.25, Vec3::one(), {0.16, 0.29, 0.44}, Vec3::one(), Vec3::zero());
        Material floor = ct(0.0, 0.9, 1.0, 1.0, 0.0, 5.0, 0.3, 0.0, 1.0, Vec3::one(), {0.58, 0.63, 0.44}, {0.9, 0.9, 0.9}, Vec3::zero());
        s->prims.push_back(Plane(0.0, 1.0, 0.0, 0.0, boxed(floor)));   // Bottom
        if (!obj_or_nothing(blue, root, "tachikoma.obj", s->prims, false)) {
            // stand-in: a pod (body) on four leg blobs
            MaterialPtr b = boxed(blue);
            add_grid(s->prims, b, blob({0, 55, 0}, 32.0, 0.1, 0.4, {1.2, 0.9, 1.4}), scaled(64, detail), scaled(48, detail));
            for (int k = 0; k < 4; ++k) {
                double x = (k & 1) ? 34.0 : -34.0, z = (k & 2) ? 38.0 : -38.0;
                add_grid(s->prims, b, blob({x, 22, z}, 14.0, 0.08, 0.9 * k, {0.8, 1.6, 0.8}), scaled(32, detail), scaled(24, detail));
            }
        }
        s->background = {0.2, 0.2, 0.2};
        s->skybox = skybox_or_standin(root, "city_y_up", detail);
        return s;
    }
};

// ---------------------------------------------------------------------- my_scene/teapot.rs
struct TeapotConfig : Base {
    using Base::Base;
    Camera get_camera(uint32_t w, uint32_t h, double fov) const override {                // teapot.rs:19-28
        return Camera::create({-0.2, 1.0, 2.0}, {0, 0.6, 0}, {0, 1, 0}, fov, w, h);
    }
    std::shared_ptr<Scene> get_scene() const override {                                    // teapot.rs:30-62
        auto s = std::make_shared<Scene>();
        s->lights.push_back(SphereLight({0.6, 2.0, 1.2}, Vec3::one(), 1.0));
        Material porcelain = ct(0.0, 0.9, 1.0, 1.0, 0.0, 5.0, 0.1, 0.0, 1.1, Vec3::one(), {0.9, 0.85, 0.7}, Vec3::one(), Vec3::zero());
        Transform rotate(Mat4::rotate_x_deg_matrix(1.0));                                  // teapot.rs:39-40
        if (!obj_or_nothing(porcelain, root, "teapot.obj", s->prims, false, &rotate)) {
            // stand-in ("2500 polys", teapot.rs:18): a squat pot + a spout blob, through the same mut_transform
            MaterialPtr m = boxed(porcelain);
            add_grid_transformed(s->prims, m, blob({0, 0.55, 0}, 0.55, 0.05, 0.2, {1.25, 0.9, 1.25}), scaled(36, detail), scaled(28, detail), rotate);
            add_grid_transformed(s->prims, m, blob({0.78, 0.7, 0}, 0.16, 0.1, 1.1, {1.9, 1.0, 0.7}), scaled(16, detail), scaled(12, detail), rotate);
        }
        s->background = {0.3, 0.5, 0.8};
        s->skybox = skybox_or_standin(root, "miramar_y_up", detail);
        return s;
    }
};

} // namespace

std::unique_ptr<SceneConfig> scene_by_name(const std::string &name, const std::string &asset_root, double detail) {
    if (name == "bunny") return std::unique_ptr<SceneConfig>(new BunnyConfig(asset_root, detail));
    if (name == "cornell") return std::unique_ptr<SceneConfig>(new CornellConfig(asset_root, detail));
    if (name == "cow") return std::unique_ptr<SceneConfig>(new CowConfig(asset_root, detail));
    if (name == "easing") return std::unique_ptr<SceneConfig>(new EasingConfig(asset_root, detail));
    if (name == "fresnel") return std::unique_ptr<SceneConfig>(new FresnelConfig(asset_root, detail));
    if (name == "heptoroid-shiny") return std::unique_ptr<SceneConfig>(new HeptoroidConfig(asset_root, detail, 0));
    if (name == "heptoroid-refractive") return std::unique_ptr<SceneConfig>(new HeptoroidConfig(asset_root, detail, 1));
    if (name == "heptoroid-white") return std::unique_ptr<SceneConfig>(new HeptoroidConfig(asset_root, detail, 2));
    if (name == "lucy") return std::unique_ptr<SceneConfig>(new LucyConfig(asset_root, detail));
    if (name == "sibenik") return std::unique_ptr<SceneConfig>(new SibenikConfig(asset_root, detail));
    if (name == "sphere") return std::unique_ptr<SceneConfig>(new SphereConfig(asset_root, detail));
    if (name == "sponza") return std::unique_ptr<SceneConfig>(new SponzaConfig(asset_root, detail));
    if (name == "tachikoma") return std::unique_ptr<SceneConfig>(new TachikomaConfig(asset_root, detail));
    if (name == "teapot") return std::unique_ptr<SceneConfig>(new TeapotConfig(asset_root, detail));
    return nullptr;   // my_scene/mod.rs:43: `_ => return None`
}

} // namespace rt
