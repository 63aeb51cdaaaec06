//! flatten.rs — turns the crate's `Scene` (a graph of `Box<dyn Trait>` objects, src/scene/scene.rs:7-12)
//! into the plain arrays of `RtxSceneDesc`.  Drop in as `src/raytracer/gpu/flatten.rs`.
//!
//! Trait objects cannot cross an FFI, so each trait gains ONE method that pushes the concrete type's
//! fields into a `FlatScene` (Rust 2015 syntax, like the rest of the crate):
//!
//!   src/geometry/prim.rs:5-10        trait Prim      { ...; fn flatten(&self, out: &mut FlatScene); }
//!   src/material/material.rs:4-14    trait Material  { ...; fn flatten(&self, out: &mut FlatScene) -> i32; }
//!   src/material/texture.rs:3-6      trait Texture   { ...; fn flatten(&self, out: &mut FlatScene) -> i32; }
//!   src/light/light.rs:3-8           trait Light     { ...; fn flatten(&self, out: &mut FlatScene); }
//!
//! and `Octree<T>` (src/raytracer/octree.rs:8-13) exposes its `prims` and `infinites` vectors in
//! insertion order (`pub fn prims_in_order(&self) -> impl Iterator<Item = &T>`): that order defines
//! the prim ids librtx reports (rtx.h: finite prims first, planes after).  The octree itself is
//! rebuilt inside librtx exactly as `Octree::from_iter` builds it (bit-identical nodes), so nothing
//! of its node structure has to cross.
use std::collections::HashMap;
use super::rtx_ffi::*;
use vec3::Vec3;

fn v3(v: &Vec3) -> [f64; 3] { [v.x, v.y, v.z] }

#[derive(Default)]
pub struct FlatScene {
    pub sphere_center: Vec<f64>, pub sphere_radius: Vec<f64>, pub sphere_material: Vec<i32>,
    pub plane_abcd: Vec<f64>, pub plane_material: Vec<i32>,
    pub tri_vertices: Vec<f64>, pub tri_normals: Vec<f64>, pub tri_uv: Vec<f64>, pub tri_material: Vec<i32>,
    pub order_kind: Vec<i32>, pub order_index: Vec<i32>,
    pub materials: Vec<RtxMaterial>, pub textures: Vec<RtxTexture>,
    pub image_pixels: Vec<Vec<u8>>, pub image_sizes: Vec<(i32, i32)>,
    pub lights: Vec<RtxLight>,
    pub background: [f64; 3], pub skybox_image: [i32; 6],
    // the OBJ importer clones one Box<Material> per triangle (util/import.rs:104): de-duplicate by value
    material_index: HashMap<Vec<u64>, i32>,
}

impl FlatScene {
    pub fn push_sphere(&mut self, center: &Vec3, radius: f64, material: i32) {
        self.order_kind.push(RTX_PRIM_SPHERE);
        self.order_index.push(self.sphere_radius.len() as i32);
        self.sphere_center.extend_from_slice(&v3(center));
        self.sphere_radius.push(radius);
        self.sphere_material.push(material);
    }
    pub fn push_plane(&mut self, a: f64, b: f64, c: f64, d: f64, material: i32) {
        self.order_kind.push(RTX_PRIM_PLANE);
        self.order_index.push(self.plane_material.len() as i32);
        self.plane_abcd.extend_from_slice(&[a, b, c, d]);
        self.plane_material.push(material);
    }
    /// vertices / normals as Triangle holds them (triangle.rs:89-100), texinfo as (u, v) pairs
    pub fn push_triangle(&mut self, v: &[Vec3; 3], n: &[Vec3; 3], uv: &[(f64, f64); 3], material: i32) {
        self.order_kind.push(RTX_PRIM_TRIANGLE);
        self.order_index.push(self.tri_material.len() as i32);
        for k in 0..3 {
            self.tri_vertices.extend_from_slice(&v3(&v[k]));
            self.tri_normals.extend_from_slice(&v3(&n[k]));
            self.tri_uv.push(uv[k].0);
            self.tri_uv.push(uv[k].1);
        }
        self.tri_material.push(material);
    }
    /// Returns the row of an equal material if one was pushed before.
    pub fn push_material(&mut self, m: RtxMaterial) -> i32 {
        let mut key: Vec<u64> = vec![m.kind as u64, m.texture as u64];
        for x in [m.k_a, m.k_d, m.k_s, m.k_sg, m.k_tg, m.shine, m.gauss_constant, m.glossiness, m.ior].iter() { key.push(x.to_bits()); }
        for a in [m.ambient, m.diffuse, m.specular, m.transmission].iter() { for x in a.iter() { key.push(x.to_bits()); } }
        if let Some(&i) = self.material_index.get(&key) { return i; }
        let i = self.materials.len() as i32;
        self.materials.push(m);
        self.material_index.insert(key, i);
        i
    }
    pub fn push_texture(&mut self, t: RtxTexture) -> i32 { self.textures.push(t); self.textures.len() as i32 - 1 }
    /// Surface as ImageTexture holds it (imagetexture.rs:7-9): RGBA8 rows, row 0 first.
    pub fn push_image(&mut self, width: usize, height: usize, rgba: Vec<u8>) -> i32 {
        self.image_sizes.push((width as i32, height as i32));
        self.image_pixels.push(rgba);
        self.image_pixels.len() as i32 - 1
    }
    pub fn push_light(&mut self, kind: i32, position: &Vec3, color: &Vec3, radius: f64) {
        self.lights.push(RtxLight { kind: kind, reserved: 0, position: v3(position), color: v3(color), radius: radius });
    }

    /// The descriptor borrows this object's vectors: keep `self` (and `images`) alive across rtx_scene_create.
    pub fn desc(&self, images: &mut Vec<RtxImage>) -> RtxSceneDesc {
        images.clear();
        for (px, &(w, h)) in self.image_pixels.iter().zip(self.image_sizes.iter()) {
            images.push(RtxImage { width: w, height: h, rgba: px.as_ptr() });
        }
        RtxSceneDesc {
            n_spheres: self.sphere_radius.len() as i32, sphere_center: self.sphere_center.as_ptr(),
            sphere_radius: self.sphere_radius.as_ptr(), sphere_material: self.sphere_material.as_ptr(),
            n_planes: self.plane_material.len() as i32, plane_abcd: self.plane_abcd.as_ptr(), plane_material: self.plane_material.as_ptr(),
            n_triangles: self.tri_material.len() as i32, tri_vertices: self.tri_vertices.as_ptr(), tri_normals: self.tri_normals.as_ptr(),
            tri_uv: self.tri_uv.as_ptr(), tri_material: self.tri_material.as_ptr(),
            n_prims: self.order_kind.len() as i32, order_kind: self.order_kind.as_ptr(), order_index: self.order_index.as_ptr(),
            n_materials: self.materials.len() as i32, materials: self.materials.as_ptr(),
            n_textures: self.textures.len() as i32, textures: self.textures.as_ptr(),
            n_images: images.len() as i32, images: images.as_ptr(),
            n_lights: self.lights.len() as i32, lights: self.lights.as_ptr(),
            background: self.background, skybox_image: self.skybox_image,
        }
    }
}

// ---- the per-type implementations a maintainer adds next to each type ---------------------------
//
// impl Prim for Sphere   (sphere.rs:36):   fn flatten(&self, out: &mut FlatScene) {
//     let m = self.material.flatten(out); out.push_sphere(&self.center, self.radius, m); }
// impl Prim for Plane    (plane.rs:25):    fn flatten(&self, out: &mut FlatScene) {
//     let m = self.material.flatten(out); out.push_plane(self.a, self.b, self.c, self.d, m); }
// impl Prim for Triangle (triangle.rs:108): fn flatten(&self, out: &mut FlatScene) {
//     let m = self.material.flatten(out);
//     let uv = [(self.texinfo[0].u, self.texinfo[0].v), (self.texinfo[1].u, self.texinfo[1].v), (self.texinfo[2].u, self.texinfo[2].v)];
//     out.push_triangle(&self.vertices, &self.normals, &uv, m); }        // private fields: fine inside triangle.rs
//
// impl Material for CookTorranceMaterial (cooktorrancematerial.rs:25):
//     fn flatten(&self, out: &mut FlatScene) -> i32 {
//         let tex = match self.diffuse_texture { Some(ref t) => t.flatten(out), None => -1 };
//         out.push_material(RtxMaterial { kind: RTX_MAT_COOK_TORRANCE, texture: tex, k_a: self.k_a, k_d: self.k_d, k_s: self.k_s,
//             k_sg: self.k_sg, k_tg: self.k_tg, ambient: v3(&self.ambient), diffuse: v3(&self.diffuse), specular: v3(&self.specular),
//             transmission: v3(&self.transmission), shine: self.roughness, gauss_constant: self.gauss_constant,
//             glossiness: self.glossiness, ior: self.ior }) }
// impl Material for PhongMaterial: the same with kind RTX_MAT_PHONG, shine: self.shininess, gauss_constant: 0.0
// impl Material for FlatMaterial:  kind RTX_MAT_FLAT, diffuse: v3(&self.color), everything else 0, texture -1
//
// impl Texture for CheckerTexture (checkertexture.rs:12): push_texture(RtxTexture { kind: RTX_TEX_CHECKER, image: -1,
//         color1: [c1.r, c1.g, c1.b], color2: [c2.r, c2.g, c2.b], scale: self.scale })
// impl Texture for UVTexture: kind RTX_TEX_UV;   impl Texture for ImageTexture: push_image(w, h, bytes) then kind RTX_TEX_IMAGE
//
// impl Light for PointLight:  out.push_light(RTX_LIGHT_POINT,  &self.position, &self.color, 0.0)
// impl Light for SphereLight: out.push_light(RTX_LIGHT_SPHERE, &self.position, &self.color, self.radius)

/// Scene -> FlatScene (scene.rs:7-12): lights in order (renderer.rs:114 folds in this order), prims in the
/// octree's insertion order, background, the CubeMap's six faces in load order (cubemap.rs:12-19).
pub fn flatten_scene(scene: &::scene::Scene) -> FlatScene {
    let mut out = FlatScene::default();
    out.skybox_image = [-1; 6];
    for light in scene.lights.iter() { light.flatten(&mut out); }
    for prim in scene.octree.prims_in_order() { prim.flatten(&mut out); }
    out.background = v3(&scene.background);
    if let Some(ref cube) = scene.skybox {
        for (f, face) in cube.faces.iter().enumerate() {
            let s = &face.image;
            let mut bytes = Vec::with_capacity(s.width * s.height * 4);
            for px in s.buffer.iter() { bytes.extend_from_slice(&[px.r, px.g, px.b, px.a]); }
            out.skybox_image[f] = out.push_image(s.width, s.height, bytes);
        }
    }
    out
}
