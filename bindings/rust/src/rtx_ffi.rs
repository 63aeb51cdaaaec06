//! rtx_ffi.rs — raw bindings of `include/rtx.h` (librtx.so, the MI355X trace loop).
//!
//! Drop into the reference crate as `src/raytracer/gpu/rtx_ffi.rs`.  Every `#[repr(C)]` struct below
//! mirrors one struct of rtx.h field for field; `tests/test_rust_shim_layout.py` parses THIS file,
//! computes the repr(C) layout of each struct and compares every size and offset with what gcc
//! reports for rtx.h, and checks that the extern block names exactly the symbols the header declares.
#![allow(dead_code)]
use std::os::raw::{c_char, c_void};

pub type RtxStatus = i32;
pub const RTX_OK: RtxStatus = 0;
pub const RTX_ERR_INVALID_ARGUMENT: RtxStatus = 1;
pub const RTX_ERR_NO_DEVICE: RtxStatus = 2;
pub const RTX_ERR_HIP: RtxStatus = 3;
pub const RTX_ERR_OUT_OF_MEMORY: RtxStatus = 4;
pub const RTX_ERR_LIMIT: RtxStatus = 5;
pub const RTX_ERR_QUEUE_OVERFLOW: RtxStatus = 6;

pub const RTX_MAT_COOK_TORRANCE: i32 = 0;
pub const RTX_MAT_PHONG: i32 = 1;
pub const RTX_MAT_FLAT: i32 = 2;
pub const RTX_TEX_CHECKER: i32 = 0;
pub const RTX_TEX_UV: i32 = 1;
pub const RTX_TEX_IMAGE: i32 = 2;
pub const RTX_LIGHT_POINT: i32 = 0;
pub const RTX_LIGHT_SPHERE: i32 = 1;
pub const RTX_PRIM_SPHERE: i32 = 0;
pub const RTX_PRIM_PLANE: i32 = 1;
pub const RTX_PRIM_TRIANGLE: i32 = 2;
pub const RTX_RNG_CENTER: u32 = 0;
pub const RTX_RNG_PHILOX: u32 = 1;
pub const RTX_FLAG_REFERENCE_ORDER: u32 = 1;
pub const RTX_FLAG_NO_COOP: u32 = 2;
pub const RTX_FLAG_SERIAL: u32 = 4;
pub const RTX_FLAG_OCTREE_WALK: u32 = 8;
pub const RTX_FLAG_COMPACT_OUTPUT: u32 = 16;
pub const RTX_PROBE_FRESNEL: i32 = 0;
pub const RTX_PROBE_REFRACT: i32 = 1;
pub const RTX_PROBE_REFLECT: i32 = 2;
pub const RTX_PROBE_MATERIAL: i32 = 3;
pub const RTX_PROBE_TEXTURE: i32 = 4;
pub const RTX_PROBE_SKYBOX: i32 = 5;
pub const RTX_PROBE_BBOX: i32 = 6;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxMaterial {
    pub kind: i32,
    pub texture: i32,
    pub k_a: f64,
    pub k_d: f64,
    pub k_s: f64,
    pub k_sg: f64,
    pub k_tg: f64,
    pub ambient: [f64; 3],
    pub diffuse: [f64; 3],
    pub specular: [f64; 3],
    pub transmission: [f64; 3],
    pub shine: f64,
    pub gauss_constant: f64,
    pub glossiness: f64,
    pub ior: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxTexture {
    pub kind: i32,
    pub image: i32,
    pub color1: [f64; 3],
    pub color2: [f64; 3],
    pub scale: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxImage {
    pub width: i32,
    pub height: i32,
    pub rgba: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxLight {
    pub kind: i32,
    pub reserved: i32,
    pub position: [f64; 3],
    pub color: [f64; 3],
    pub radius: f64,
}

#[repr(C)]
pub struct RtxSceneDesc {
    pub n_spheres: i32,
    pub sphere_center: *const f64,
    pub sphere_radius: *const f64,
    pub sphere_material: *const i32,
    pub n_planes: i32,
    pub plane_abcd: *const f64,
    pub plane_material: *const i32,
    pub n_triangles: i32,
    pub tri_vertices: *const f64,
    pub tri_normals: *const f64,
    pub tri_uv: *const f64,
    pub tri_material: *const i32,
    pub n_prims: i32,
    pub order_kind: *const i32,
    pub order_index: *const i32,
    pub n_materials: i32,
    pub materials: *const RtxMaterial,
    pub n_textures: i32,
    pub textures: *const RtxTexture,
    pub n_images: i32,
    pub images: *const RtxImage,
    pub n_lights: i32,
    pub lights: *const RtxLight,
    pub background: [f64; 3],
    pub skybox_image: [i32; 6],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxCamera {
    pub position: [f64; 3],
    pub look_at: [f64; 3],
    pub up: [f64; 3],
    pub fov_deg: f64,
    pub image_width: u32,
    pub image_height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtxOptions {
    pub reflect_depth: u32,
    pub refract_depth: u32,
    pub shadow_samples: u32,
    pub gloss_samples: u32,
    pub pixel_samples: u32,
    pub rng_mode: u32,
    pub seed: u64,
    pub frame: u32,
    pub shard_index: u32,
    pub shard_count: u32,
    pub batch_rays: u32,
    pub pool_rays: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtxStats {
    pub rays_primary: u64,
    pub rays_reflect: u64,
    pub rays_gloss: u64,
    pub rays_refract: u64,
    pub rays_shadow: u64,
    pub ms_total: f64,
    pub ms_device: f64,
    pub ms_closest: f64,
    pub ms_shadow: f64,
    pub launches_closest: u32,
    pub launches_shadow: u32,
    pub peak_pool_rays: u64,
    pub work: [[u64; 3]; 2],
    pub exact_shadow_rerenders: u64,
    pub ms_shade: f64,
    pub ms_shadow_light: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxSceneInfo {
    pub n_finite: i32,
    pub n_infinite: i32,
    pub octree_depth: i32,
    pub n_nodes: i32,
    pub n_interior: i32,
    pub n_leaf_entries: i64,
    pub max_leaf_entries: i32,
    pub all_opaque: i32,
    pub device_bytes: i64,
    pub root_min: [f64; 3],
    pub root_max: [f64; 3],
    pub wide_nodes: i32,
    pub wide_depth: i32,
    pub bvh4_nodes: i32,
    pub hierarchy_on_device: i32,
}

/// Opaque handle (`typedef struct rtx_scene rtx_scene`).
pub enum RtxScene {}

#[link(name = "rtx")]
extern "C" {
    pub fn rtx_scene_create(desc: *const RtxSceneDesc, device: i32, out: *mut *mut RtxScene) -> RtxStatus;
    pub fn rtx_scene_destroy(scene: *mut RtxScene) -> RtxStatus;
    pub fn rtx_scene_get_info(scene: *const RtxScene, info: *mut RtxSceneInfo) -> RtxStatus;
    pub fn rtx_render(scene: *mut RtxScene, camera: *const RtxCamera, options: *const RtxOptions,
                      rgba_out: *mut u8, primary_prim_id_out: *mut i32, stats: *mut RtxStats) -> RtxStatus;
    pub fn rtx_render_device(scene: *mut RtxScene, camera: *const RtxCamera, options: *const RtxOptions,
                             rgba_out_dev: *mut c_void, primary_prim_id_out_dev: *mut c_void,
                             stream: *mut c_void, stats: *mut RtxStats) -> RtxStatus;
    pub fn rtx_render_device_noalloc(scene: *mut RtxScene, camera: *const RtxCamera, options: *const RtxOptions,
                                     rgba_out_dev: *mut c_void, primary_prim_id_out_dev: *mut c_void,
                                     stream: *mut c_void, stats: *mut RtxStats) -> RtxStatus;
    pub fn rtx_nearest_hits(scene: *mut RtxScene, n: i64, origin: *const f64, direction: *const f64,
                            prim_id_out: *mut i32, t_out: *mut f64) -> RtxStatus;
    pub fn rtx_nearest_hits_ex(scene: *mut RtxScene, n: i64, origin: *const f64, direction: *const f64,
                               prim_id_out: *mut i32, t_out: *mut f64, flags: u32) -> RtxStatus;
    pub fn rtx_shadow_products(scene: *mut RtxScene, n: i64, origin: *const f64, target: *const f64,
                               shadow_out: *mut f64) -> RtxStatus;
    pub fn rtx_shadow_products_ex(scene: *mut RtxScene, n: i64, origin: *const f64, target: *const f64,
                                  shadow_out: *mut f64, flags: u32) -> RtxStatus;
    pub fn rtx_shade_probe(scene: *mut RtxScene, which: i32, n: i64, input: *const f64, out: *mut f64) -> RtxStatus;
    pub fn rtx_octree_dump(desc: *const RtxSceneDesc, depth: *mut i32, n_nodes: *mut i64, n_entries: *mut i64,
                           bbox_out: *mut f64, shape_out: *mut i32, entries_out: *mut i32) -> RtxStatus;
    pub fn rtx_last_kernel_names(scene: *const RtxScene, out: *mut c_char, cap: i64) -> RtxStatus;
    pub fn rtx_hierarchy_dump(scene: *mut RtxScene, which: i32, out: *mut c_void, cap_bytes: i64, n_bytes: *mut i64) -> RtxStatus;
    pub fn rtx_strerror(status: RtxStatus) -> *const c_char;
    pub fn rtx_last_error() -> *const c_char;
    pub fn rtx_version() -> *const c_char;
}
