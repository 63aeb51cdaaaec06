//! renderer_gpu.rs — the body of `Renderer::render` (src/raytracer/renderer.rs:30-65) on librtx.so.
//! Drop in as `src/raytracer/gpu/mod.rs`; `Renderer::render` then reads
//!
//!     pub fn render(&self, camera: Camera, shared_scene: Arc<Scene>) -> Surface {
//!         gpu::render(&self.options, &camera, &shared_scene)
//!     }
//!
//! Everything above it (scene files, JSON config, CLI, Animator) stays as it is.  The Animator calls
//! render once per frame (animator.rs:47) with the same Arc<Scene>: the handle is cached per scene so
//! the flattening, the octree build and the upload happen once per animation.
pub mod rtx_ffi;
pub mod flatten;

use std::ffi::CStr;
use std::ptr;
use std::sync::{Arc, Mutex};

use self::rtx_ffi::*;
use raytracer::compositor::{ColorRGBA, Surface};
use raytracer::RenderOptions;
use scene::{Camera, Scene};

/// RAII owner of an rtx_scene handle (what `Arc<Scene>` is on the CPU path).
pub struct GpuScene { handle: *mut RtxScene }
unsafe impl Send for GpuScene {}

impl GpuScene {
    pub fn new(scene: &Scene, device: i32) -> Result<GpuScene, String> {
        let flat = flatten::flatten_scene(scene);
        let mut images = Vec::new();
        let desc = flat.desc(&mut images);
        let mut handle: *mut RtxScene = ptr::null_mut();
        let st = unsafe { rtx_scene_create(&desc, device, &mut handle) };   // copies what it needs; `flat` may go afterwards
        if st != RTX_OK { return Err(last_error(st)); }
        Ok(GpuScene { handle: handle })
    }

    /// One frame: W*H RGBA8 pixels, row 0 = top (Surface.buffer order, surface.rs:18-25).
    pub fn render(&mut self, options: &RenderOptions, camera: &Camera, frame: u32) -> Result<Surface, String> {
        let cam = RtxCamera {
            position: [camera.position.x, camera.position.y, camera.position.z],
            look_at: [camera.look_at.x, camera.look_at.y, camera.look_at.z],
            up: [camera.up.x, camera.up.y, camera.up.z],
            fov_deg: camera.fov_deg,
            image_width: camera.image_width as u32, image_height: camera.image_height as u32,
        };
        let opt = RtxOptions {
            reflect_depth: options.reflect_depth, refract_depth: options.refract_depth,
            shadow_samples: options.shadow_samples, gloss_samples: options.gloss_samples,
            pixel_samples: options.pixel_samples,
            rng_mode: RTX_RNG_PHILOX, seed: 0, frame: frame,      // thread_rng() has no seed to carry over (renderer.rs:69)
            ..Default::default()
        };
        let (w, h) = (camera.image_width as usize, camera.image_height as usize);
        let mut rgba = vec![0u8; w * h * 4];
        let st = unsafe { rtx_render(self.handle, &cam, &opt, rgba.as_mut_ptr(), ptr::null_mut(), ptr::null_mut()) };
        if st != RTX_OK { return Err(last_error(st)); }
        let mut surface = Surface::new(w, h, ColorRGBA::new_rgb(0, 0, 0));
        for (px, c) in surface.buffer.iter_mut().zip(rgba.chunks(4)) {
            *px = ColorRGBA::new_rgba(c[0], c[1], c[2], c[3]);
        }
        Ok(surface)
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) { unsafe { rtx_scene_destroy(self.handle); } }
}

fn last_error(st: RtxStatus) -> String {
    unsafe {
        format!("librtx: {} ({}): {}", CStr::from_ptr(rtx_strerror(st)).to_string_lossy(), st,
                CStr::from_ptr(rtx_last_error()).to_string_lossy())
    }
}

// a plain Once + Mutex<Option<..>> keeps the crate free of new dependencies (no lazy_static)
struct Cache { key: usize, frame: u32, scene: GpuScene }
static mut CACHE: Option<Mutex<Option<Cache>>> = None;
static CACHE_INIT: ::std::sync::Once = ::std::sync::ONCE_INIT;

/// The replacement body of Renderer::render.  Panics on error like the CPU path does (renderer.rs:52).
pub fn render(options: &RenderOptions, camera: &Camera, shared_scene: &Arc<Scene>) -> Surface {
    CACHE_INIT.call_once(|| unsafe { CACHE = Some(Mutex::new(None)); });
    let mut guard = unsafe { CACHE.as_ref().unwrap().lock().unwrap() };
    let key = &**shared_scene as *const Scene as usize;      // same Arc<Scene> => same handle (Animator, animator.rs:47)
    let stale = match *guard { Some(ref c) => c.key != key, None => true };
    if stale {
        let device = ::std::env::var("RTX_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        *guard = Some(Cache { key: key, frame: 0, scene: GpuScene::new(shared_scene, device).expect("rtx_scene_create") });
    }
    let cache = guard.as_mut().unwrap();
    let frame = cache.frame;
    cache.frame += 1;
    cache.scene.render(options, camera, frame).expect("rtx_render")
}
