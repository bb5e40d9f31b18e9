//! Drop-in for the reference's `src/raytracer/` directory: flattens `Config` and hands the render
//! loop to the MI355X library.  Add `mod gpu_render;` to `src/raytracer/mod.rs`, add
//! `rt_amd_sys = { path = "<this repo>/rust/rt_amd_sys" }` to `Cargo.toml`, and call the two
//! functions from `RayTracer::new` (in place of the tree build, raytracer.rs:31-35) and
//! `RayTracer::render` (in place of its body, raytracer.rs:41-78).
//!
//! Not compiled where it was written (no Rust toolchain there).
use crate::imgcomparator::Image;
use crate::raytracer::config::light::Light;
use crate::raytracer::config::shape::Shape;
use crate::raytracer::config::Config;
use rt_amd_sys as sys;
use std::ffi::CString;
use std::ptr;

fn push3(v: &mut Vec<f32>, a: &nalgebra::Vector3<f32>) { v.extend_from_slice(&[a.x, a.y, a.z]); }
fn push_mat(v: &mut Vec<f32>, d: &nalgebra::Vector3<f32>, s: &nalgebra::Vector3<f32>, sh: f32) { push3(v, d); push3(v, s); v.push(sh); }
fn ptr_or_null<T>(v: &Vec<T>) -> *const T { if v.is_empty() { ptr::null() } else { v.as_ptr() } }

/// RayTracer::new: build the tree on the host and make the scene resident on GPU `device`.
pub fn upload(config: &Config, device: i32) -> Result<(), String> {
    let (mut v0, mut v1, mut v2, mut tm, mut to) = (vec![], vec![], vec![], vec![], vec![]);
    let (mut sc, mut sr, mut sm, mut so) = (vec![], vec![], vec![], vec![]);
    let (mut pp, mut pn, mut pm, mut po) = (vec![], vec![], vec![], vec![]);
    for (i, shape) in config.get_scene_objects().iter().enumerate() {
        match shape {
            Shape::Triangle { v0: a, v1: b, v2: c, diffuse_color, specular_color, shininess, .. } => {
                push3(&mut v0, a); push3(&mut v1, b); push3(&mut v2, c);
                push_mat(&mut tm, diffuse_color, specular_color, *shininess); to.push(i as u32);
            }
            Shape::Sphere { center, radius, diffuse_color, specular_color, shininess, .. } => {
                push3(&mut sc, center); sr.push(*radius);
                push_mat(&mut sm, diffuse_color, specular_color, *shininess); so.push(i as u32);
            }
            Shape::Plane { point, normal, diffuse_color, specular_color, shininess, .. } => {
                push3(&mut pp, point); push3(&mut pn, normal);
                push_mat(&mut pm, diffuse_color, specular_color, *shininess); po.push(i as u32);
            }
        }
    }
    let mut lights: Vec<f32> = vec![];
    for l in config.get_lights() {
        match l {
            Light::Point { position, color } => { lights.push(0.0); push3(&mut lights, position); push3(&mut lights, color); }
            Light::Directional { direction, color } => { lights.push(1.0); push3(&mut lights, direction); push3(&mut lights, color); }
        }
    }
    let scene = sys::rt_scene {
        tri_v0: ptr_or_null(&v0), tri_v1: ptr_or_null(&v1), tri_v2: ptr_or_null(&v2),
        sph_center: ptr_or_null(&sc), sph_radius: ptr_or_null(&sr),
        pl_point: ptr_or_null(&pp), pl_normal: ptr_or_null(&pn),
        tri_mat: ptr_or_null(&tm), sph_mat: ptr_or_null(&sm), pl_mat: ptr_or_null(&pm),
        tri_order: ptr_or_null(&to), sph_order: ptr_or_null(&so), pl_order: ptr_or_null(&po),
        ntri: to.len() as u32, nsph: so.len() as u32, npl: po.len() as u32,
        lights: ptr_or_null(&lights), nlights: (lights.len() / 7) as u32,
    };
    unsafe {
        sys::check(sys::rt_init(device))?;
        sys::check(sys::rt_upload_scene(&scene))
    }
}

/// RayTracer::new for a scene whose laid-out form was cached in `cache_path` by an earlier run
/// (`save_cache` below): maps the file, checks it, uploads it; no tree build.  `scene_path` is
/// the text file the cache must still belong to (size and modification time).
pub fn upload_from_cache(cache_path: &str, scene_path: &str, device: i32) -> Result<bool, String> {
    let (cp, sp) = (CString::new(cache_path).map_err(|e| e.to_string())?, CString::new(scene_path).map_err(|e| e.to_string())?);
    unsafe {
        let cache = sys::rth_scene_cache_open(cp.as_ptr());
        if cache.is_null() { return Ok(false); }                       // no cache yet (or a damaged one): build as usual
        let fresh = sys::rth_scene_cache_is_fresh(cache, sp.as_ptr()) == 1;
        let mut result = Ok(false);
        if fresh {
            let mut bytes: usize = 0;
            let blob = sys::rth_scene_cache_blob(cache, &mut bytes);
            result = sys::check(sys::rt_init(device)).and_then(|_| sys::check(sys::rt_upload_scene_blob(blob, bytes))).map(|_| true);
        }
        sys::rth_scene_cache_close(cache);
        result
    }
}

/// Writes the resident scene next to the scene file so that the next start can skip parse and build.
/// `config_handle` is the C-side twin of the Config (rth_load_config_file), which carries the frame settings.
pub fn save_cache(cache_path: &str, scene_path: &str, config_handle: *const sys::rth_config) -> Result<(), String> {
    let (cp, sp) = (CString::new(cache_path).map_err(|e| e.to_string())?, CString::new(scene_path).map_err(|e| e.to_string())?);
    unsafe {
        let (mut dptr, mut bytes) = (ptr::null_mut(), 0usize);
        sys::check(sys::rt_scene_blob(&mut dptr, &mut bytes))?;
        let mut host = vec![0u8; bytes];
        sys::check(sys::rt_download_scene_blob(host.as_mut_ptr() as *mut _, bytes))?;
        sys::check(sys::rth_scene_cache_write(cp.as_ptr(), config_handle, host.as_ptr() as *const _, bytes, sp.as_ptr()))
    }
}

/// RayTracer::render: same frame constants as the CPU loop computes before `par_chunks_mut`,
/// then one call that fills the `Vec<u32>` the `Image` is made of.
pub fn render(config: &Config) -> Result<Image, String> {
    let cv = config.camera.direction().normalize();
    let right = cv.cross(&config.camera.up).normalize();
    let up = right.cross(&cv).normalize();
    let fovrad = config.camera.fov * std::f32::consts::PI / 180.0;
    let pixel_h = (fovrad / 2.0).tan();
    let frame = sys::rt_frame {
        pos: [config.camera.position.x, config.camera.position.y, config.camera.position.z],
        cv: [cv.x, cv.y, cv.z], right: [right.x, right.y, right.z], up: [up.x, up.y, up.z],
        pixel_w: pixel_h * (config.width as f32 / config.height as f32), pixel_h,
        half_w: config.width as f32 / 2.0, half_h: config.height as f32 / 2.0,
        width: config.width, height: config.height, maxdepth: config.maxdepth,
        ambient: [config.ambient.x, config.ambient.y, config.ambient.z],
    };
    let mut data = vec![0u32; config.width as usize * config.height as usize];
    unsafe { sys::check(sys::rt_render(&frame, data.as_mut_ptr(), ptr::null_mut(), 0))?; }
    Ok(Image::new(config.width, config.height, data))
}
