//! Raw bindings to `include/rt_amd.h` (all of it) and to the scene-cache and PNG part of `include/rt_host.h`
//! (its parser and comparator twins are not bound: the Rust host has the originals), plus a thin safe layer.
//!
//! Not compiled where it was written (no Rust toolchain there); the declarations follow the C
//! header field for field.  Every struct is `#[repr(C)]` plain data.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_frame {
    pub pos: [f32; 3],
    pub cv: [f32; 3],
    pub right: [f32; 3],
    pub up: [f32; 3],
    pub pixel_w: f32,
    pub pixel_h: f32,
    pub half_w: f32,
    pub half_h: f32,
    pub width: u32,
    pub height: u32,
    pub maxdepth: u32,
    pub ambient: [f32; 3],
}

#[repr(C)]
pub struct rt_scene {
    pub tri_v0: *const f32,
    pub tri_v1: *const f32,
    pub tri_v2: *const f32,
    pub sph_center: *const f32,
    pub sph_radius: *const f32,
    pub pl_point: *const f32,
    pub pl_normal: *const f32,
    pub tri_mat: *const f32,
    pub sph_mat: *const f32,
    pub pl_mat: *const f32,
    pub tri_order: *const u32,
    pub sph_order: *const u32,
    pub pl_order: *const u32,
    pub ntri: u32,
    pub nsph: u32,
    pub npl: u32,
    pub lights: *const f32,
    pub nlights: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_stats {
    pub primary: u64,
    pub shadow: u64,
    pub secondary: u64,
    pub nodes_visited: u64,
    pub prims_tested: u64,
    pub box_passes: u64,
    pub shape_passes: u64,
    pub phase_passes: [u64; 9],
    pub kernel_ms: f64,
    pub total_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_bvh_info {
    pub node_pairs: u32,
    pub leaves: u32,
    pub max_depth: u32,
    pub max_leaf: u32,
    pub bvh_prims: u32,
    pub list_prims: u32,
    pub build_ms: f64,
    pub sah_cost: f64,
}

pub const RT_BUILD_ON_GPU: u32 = 1;

/// Opaque handles of `include/rt_host.h`.
#[repr(C)]
pub struct rth_config { _private: [u8; 0] }
#[repr(C)]
pub struct rth_scene_cache { _private: [u8; 0] }

extern "C" {
    /// The reference computes these constants itself (raytracer.rs:46-55); this is for callers that do not.
    pub fn rt_frame_from_camera(position: *const f32, look_at: *const f32, up: *const f32, fov_degrees: f32, width: u32, height: u32,
                                maxdepth: u32, ambient: *const f32, out: *mut rt_frame) -> c_int;
    pub fn rt_init(device: c_int) -> c_int;
    pub fn rt_init_multi(ngpus: c_int) -> c_int;
    pub fn rt_device_count() -> c_int;
    pub fn rt_render_frame(frame: *const rt_frame, d_frame: *mut *const u32, stats: *mut rt_stats, count_work: c_int) -> c_int;
    pub fn rt_get_bvh_info(out: *mut rt_bvh_info) -> c_int;
    pub fn rt_upload_scene(scene: *const rt_scene) -> c_int;
    /// `flags`: `RT_BUILD_ON_GPU` builds the tree on the GPU (same split search, an order of magnitude faster for big scenes).
    pub fn rt_upload_scene_ex(scene: *const rt_scene, flags: u32) -> c_int;
    pub fn rt_upload_scene_blob(host_blob: *const c_void, bytes: usize) -> c_int;
    pub fn rt_download_scene_blob(host_dst: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_compare_frames(d_a: *const u32, width_a: u32, height_a: u32, d_b: *const u32, width_b: u32, height_b: u32,
                             d_diff: *mut u32, count: *mut u64) -> c_int;
    pub fn rt_compare_frames_noalloc(d_a: *const u32, width_a: u32, height_a: u32, d_b: *const u32, width_b: u32, height_b: u32,
                             d_diff: *mut u32, d_count: *mut u64) -> c_int;
    pub fn rt_pack_rgb(d_pixels: *const u32, width: u32, height: u32, d_rgb: *mut u8) -> c_int;
    pub fn rt_assemble_tiles(d_gathered: *const u32, rows_per_shard_stride: u32, nshards: u32, tile_rows: u32, width: u32, height: u32,
                             d_frame: *mut u32) -> c_int;
    /// Measured costs of the tile regions of a frame: out after a frame, back in before the first frame of a later run.
    pub fn rt_tile_costs_count(frame: *const rt_frame, gpu: c_int, count: *mut u32) -> c_int;
    pub fn rt_tile_costs_export(frame: *const rt_frame, gpu: c_int, costs: *mut u32, count: u32) -> c_int;
    pub fn rt_tile_costs_import(frame: *const rt_frame, gpu: c_int, costs: *const u32, count: u32) -> c_int;
    pub fn rt_tile_costs_guess(frame: *const rt_frame, gpu: c_int, costs: *mut u32, count: u32) -> c_int;
    /// `RT_LANES` = 4 lanes (stream + work buffers each); launches on different lanes may overlap on the GPU.
    pub fn rt_select_lane(lane: c_int) -> c_int;
    pub fn rt_stream() -> *mut c_void;
    pub fn rth_load_config_file(path: *const c_char) -> *mut rth_config;
    pub fn rth_config_free(c: *mut rth_config);
    pub fn rth_scene_cache_write(path: *const c_char, c: *const rth_config, blob: *const c_void, bytes: usize, source_path: *const c_char) -> c_int;
    pub fn rth_scene_cache_open(path: *const c_char) -> *mut rth_scene_cache;
    pub fn rth_scene_cache_blob(h: *const rth_scene_cache, bytes: *mut usize) -> *const c_void;
    pub fn rth_scene_cache_is_fresh(h: *const rth_scene_cache, source_path: *const c_char) -> c_int;
    pub fn rth_scene_cache_close(h: *mut rth_scene_cache);
    pub fn rth_save_png(path: *const c_char, pixels: *const u32, width: u32, height: u32) -> c_int;
    pub fn rt_render(frame: *const rt_frame, out_pixels: *mut u32, stats: *mut rt_stats, count_work: c_int) -> c_int;
    pub fn rt_render_tiles(frame: *const rt_frame, shard: u32, nshards: u32, tile_rows: u32, d_out: *mut u32,
                           stats: *mut rt_stats, count_work: c_int) -> c_int;
    pub fn rt_shard_rows(height: u32, shard: u32, nshards: u32, tile_rows: u32) -> u32;
    pub fn rt_scene_blob(device_ptr: *mut *mut c_void, bytes: *mut usize) -> c_int;
    pub fn rt_copy_scene_blob(device_dst: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_adopt_scene_blob(device_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_synchronize() -> c_int;
    pub fn rt_wait_event(hip_event: *mut c_void) -> c_int;
    pub fn rt_record_event(hip_event: *mut c_void) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_shutdown();
}

/// `Err(String)` of the failing call, as the reference's `Result<_, String>` expects.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(rt_last_error()).to_string_lossy().into_owned() }
}

pub fn check(rc: c_int) -> Result<(), String> {
    if rc == 0 { Ok(()) } else { Err(last_error()) }
}
