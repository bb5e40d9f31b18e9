//! Raw bindings to `include/rt_amd.h` plus a thin safe layer.
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
    pub kernel_ms: f64,
    pub total_ms: f64,
}

extern "C" {
    pub fn rt_init(device: c_int) -> c_int;
    pub fn rt_upload_scene(scene: *const rt_scene) -> c_int;
    pub fn rt_render(frame: *const rt_frame, out_pixels: *mut u32, stats: *mut rt_stats, count_work: c_int) -> c_int;
    pub fn rt_render_tiles(frame: *const rt_frame, shard: u32, nshards: u32, tile_rows: u32, d_out: *mut u32,
                           stats: *mut rt_stats, count_work: c_int) -> c_int;
    pub fn rt_shard_rows(height: u32, shard: u32, nshards: u32, tile_rows: u32) -> u32;
    pub fn rt_scene_blob(device_ptr: *mut *mut c_void, bytes: *mut usize) -> c_int;
    pub fn rt_copy_scene_blob(device_dst: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_adopt_scene_blob(device_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_synchronize() -> c_int;
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
