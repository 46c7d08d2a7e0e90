// gpu.rs — bindings for include/rtx.h (librtx_hip.so).  UNTESTED: no Rust toolchain in the build image.
//
// Layouts mirror the C structs field for field; all matrices are 16 f64, column-major, which is
// nalgebra's own storage order (`Matrix4::as_slice()`).

use std::os::raw::c_char;
use std::sync::Once;

pub const RTX_OK: i32 = 0;
pub const NODE_SPHERE: i32 = 0;
pub const NODE_PLANE: i32 = 1;
pub const NODE_CUBE: i32 = 2;
pub const NODE_GROUP: i32 = 3;
pub const NODE_CSG: i32 = 4;
pub const CSG_UNION: i32 = 0;
pub const CSG_INTERSECTION: i32 = 1;
pub const CSG_DIFFERENCE: i32 = 2;
pub const PAT_PLAIN: i32 = 0;
pub const PAT_STRIPES: i32 = 1;
pub const PAT_RINGS: i32 = 2;
pub const PAT_CHECKERS: i32 = 3;
pub const PAT_GRADIENT: i32 = 4;
pub const PAT_TEST: i32 = 5;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxMaterial {
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub color_a: [f64; 3],
    pub color_b: [f64; 3],
    pub pattern_inverse: [f64; 16],
    pub pattern_kind: i32,
    pub _pad: i32,
}

#[repr(C)]
pub struct RtxSceneDesc {
    pub n_nodes: i32,
    pub node_kind: *const i32,
    pub node_inverse: *const f64,
    pub node_bounds: *const f64,
    pub node_first_child: *const i32,
    pub node_child_count: *const i32,
    pub node_csg_op: *const i32,
    pub node_shape_index: *const i32,
    pub node_material: *const i32,
    pub n_children: i32,
    pub children: *const i32,
    pub n_roots: i32,
    pub roots: *const i32,
    pub n_materials: i32,
    pub materials: *const RtxMaterial,
    pub n_lights: i32,
    pub light_position: *const f64,
    pub light_intensity: *const f64,
}

#[repr(C)]
pub struct RtxCamera {
    pub hsize: i32,
    pub vsize: i32,
    pub pixel_size: f64,
    pub half_width: f64,
    pub half_height: f64,
    pub inverse: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtxBands {
    pub first: i32,
    pub stride: i32,
}

#[repr(C)]
pub struct RtxStats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_secondary: u64,
    pub ev: [u64; 24],
    pub overflow_pixels: u64,
    pub kernel_ms: f64,
    pub rays_shadow_skipped: u64,
}

pub enum RtxCtx {}
pub enum RtxScene {}
pub enum RtxComm {}
pub const RTX_COMM_ID_BYTES: usize = 128;

#[link(name = "rtx_hip")]
extern "C" {
    pub fn rtx_init(device: i32, out: *mut *mut RtxCtx) -> i32;
    pub fn rtx_shutdown(ctx: *mut RtxCtx);
    pub fn rtx_scene_create(ctx: *mut RtxCtx, desc: *const RtxSceneDesc, out: *mut *mut RtxScene) -> i32;
    pub fn rtx_scene_destroy(scene: *mut RtxScene);
    pub fn rtx_render(ctx: *mut RtxCtx, scene: *const RtxScene, cam: *const RtxCamera, max_depth: i32,
                      bands: RtxBands, flags: u32, rgb: *mut u8, hit: *mut i32, stats: *mut RtxStats) -> i32;
    pub fn rtx_render_batch(ctx: *mut RtxCtx, scene: *const RtxScene, cams: *const RtxCamera, n_cams: i32,
                            max_depth: i32, flags: u32, rgb: *const *mut u8, hit: *const *mut i32,
                            stats: *mut RtxStats) -> i32;
    pub fn rtx_render_device(ctx: *mut RtxCtx, scene: *const RtxScene, cam: *const RtxCamera, max_depth: i32,
                             bands: RtxBands, flags: u32, d_rgb: *mut std::ffi::c_void, d_hit: *mut std::ffi::c_void,
                             stream: *mut std::ffi::c_void, stats: *mut RtxStats) -> i32;
    pub fn rtx_render_device_noalloc(ctx: *mut RtxCtx, scene: *const RtxScene, cam: *const RtxCamera, max_depth: i32,
                                     bands: RtxBands, flags: u32, d_rgb: *mut std::ffi::c_void,
                                     d_hit: *mut std::ffi::c_void, stream: *mut std::ffi::c_void) -> i32;
    pub fn rtx_ctx_poll(ctx: *mut RtxCtx, wait: i32) -> i32;
    pub fn rtx_set_option(ctx: *mut RtxCtx, name: *const c_char, value: i32) -> i32;
    // one frame on N GPUs: tile bands per GPU + one RCCL all-gather + de-interleave, all inside the library
    pub fn rtx_comm_unique_id(id: *mut std::ffi::c_void) -> i32;
    pub fn rtx_comm_init_rank(ctx: *mut RtxCtx, id: *const std::ffi::c_void, rank: i32, world_size: i32,
                              out: *mut *mut RtxComm) -> i32;
    pub fn rtx_comm_init_all(ctxs: *const *mut RtxCtx, n: i32, out: *mut *mut RtxComm) -> i32;
    pub fn rtx_comm_concurrency(comm: *const RtxComm) -> i32;
    pub fn rtx_comm_init_loopback(ctx: *mut RtxCtx, world_size: i32, out: *mut *mut RtxComm) -> i32;
    pub fn rtx_comm_destroy(comm: *mut RtxComm);
    pub fn rtx_render_gathered(comm: *mut RtxComm, scene: *const RtxScene, cam: *const RtxCamera, max_depth: i32,
                               flags: u32, d_frame: *mut std::ffi::c_void, d_hit_frame: *mut std::ffi::c_void,
                               stream: *mut std::ffi::c_void, stats: *mut RtxStats) -> i32;
    pub fn rtx_comm_wait(comm: *mut RtxComm, stream: *mut std::ffi::c_void) -> i32;
    pub fn rtx_render_multi(comms: *const *mut RtxComm, scenes: *const *const RtxScene, n: i32, cam: *const RtxCamera,
                            max_depth: i32, flags: u32, rgb: *mut u8, hit: *mut i32, stats: *mut RtxStats) -> i32;
    pub fn rtx_strerror(code: i32) -> *const c_char;
    pub fn rtx_last_error(ctx: *const RtxCtx) -> *const c_char;
}

/// The process-wide context on HIP device 0 (the reference is single-threaded, main.rs:11-28; the C ABI is
/// not thread-safe per context either).
pub fn context() -> *mut RtxCtx {
    static INIT: Once = Once::new();
    static mut CTX: *mut RtxCtx = std::ptr::null_mut();
    unsafe {
        INIT.call_once(|| {
            let mut c: *mut RtxCtx = std::ptr::null_mut();
            check(rtx_init(0, &mut c), std::ptr::null());
            CTX = c;
        });
        CTX
    }
}

/// Errors are panics, like the `unwrap()`s of the reference's own I/O (main.rs:21,27).
pub fn check(rc: i32, ctx: *const RtxCtx) {
    if rc != RTX_OK {
        let text = |p: *const c_char| unsafe {
            if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
        };
        let detail = if ctx.is_null() { String::new() } else { text(unsafe { rtx_last_error(ctx) }) };
        panic!("rtx: {} ({})", text(unsafe { rtx_strerror(rc) }), detail);
    }
}
