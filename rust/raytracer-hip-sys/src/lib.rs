//! FFI declarations of `librt_hip.so`: `include/rt_abi.h`, declaration for declaration, in the header's order.
//!
//! `tools/check_rust_abi.py` (run by `tests/test_rust_abi.py`) parses this file and the header and fails when a
//! constant, a struct's field list (names, types, order), a struct's size or a function's argument list differs, so
//! an edit to one side without the other does not survive the CPU test suite.  There is no Rust toolchain in the
//! image this repository is built in: the file is checked against the header, not compiled.
#![allow(non_camel_case_types)]
#![allow(non_upper_case_globals)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 2;

pub const RT_OK: c_int = 0;
pub const RT_E_ABI: c_int = -1;
pub const RT_E_DEVICE: c_int = -2;
pub const RT_E_OOM: c_int = -3;
pub const RT_E_SCENE: c_int = -4;
pub const RT_E_OVERFLOW: c_int = -5;
pub const RT_E_LIMIT: c_int = -6;

// shape kinds: the order of the reference's `Shape` enum (src/shapes.rs:27-36)
pub const RT_SPHERE: u32 = 0;
pub const RT_PLANE: u32 = 1;
pub const RT_CUBE: u32 = 2;
pub const RT_CYLINDER: u32 = 3;
pub const RT_CONE: u32 = 4;
pub const RT_TRIANGLE: u32 = 5;
pub const RT_SMOOTH_TRIANGLE: u32 = 6;
pub const RT_SHAPE_KIND_MASK: u32 = 0xF;
pub const RT_SHAPE_SHADOW: u32 = 0x10;
pub const RT_SHAPE_CLOSED: u32 = 0x20;
pub const RT_VCLASS_NEVER: u32 = 0xFFFF_FFFF;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtXform {
    pub m: [f32; 12],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtShape {
    pub kind_flags: u32,
    pub xform: u32,
    pub material: u32,
    pub parent: i32,
    pub vclass: u32,
    pub prim: u32,
    pub mn: f32,
    pub mx: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtTriGeom {
    pub p1: [f32; 3],
    pub pad0: f32,
    pub e1: [f32; 3],
    pub pad1: f32,
    pub e2: [f32; 3],
    pub pad2: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtTriNormals {
    pub n1: [f32; 3],
    pub pad0: f32,
    pub n2: [f32; 3],
    pub pad1: f32,
    pub n3: [f32; 3],
    pub pad2: f32,
}

pub const RT_NODE_ALL_TRIS: u32 = 0x1;
pub const RT_NODE_PRUNABLE: u32 = 0x2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtNode {
    pub bmin: [f32; 3],
    pub xform: u32,
    pub bmax: [f32; 3],
    pub flags: u32,
    pub first_shape: u32,
    pub n_shapes: u32,
    pub skip: u32,
    pub parent: i32,
    pub first_tri: u32,
    pub next_xf: i32,
    pub idrun: u32,
    pub n_children: u32,
}

// csg ops: src/csg.rs:110-114
pub const RT_CSG_UNION: u32 = 0;
pub const RT_CSG_INTERSECT: u32 = 1;
pub const RT_CSG_DIFFERENCE: u32 = 2;
pub const RT_CSG_LEFT_IS_SHAPE: u32 = 0x1;
pub const RT_CSG_RIGHT_IS_SHAPE: u32 = 0x2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtCsgNode {
    pub bmin: [f32; 3],
    pub op: u32,
    pub bmax: [f32; 3],
    pub flags: u32,
    pub left: i32,
    pub right: i32,
    pub left_lo: u32,
    pub left_hi: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtCsgRoot {
    pub first: u32,
    pub count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtMaterial {
    pub color: [f32; 3],
    pub ambient: f32,
    pub diffuse: f32,
    pub specular: f32,
    pub shininess: f32,
    pub reflective: f32,
    pub transparency: f32,
    pub refractive_index: f32,
    pub pattern: i32,
    pub pad: u32,
}

pub const RT_PAT_STRIPED: u32 = 0;
pub const RT_PAT_GRADIENT: u32 = 1;
pub const RT_PAT_RING: u32 = 2;
pub const RT_PAT_CHECKER: u32 = 3;
pub const RT_PAT_XYZ_RGB: u32 = 4;
pub const RT_PAT_TEXTURE_MAP: u32 = 5;
pub const RT_PAT_CUBE_MAP: u32 = 6;
pub const RT_UVMAP_SPHERICAL: u32 = 0;
pub const RT_UVMAP_PLANAR: u32 = 1;
pub const RT_UVMAP_CYLINDRICAL: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtPattern {
    pub kind: u32,
    pub xform: u32,
    pub uv_mapping: u32,
    pub pad: u32,
    pub a: [f32; 3],
    pub pad_a: f32,
    pub b: [f32; 3],
    pub pad_b: f32,
    pub uv: [i32; 6],
    pub pad2: [u32; 2],
}

pub const RT_UV_CHECKER: u32 = 0;
pub const RT_UV_ALIGN_CHECK: u32 = 1;
pub const RT_UV_IMAGE: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtUvPattern {
    pub kind: u32,
    pub width: f32,
    pub height: f32,
    pub image: i32,
    pub c: [[f32; 3]; 5],
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtImage {
    pub width: u32,
    pub height: u32,
    pub offset: u64,
}

pub const RT_LIGHT_POINT: u32 = 0;
pub const RT_LIGHT_AREA: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtLight {
    pub kind: u32,
    pub usteps: u32,
    pub vsteps: u32,
    pub pad: u32,
    pub intensity: [f32; 3],
    pub pad_i: f32,
    pub position: [f32; 3],
    pub pad_p: f32,
    pub corner: [f32; 3],
    pub pad_c: f32,
    pub uvec: [f32; 3],
    pub pad_u: f32,
    pub vvec: [f32; 3],
    pub pad_v: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtScene {
    pub abi_version: u32,
    pub recursion_limit: u32,
    pub n_xforms: u32,
    pub n_shapes: u32,
    pub n_tris: u32,
    pub n_nodes: u32,
    pub n_csg_nodes: u32,
    pub n_csg_roots: u32,
    pub n_materials: u32,
    pub n_patterns: u32,
    pub n_uvpatterns: u32,
    pub n_images: u32,
    pub n_lights: u32,
    pub n_loose_shapes: u32,
    pub n_texels: u64,
    pub xforms: *const RtXform,
    pub shapes: *const RtShape,
    pub tri_geom: *const RtTriGeom,
    pub tri_normals: *const RtTriNormals,
    pub nodes: *const RtNode,
    pub csg_nodes: *const RtCsgNode,
    pub csg_roots: *const RtCsgRoot,
    pub materials: *const RtMaterial,
    pub patterns: *const RtPattern,
    pub uvpatterns: *const RtUvPattern,
    pub images: *const RtImage,
    pub texels: *const f32,
    pub lights: *const RtLight,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtCamera {
    pub hsize: u32,
    pub vsize: u32,
    pub pixel_size: f32,
    pub half_width: f32,
    pub half_height: f32,
    pub inv_transform: [f32; 12],
}

pub const RT_JITTER_CONST: u32 = 0;
pub const RT_JITTER_COUNTER: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtRenderOptions {
    pub jitter_mode: u32,
    pub jitter_value: f32,
    pub seed: u64,
    pub tile_h: u32,
    pub n_parts: u32,
    pub part: u32,
    pub flags: u32,
}

pub const RT_RENDER_NO_PRUNE: u32 = 0x1;
pub const RT_RENDER_COUNT_WORK: u32 = 0x2;
pub const RT_RENDER_WORK_MAP: u32 = 0x8;
pub const RT_RENDER_SHORT_STACK: u32 = 0x10;
pub const RT_RENDER_NO_ACCEL: u32 = 0x4;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtAux {
    pub hit_mask: *mut u8,
    pub object_id: *mut i32,
    pub hit_t: *mut f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtStats {
    pub primary: u64,
    pub shadow: u64,
    pub reflect: u64,
    pub refract: u64,
    pub overflow: u64,
    pub node_visits: u64,
    pub tri_tests: u64,
    pub shape_tests: u64,
    pub shaded_hits: u64,
    pub node_xforms: u64,
    pub aux_tests: u64,
    pub kernel_ms: f64,
    pub total_ms: f64,
    pub rays_reference: u64,
}

/// One GPU (opaque).
#[repr(C)]
pub struct RtContext {
    _private: [u8; 0],
}
/// A scene resident in that GPU's HBM (opaque).
#[repr(C)]
pub struct RtDeviceScene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtAccelView {
    pub n_nodes: u64,
    pub n_tris: u64,
    pub n_roots: u64,
    pub nodes: *const c_void,
    pub tris: *const c_void,
    pub root_of: *const u32,
    pub gate_next: *const u32,
    pub owner: *mut c_void,
}

// what Scene::render does after Camera::render (src/scene.rs:215-224)
pub const RT_DITHER_NONE: u32 = 0;
pub const RT_DITHER_BAYER2: u32 = 1;
pub const RT_DITHER_BAYER4: u32 = 2;
pub const RT_DITHER_BAYER8: u32 = 3;
pub const RT_DITHER_BAYER16: u32 = 4;
pub const RT_DITHER_BAYER_COLOR: u32 = 5;

pub const RT_STATIC_MAX_RAYS: u32 = 65536;
pub const RT_MULTI_MAX: u32 = 16;

extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_context_create(device: c_int, out: *mut *mut RtContext) -> c_int;
    pub fn rt_context_destroy(ctx: *mut RtContext);
    pub fn rt_last_error(ctx: *const RtContext) -> *const c_char;
    pub fn rt_scene_create(ctx: *mut RtContext, scene: *const RtScene, out: *mut *mut RtDeviceScene) -> c_int;
    pub fn rt_scene_destroy(ctx: *mut RtContext, scene: *mut RtDeviceScene);
    pub fn rt_accel_build(scene: *const RtScene, out: *mut RtAccelView) -> c_int;
    pub fn rt_accel_build_device(ctx: *mut RtContext, scene: *const RtScene, out: *mut RtAccelView) -> c_int;
    pub fn rt_accel_free(view: *mut RtAccelView);
    pub fn rt_part_rows(cam: *const RtCamera, opts: *const RtRenderOptions) -> u32;
    pub fn rt_render(
        ctx: *mut RtContext,
        scene: *const RtDeviceScene,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        out_rgb: *mut f32,
        aux: *mut RtAux,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_render_u8(
        ctx: *mut RtContext,
        scene: *const RtDeviceScene,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        dither: u32,
        out_u8: *mut u8,
        aux: *mut RtAux,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_canvas_to_u8_device(
        ctx: *mut RtContext,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        dither: u32,
        d_rgb: *const f32,
        d_out_u8: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_host_alloc(ctx: *mut RtContext, bytes: u64) -> *mut c_void;
    pub fn rt_host_free(ctx: *mut RtContext, p: *mut c_void);
    pub fn rt_render_device(
        ctx: *mut RtContext,
        scene: *const RtDeviceScene,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        d_out_rgb: *mut f32,
        d_aux: *mut RtAux,
        stream: *mut c_void,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_device_count() -> c_int;
    pub fn rt_render_multi(
        ctxs: *const *mut RtContext,
        scenes: *const *const RtDeviceScene,
        n: u32,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        out_rgb: *mut f32,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_render_multi_device(
        ctxs: *const *mut RtContext,
        scenes: *const *const RtDeviceScene,
        n: u32,
        cam: *const RtCamera,
        opts: *const RtRenderOptions,
        d_out_rgb: *mut f32,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_check(ctx: *mut RtContext, stream: *mut c_void) -> c_int;
    pub fn rt_color_at(
        ctx: *mut RtContext,
        scene: *const RtDeviceScene,
        opts: *const RtRenderOptions,
        origins: *const f32,
        directions: *const f32,
        n: u32,
        remaining: u32,
        out_rgb: *mut f32,
        hit_t: *mut f32,
        hit_id: *mut i32,
        hit_uv: *mut f32,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_color_at_static(
        ctx: *mut RtContext,
        scene: *const RtDeviceScene,
        opts: *const RtRenderOptions,
        origins: *const f32,
        directions: *const f32,
        n: u32,
        remaining: u32,
        out_rgb: *mut f32,
        hit_t: *mut f32,
        hit_id: *mut i32,
        hit_uv: *mut f32,
        stats: *mut RtStats,
    ) -> c_int;
}
