//! `src/gpu.rs` of lerouxrgd/raytracer: the body of `Camera::render` on an MI355X.
//!
//! Flattens `World` into the plain arrays of `include/rt_abi.h` (crate `raytracer-hip-sys`) and calls
//! `rt_scene_create` + `rt_render`.  The tested twin of this file is the C++ host in
//! `raytracer_amd/csrc/host/rt_host.cpp`: every function here names the function there that it mirrors, and follows the
//! same rules in the same order, so the two can be read side by side.  There is no Rust toolchain in the image this
//! repository is built in: this file is source for the maintainer, not a tested build product.
//!
//! Edits to existing reference files that this module needs (visibility only, no behaviour):
//!   * `src/lib.rs`: `pub mod gpu;`
//!   * `src/shapes/{sphere,plane,cube,cylinder,cone,triangle,smooth_triangle}.rs`: fields `pub(super)` -> `pub(crate)`
//!   * `src/groups.rs`: `GroupData` and its fields `pub(crate)`, plus
//!     `pub(crate) fn with_data<R>(&self, f: impl FnOnce(&GroupData) -> R) -> R` (a `GROUPS.read_recursive()` lookup)
//!   * `src/csg.rs`: fields `op`, `left`, `right` of `Csg` `pub(crate)`
//!   * `src/lights.rs`: fields of `PointLight` and `AreaLight` `pub(crate)`
//!   * `src/patterns/uv_pattern.rs`: fields of `UvAlignCheck` `pub(crate)`, plus
//!     `pub(crate) fn with_canvas<R>(&self, f: impl FnOnce(&Canvas) -> R) -> R` on `UvImage` (a `UV_IMAGES.read()` lookup)
//!   * `src/canvas.rs`: `pub(crate) fn buffer_mut(&mut self) -> &mut Rgb32FImage`
//!   * `src/camera.rs`: the body of `render` becomes `crate::gpu::render(self, world).expect("GPU render failed")`
//!   * `Cargo.toml`: `raytracer-hip-sys = { path = "raytracer-hip-sys" }`
use std::collections::HashMap;
use std::ffi::CStr;
use std::ptr;

use raytracer_hip_sys::*;

use crate::bounds::BoundingBox;
use crate::camera::Camera;
use crate::canvas::Canvas;
use crate::csg::{Csg, CsgChild, CsgOp};
use crate::groups::Group;
use crate::lights::Light;
use crate::materials::Material;
use crate::matrices::Matrix;
use crate::patterns::{Pattern, UvMapping, UvPattern};
use crate::shapes::Shape;
use crate::tuples::{Color, Point, Vector};
use crate::world::World;

// ---------------------------------------------------------------------------------------------------------------------
// Value classes (rt_host.cpp: KeyBuilder, ClassTable)
// ---------------------------------------------------------------------------------------------------------------------

/// The value of a shape / material / pattern as words: floats with -0 folded onto +0, a NaN noted.  Equal keys <=> the
/// reference's derived `==` (which compares f32 with `==`: -0 == +0, NaN != NaN).  rt_host.cpp: `KeyBuilder`.
#[derive(Default)]
struct KeyBuilder {
    words: Vec<u32>,
    nan: bool,
}

impl KeyBuilder {
    fn f(&mut self, v: f32) {
        if v.is_nan() {
            self.nan = true;
        }
        self.words.push(if v == 0.0 { 0 } else { v.to_bits() });
    }
    fn fs(&mut self, v: &[f32]) {
        for &x in v {
            self.f(x);
        }
    }
    fn u(&mut self, v: u32) {
        self.words.push(v);
    }
    fn m4(&mut self, m: &Matrix<4, 4>) {
        for r in 0..4 {
            self.fs(&m[r]);
        }
    }
    fn color(&mut self, c: Color) {
        self.fs(&[c.r(), c.g(), c.b()]);
    }
}

/// Value classes, ids in first-seen order; a key that holds a NaN has no class (`RT_VCLASS_NEVER`: the value is not
/// equal to any value, itself included).  rt_host.cpp: `ClassTable`.
#[derive(Default)]
struct ClassTable {
    ids: HashMap<Vec<u32>, u32>,
    next: u32,
}

impl ClassTable {
    fn get(&mut self, k: KeyBuilder) -> u32 {
        if k.nan {
            return RT_VCLASS_NEVER;
        }
        let next = &mut self.next;
        *self.ids.entry(k.words).or_insert_with(|| {
            let id = *next;
            *next += 1;
            id
        })
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small conversions
// ---------------------------------------------------------------------------------------------------------------------

fn p3(p: Point) -> [f32; 3] {
    [p.x(), p.y(), p.z()]
}
fn v3(v: Vector) -> [f32; 3] {
    [v.x(), v.y(), v.z()]
}
fn c3(c: Color) -> [f32; 3] {
    [c.r(), c.g(), c.b()]
}

/// Rows 0..2 of a matrix, row-major: `Mul<Point>` / `Mul<Vector>` never read the w row (src/matrices.rs:220-236).
fn rows012(m: &Matrix<4, 4>) -> [f32; 12] {
    let mut out = [0.0f32; 12];
    for r in 0..3 {
        for c in 0..4 {
            out[r * 4 + c] = m[r][c];
        }
    }
    out
}

fn is_identity(m: &Matrix<4, 4>) -> bool {
    // rt_host.cpp: m4_is_identity -- value comparison, so -0 counts as 0 like everywhere else
    *m == Matrix::identity()
}

/// Conservative box arithmetic for the RT_NODE_PRUNABLE flag (rt_host.cpp: `Box`, `box_merge`, `box_transform`); the
/// box that goes into `RtNode::bmin/bmax` is never computed here: it is `Group::bounds()` itself.
#[derive(Clone, Copy)]
struct TrueBox {
    lo: [f32; 3],
    hi: [f32; 3],
}

impl TrueBox {
    fn empty() -> Self {
        Self { lo: [f32::INFINITY; 3], hi: [f32::NEG_INFINITY; 3] }
    }
    fn everything() -> Self {
        Self { lo: [f32::NEG_INFINITY; 3], hi: [f32::INFINITY; 3] }
    }
    fn of(bb: BoundingBox) -> Self {
        Self { lo: p3(bb.min), hi: p3(bb.max) }
    }
    fn add(&mut self, p: [f32; 3]) {
        for i in 0..3 {
            if p[i] < self.lo[i] {
                self.lo[i] = p[i];
            }
            if p[i] > self.hi[i] {
                self.hi[i] = p[i];
            }
        }
    }
    fn merge(&mut self, o: &TrueBox) {
        self.add(o.lo);
        self.add(o.hi);
    }
    fn transform(&self, t: Matrix<4, 4>) -> Self {
        let bb = BoundingBox::default()
            .with_min(Point::new(self.lo[0], self.lo[1], self.lo[2]))
            .with_max(Point::new(self.hi[0], self.hi[1], self.hi[2]));
        Self::of(bb.transform(t)) // the reference's eight-corner transform, src/bounds.rs:36-53
    }
    fn is_empty(&self) -> bool {
        self.lo[0] > self.hi[0] && self.lo[1] > self.hi[1] && self.lo[2] > self.hi[2]
    }
    /// false if any coordinate is a NaN
    fn inside(&self, bb: &BoundingBox) -> bool {
        let (lo, hi) = (p3(bb.min), p3(bb.max));
        (0..3).all(|i| self.lo[i] >= lo[i] && self.hi[i] <= hi[i])
    }
}

/// What a shape really covers in its parent's space (rt_host.cpp: `true_psb`): the reference's box for every shape but
/// the cone, whose side hit for rays parallel to the side skips the min/max test (cone.rs:131-141) and whose caps have
/// radius sqrt|y| (cone.rs:90-94) -- a cone is treated as unbounded.
fn true_psb(s: &Shape) -> TrueBox {
    match s {
        Shape::Cone(_) => TrueBox::everything(),
        _ => TrueBox::of(s.parent_space_bounds()),
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The flattened world (rt_host.cpp: Flat, Flattener)
// ---------------------------------------------------------------------------------------------------------------------

#[derive(Default)]
pub struct Flat {
    pub xforms: Vec<RtXform>,
    pub shapes: Vec<RtShape>,
    pub tri_geom: Vec<RtTriGeom>,
    pub tri_normals: Vec<RtTriNormals>,
    pub nodes: Vec<RtNode>,
    pub csg_nodes: Vec<RtCsgNode>,
    pub csg_roots: Vec<RtCsgRoot>,
    pub materials: Vec<RtMaterial>,
    pub patterns: Vec<RtPattern>,
    pub uvpatterns: Vec<RtUvPattern>,
    pub images: Vec<RtImage>,
    pub texels: Vec<f32>,
    pub lights: Vec<RtLight>,
    pub n_loose_shapes: u32,
    pub recursion_limit: u32,
}

#[derive(Default)]
struct Flattener {
    f: Flat,
    xform_ids: HashMap<[u32; 12], u32>,
    uv_classes: ClassTable,
    pattern_classes: ClassTable,
    material_classes: ClassTable,
    shape_classes: ClassTable,
    image_slot: HashMap<crate::patterns::UvImage, i32>,
    uv_slot: HashMap<u32, i32>,        // uv pattern class -> uvpatterns[] index
    pattern_slot: HashMap<u32, i32>,   // pattern class -> patterns[] index
    material_slot: HashMap<u32, u32>,  // material class -> materials[] index
    /// triangle data already emitted, by the bytes of a node's whole run of own triangles: instances of one mesh,
    /// divided the same way, share their tri_geom / tri_normals entries (rt_host.cpp: `geometry_run`, which keys the run
    /// by (mesh, triangle indices) because its builder knows the mesh; the reference re-parses the OBJ file per
    /// instance, scene.rs:593-599, so here the content is the key)
    runs: HashMap<Vec<u32>, u32>,
}

impl Flattener {
    /// rt_host.cpp: `xform_index`.  Entry 0 is the identity; equal inverses share an entry.
    fn xform_index(&mut self, t: &Matrix<4, 4>) -> u32 {
        if is_identity(t) {
            return 0;
        }
        // exact cofactor inverse (src/matrices.rs:185-197); the reference unwraps it at every use
        let inv = t.inverse().expect("transform is not invertible");
        let rows = rows012(&inv);
        let mut key = [0u32; 12];
        for i in 0..12 {
            key[i] = rows[i].to_bits();
        }
        if let Some(&id) = self.xform_ids.get(&key) {
            return id;
        }
        self.f.xforms.push(RtXform { m: rows });
        let id = (self.f.xforms.len() - 1) as u32;
        self.xform_ids.insert(key, id);
        id
    }

    /// rt_host.cpp: `tables()`, image part.  A `UvImage` is a key into UV_IMAGES (uv_pattern.rs:109-117); the canvas is
    /// copied once per key.
    fn image(&mut self, im: crate::patterns::UvImage) -> i32 {
        if let Some(&i) = self.image_slot.get(&im) {
            return i;
        }
        let offset = self.f.texels.len() as u64;
        let (w, h) = im.with_canvas(|c: &Canvas| {
            for y in 0..c.height() {
                for x in 0..c.width() {
                    self.f.texels.extend_from_slice(&c3(c.pixel_at(x, y)));
                }
            }
            (c.width(), c.height())
        });
        self.f.images.push(RtImage { width: w, height: h, offset });
        let i = (self.f.images.len() - 1) as i32;
        self.image_slot.insert(im, i);
        i
    }

    /// rt_host.cpp: `tables()`, uv-pattern part.  Returns (uvpatterns[] index, value class).
    fn uv_pattern(&mut self, uv: &UvPattern) -> (i32, u32) {
        let mut r = RtUvPattern::default();
        r.image = -1;
        let mut k = KeyBuilder::default();
        match uv {
            UvPattern::Checker(c) => {
                r.kind = RT_UV_CHECKER;
                r.width = c.width;
                r.height = c.height;
                r.c[0] = c3(c.a);
                r.c[1] = c3(c.b);
                k.u(RT_UV_CHECKER);
                k.f(c.width);
                k.f(c.height);
                k.color(c.a);
                k.color(c.b);
            }
            UvPattern::AlignCheck(a) => {
                r.kind = RT_UV_ALIGN_CHECK;
                k.u(RT_UV_ALIGN_CHECK);
                for (i, c) in [a.main, a.ul, a.ur, a.bl, a.br].into_iter().enumerate() {
                    r.c[i] = c3(c);
                    k.color(c);
                }
            }
            UvPattern::Image(im) => {
                r.kind = RT_UV_IMAGE;
                r.image = self.image(*im);
                k.u(RT_UV_IMAGE);
                k.u(r.image as u32); // two UvImage keys are equal iff they are the same key
            }
        }
        let class = self.uv_classes.get(k);
        if class != RT_VCLASS_NEVER {
            if let Some(&i) = self.uv_slot.get(&class) {
                return (i, class);
            }
        }
        self.f.uvpatterns.push(r);
        let i = (self.f.uvpatterns.len() - 1) as i32;
        if class != RT_VCLASS_NEVER {
            self.uv_slot.insert(class, i);
        }
        (i, class)
    }

    /// rt_host.cpp: `tables()`, pattern part.  Returns (patterns[] index, value class).
    fn pattern(&mut self, p: &Pattern) -> (i32, u32) {
        let t = p.get_transform();
        let mut r = RtPattern::default();
        r.uv = [-1; 6];
        r.xform = self.xform_index(&t); // pattern_at_shape multiplies by the inverse (patterns.rs:84)
        let mut k = KeyBuilder::default();
        let two = |r: &mut RtPattern, k: &mut KeyBuilder, kind: u32, a: Color, b: Color| {
            r.kind = kind;
            r.a = c3(a);
            r.b = c3(b);
            k.u(kind);
            k.m4(&t);
            k.color(a);
            k.color(b);
        };
        match p {
            Pattern::Striped(s) => two(&mut r, &mut k, RT_PAT_STRIPED, s.a, s.b),
            Pattern::Gradient(s) => two(&mut r, &mut k, RT_PAT_GRADIENT, s.a, s.b),
            Pattern::Ring(s) => two(&mut r, &mut k, RT_PAT_RING, s.a, s.b),
            Pattern::Checker(s) => two(&mut r, &mut k, RT_PAT_CHECKER, s.a, s.b),
            Pattern::XyzRgb(_) => {
                r.kind = RT_PAT_XYZ_RGB;
                k.u(RT_PAT_XYZ_RGB);
                k.m4(&t);
            }
            Pattern::TextureMap(tm) => {
                r.kind = RT_PAT_TEXTURE_MAP;
                r.uv_mapping = match tm.uv_mapping {
                    UvMapping::Spherical => RT_UVMAP_SPHERICAL,
                    UvMapping::Planar => RT_UVMAP_PLANAR,
                    UvMapping::Cylindrical => RT_UVMAP_CYLINDRICAL,
                };
                let (i, class) = self.uv_pattern(&tm.uv_pattern);
                r.uv[0] = i;
                k.u(RT_PAT_TEXTURE_MAP);
                k.m4(&t);
                k.u(r.uv_mapping);
                k.u(class);
                k.nan |= class == RT_VCLASS_NEVER;
            }
            Pattern::CubeMap(cm) => {
                r.kind = RT_PAT_CUBE_MAP;
                k.u(RT_PAT_CUBE_MAP);
                k.m4(&t);
                // left, front, right, back, up, down (cube_map.rs:11-19)
                for (slot, uv) in [&cm.left, &cm.front, &cm.right, &cm.back, &cm.up, &cm.down].into_iter().enumerate() {
                    let (i, class) = self.uv_pattern(uv);
                    r.uv[slot] = i;
                    k.u(class);
                    k.nan |= class == RT_VCLASS_NEVER;
                }
            }
        }
        let class = self.pattern_classes.get(k);
        if class != RT_VCLASS_NEVER {
            if let Some(&i) = self.pattern_slot.get(&class) {
                return (i, class);
            }
        }
        self.f.patterns.push(r);
        let i = (self.f.patterns.len() - 1) as i32;
        if class != RT_VCLASS_NEVER {
            self.pattern_slot.insert(class, i);
        }
        (i, class)
    }

    /// rt_host.cpp: `material_slow` (its `material` wrapper is a one-entry memo of this).  Returns (materials[] index,
    /// value class); equal materials share an entry.
    fn material(&mut self, m: &Material) -> (u32, u32) {
        let mut k = KeyBuilder::default();
        k.color(m.color);
        k.fs(&[m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index]);
        let mut pattern = -1;
        match &m.pattern {
            Some(p) => {
                let (i, class) = self.pattern(p);
                pattern = i;
                k.nan |= class == RT_VCLASS_NEVER;
                k.u(1);
                k.u(class);
            }
            None => {
                k.u(0);
                k.u(0);
            }
        }
        let class = self.material_classes.get(k);
        if class != RT_VCLASS_NEVER {
            if let Some(&i) = self.material_slot.get(&class) {
                return (i, class);
            }
        }
        self.f.materials.push(RtMaterial {
            color: c3(m.color),
            ambient: m.ambient,
            diffuse: m.diffuse,
            specular: m.specular,
            shininess: m.shininess,
            reflective: m.reflective,
            transparency: m.transparency,
            refractive_index: m.refractive_index,
            pattern,
            pad: 0,
        });
        let i = (self.f.materials.len() - 1) as u32;
        if class != RT_VCLASS_NEVER {
            self.material_slot.insert(class, i);
        }
        (i, class)
    }

    /// rt_host.cpp: `fill_tri`.  Triangle::new's p1, e1, e2, normal (triangle.rs:32-48) / the three vertex normals
    /// (smooth_triangle.rs:24-40), copied, never recomputed.
    fn tri_data(s: &Shape) -> Option<(RtTriGeom, RtTriNormals)> {
        match s {
            Shape::Triangle(t) => Some((
                RtTriGeom { p1: p3(t.p1), e1: v3(t.e1), e2: v3(t.e2), ..Default::default() },
                RtTriNormals { n1: v3(t.normal), ..Default::default() },
            )),
            Shape::SmoothTriangle(t) => Some((
                RtTriGeom { p1: p3(t.p1), e1: v3(t.e1), e2: v3(t.e2), ..Default::default() },
                RtTriNormals { n1: v3(t.n1), n2: v3(t.n2), n3: v3(t.n3), ..Default::default() },
            )),
            _ => None,
        }
    }

    /// rt_host.cpp: `emit_shape`.  `prim` = Some(index) when the caller has already placed the triangle's data (a node's
    /// shared run).  Returns the shape's index = its insertion rank = its object id.
    fn emit_shape(&mut self, s: &Shape, parent_node: i32, prim: Option<u32>) -> u32 {
        let t = s.get_transform();
        let (material, mclass) = self.material(&s.get_material());
        let kind = match s {
            Shape::Sphere(_) => RT_SPHERE,
            Shape::Plane(_) => RT_PLANE,
            Shape::Cube(_) => RT_CUBE,
            Shape::Cylinder(_) => RT_CYLINDER,
            Shape::Cone(_) => RT_CONE,
            Shape::Triangle(_) => RT_TRIANGLE,
            Shape::SmoothTriangle(_) => RT_SMOOTH_TRIANGLE,
        };
        let mut r = RtShape {
            kind_flags: kind | if s.shadow() { RT_SHAPE_SHADOW } else { 0 },
            xform: self.xform_index(&t),
            material,
            parent: parent_node,
            vclass: 0,
            prim: 0,
            mn: f32::NEG_INFINITY,
            mx: f32::INFINITY,
        };
        // the shape's value, parent left out (`#[derivative(PartialEq = "ignore")]` on every `parent` field)
        let mut k = KeyBuilder::default();
        k.u(kind);
        k.m4(&t);
        k.u(mclass);
        k.u(s.shadow() as u32);
        k.nan |= mclass == RT_VCLASS_NEVER;
        match s {
            Shape::Cylinder(c) => {
                r.mn = c.min;
                r.mx = c.max;
                r.kind_flags |= if c.closed { RT_SHAPE_CLOSED } else { 0 };
                k.fs(&[c.min, c.max]); // -inf / +inf are equal to themselves: their bits are kept
                k.u(c.closed as u32);
            }
            Shape::Cone(c) => {
                r.mn = c.min;
                r.mx = c.max;
                r.kind_flags |= if c.closed { RT_SHAPE_CLOSED } else { 0 };
                k.fs(&[c.min, c.max]);
                k.u(c.closed as u32);
            }
            // Triangle == compares p1, p2, p3, e1, e2 and normal / n1..n3 (triangle.rs:14-27, smooth_triangle.rs:12-26)
            Shape::Triangle(t) => {
                for v in [p3(t.p1), p3(t.p2), p3(t.p3), v3(t.e1), v3(t.e2), v3(t.normal)] {
                    k.fs(&v);
                }
            }
            Shape::SmoothTriangle(t) => {
                for v in [p3(t.p1), p3(t.p2), p3(t.p3), v3(t.n1), v3(t.n2), v3(t.n3), v3(t.e1), v3(t.e2)] {
                    k.fs(&v);
                }
            }
            _ => {}
        }
        if let Some((g, n)) = Self::tri_data(s) {
            r.prim = match prim {
                Some(p) => p,
                None => {
                    self.f.tri_geom.push(g);
                    self.f.tri_normals.push(n);
                    (self.f.tri_geom.len() - 1) as u32
                }
            };
        }
        r.vclass = self.shape_classes.get(k);
        self.f.shapes.push(r);
        (self.f.shapes.len() - 1) as u32
    }

    /// rt_host.cpp: `emit_csg`.  Leaves are emitted in the order Csg::local_intersect gathers them (a shape child before
    /// a csg child, otherwise left then right; csg.rs:32-46): that order is their tie-break rank.  Nodes are stored in
    /// post-order (children before parents).  Returns (node index, [lo, hi) leaf shape range of the subtree).
    fn emit_csg(&mut self, c: &Csg) -> (i32, u32, u32) {
        let mut r = RtCsgNode::default();
        r.op = match c.op {
            CsgOp::Union => RT_CSG_UNION,
            CsgOp::Intersect => RT_CSG_INTERSECT,
            CsgOp::Difference => RT_CSG_DIFFERENCE,
        };
        let mut child = |this: &mut Self, ch: &CsgChild, flag: u32, flags: &mut u32| -> (i32, u32, u32) {
            match ch {
                CsgChild::Shape(s) => {
                    let i = this.emit_shape(s, -1, None);
                    *flags |= flag;
                    (i as i32, i, i + 1)
                }
                CsgChild::Csg(sub) => this.emit_csg(sub),
            }
        };
        let (l, r_);
        if matches!((&c.left, &c.right), (CsgChild::Csg(_), CsgChild::Shape(_))) {
            r_ = child(self, &c.right, RT_CSG_RIGHT_IS_SHAPE, &mut r.flags);
            l = child(self, &c.left, RT_CSG_LEFT_IS_SHAPE, &mut r.flags);
        } else {
            l = child(self, &c.left, RT_CSG_LEFT_IS_SHAPE, &mut r.flags);
            r_ = child(self, &c.right, RT_CSG_RIGHT_IS_SHAPE, &mut r.flags);
        }
        r.left = l.0;
        r.right = r_.0;
        r.left_lo = l.1; // Csg::includes (csg.rs:74-85) becomes a range test on the shape index
        r.left_hi = l.2;
        let bb = c.bounds(); // csg.rs:87-105, the reference's own box
        r.bmin = p3(bb.min);
        r.bmax = p3(bb.max);
        self.f.csg_nodes.push(r);
        ((self.f.csg_nodes.len() - 1) as i32, l.1.min(r_.1), l.2.max(r_.2))
    }

    /// rt_host.cpp: `emit_group`.  Groups depth-first, a group's own shapes before its children (groups.rs:118-123).
    /// Returns the conservative box of the subtree in the group's own space.
    fn emit_group(&mut self, g: Group, parent_node: i32) -> TrueBox {
        let idx = self.f.nodes.len() as u32;
        self.f.nodes.push(RtNode::default());
        let (transform, shapes, children): (Matrix<4, 4>, Vec<Shape>, Vec<Group>) =
            g.with_data(|d| (d.transform, d.shapes.clone(), d.children.clone()));
        let xf = self.xform_index(&transform);
        let first_shape = self.f.shapes.len() as u32;
        let mut first_tri = self.f.tri_geom.len() as u32;
        // RT_NODE_ALL_TRIS: every own shape is an identity-transform (smooth) triangle
        let all_tris = !shapes.is_empty()
            && shapes.iter().all(|s| {
                matches!(s, Shape::Triangle(_) | Shape::SmoothTriangle(_)) && is_identity(&s.get_transform())
            });
        let mut tb = TrueBox::empty();
        if all_tris {
            // the node's triangles are one contiguous run of tri_geom / tri_normals; a run with the same bytes that is
            // already there (another instance of the mesh) is shared
            let data: Vec<(RtTriGeom, RtTriNormals)> = shapes.iter().map(|s| Self::tri_data(s).unwrap()).collect();
            let mut key: Vec<u32> = Vec::with_capacity(data.len() * 18);
            for (g, n) in &data {
                for v in [g.p1, g.e1, g.e2, n.n1, n.n2, n.n3] {
                    key.extend(v.iter().map(|x| x.to_bits()));
                }
            }
            first_tri = match self.runs.get(&key) {
                Some(&first) => first,
                None => {
                    let first = self.f.tri_geom.len() as u32;
                    for (g, n) in data {
                        self.f.tri_geom.push(g);
                        self.f.tri_normals.push(n);
                    }
                    self.runs.insert(key, first);
                    first
                }
            };
        }
        for (k, s) in shapes.iter().enumerate() {
            self.emit_shape(s, idx as i32, if all_tris { Some(first_tri + k as u32) } else { None });
            tb.merge(&true_psb(s));
        }
        {
            let (parent_next, parent_run) = if parent_node >= 0 {
                let p = &self.f.nodes[parent_node as usize];
                (p.next_xf, p.idrun)
            } else {
                (-1, 0)
            };
            let n = &mut self.f.nodes[idx as usize];
            n.xform = xf;
            n.first_shape = first_shape;
            n.n_shapes = shapes.len() as u32;
            n.parent = parent_node;
            n.first_tri = first_tri;
            n.n_children = children.len() as u32;
            // the chain Shape::normal_to_world / world_to_object climbs (groups.rs:138-164): the nearest node at or
            // above this one with a transform of its own, and the number of identity nodes passed on the way up
            if xf != 0 {
                n.next_xf = idx as i32;
                n.idrun = 0;
            } else {
                n.next_xf = parent_next;
                n.idrun = 1 + parent_run;
            }
        }
        for c in &children {
            let ctb = self.emit_group(*c, idx as i32);
            let ct = c.get_transform();
            if is_identity(&ct) {
                tb.merge(&ctb);
            } else {
                tb.merge(&ctb.transform(ct));
            }
        }
        // The box the reference tests (groups.rs:111): the cached one for groups that went through cache_bounds
        // (scene.rs:197), make_bounds otherwise -- quirks included (groups.rs:176-188 re-transforms the running box once
        // per child of a top-level group).  Never a "corrected" box: SURVEY.md Q6.
        let bb = g.bounds();
        let n = &mut self.f.nodes[idx as usize];
        n.bmin = p3(bb.min);
        n.bmax = p3(bb.max);
        n.skip = self.f.nodes.len() as u32;
        // distance pruning is only allowed where the copied box really encloses what is under it
        let prunable = tb.inside(&bb) || tb.is_empty();
        n.flags = if all_tris { RT_NODE_ALL_TRIS } else { 0 } | if prunable { RT_NODE_PRUNABLE } else { 0 };
        tb
    }

    fn light(l: &Light) -> RtLight {
        match l {
            Light::PointLight(p) => RtLight {
                kind: RT_LIGHT_POINT,
                usteps: 1,
                vsteps: 1,
                intensity: c3(p.intensity),
                position: p3(p.position),
                ..Default::default()
            },
            // uvec / vvec are already divided by the step counts (lights.rs:94-101)
            Light::AreaLight(a) => RtLight {
                kind: RT_LIGHT_AREA,
                usteps: a.usteps as u32,
                vsteps: a.vsteps as u32,
                intensity: c3(a.intensity),
                position: p3(a.position),
                corner: p3(a.corner),
                uvec: v3(a.uvec),
                vvec: v3(a.vvec),
                ..Default::default()
            },
        }
    }

    /// rt_host.cpp: `run`.  The gather order of World::intersect (world.rs:44-62): csg leaves, then groups, then the
    /// world's loose shapes.
    fn run(mut self, world: &World) -> Flat {
        let mut identity = [0.0f32; 12];
        identity[0] = 1.0;
        identity[5] = 1.0;
        identity[10] = 1.0;
        self.f.xforms.push(RtXform { m: identity });
        for c in &world.csgs {
            let first = self.f.csg_nodes.len() as u32;
            self.emit_csg(c);
            self.f.csg_roots.push(RtCsgRoot { first, count: self.f.csg_nodes.len() as u32 - first });
        }
        for g in &world.groups {
            self.emit_group(*g, -1);
        }
        let loose0 = self.f.shapes.len();
        for s in &world.shapes {
            self.emit_shape(s, -1, None);
        }
        self.f.n_loose_shapes = (self.f.shapes.len() - loose0) as u32;
        self.f.lights = world.lights.iter().map(Self::light).collect();
        self.f.recursion_limit = world.recursion_limit as u32;
        self.f
    }
}

impl Flat {
    pub fn of(world: &World) -> Flat {
        Flattener::default().run(world)
    }

    /// The pointers are valid while `self` is alive and unchanged; rt_scene_create copies everything.
    pub fn view(&self) -> RtScene {
        RtScene {
            abi_version: RT_ABI_VERSION,
            recursion_limit: self.recursion_limit,
            n_xforms: self.xforms.len() as u32,
            n_shapes: self.shapes.len() as u32,
            n_tris: self.tri_geom.len() as u32,
            n_nodes: self.nodes.len() as u32,
            n_csg_nodes: self.csg_nodes.len() as u32,
            n_csg_roots: self.csg_roots.len() as u32,
            n_materials: self.materials.len() as u32,
            n_patterns: self.patterns.len() as u32,
            n_uvpatterns: self.uvpatterns.len() as u32,
            n_images: self.images.len() as u32,
            n_lights: self.lights.len() as u32,
            n_loose_shapes: self.n_loose_shapes,
            n_texels: self.texels.len() as u64,
            xforms: self.xforms.as_ptr(),
            shapes: self.shapes.as_ptr(),
            tri_geom: self.tri_geom.as_ptr(),
            tri_normals: self.tri_normals.as_ptr(),
            nodes: self.nodes.as_ptr(),
            csg_nodes: self.csg_nodes.as_ptr(),
            csg_roots: self.csg_roots.as_ptr(),
            materials: self.materials.as_ptr(),
            patterns: self.patterns.as_ptr(),
            uvpatterns: self.uvpatterns.as_ptr(),
            images: self.images.as_ptr(),
            texels: self.texels.as_ptr(),
            lights: self.lights.as_ptr(),
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Camera and the call
// ---------------------------------------------------------------------------------------------------------------------

/// rt_host.cpp: `make_rt_camera`.  pixel_size / half_width / half_height are the values Camera::new already derived
/// (camera.rs:19-38, tan on the host libm); the inverse is the one ray_for_pixel unwraps per pixel (camera.rs:58).
fn rt_camera(cam: &Camera) -> Result<RtCamera, String> {
    let inv = cam.transform.inverse().ok_or("camera transform is not invertible")?;
    Ok(RtCamera {
        hsize: cam.hsize,
        vsize: cam.vsize,
        pixel_size: cam.pixel_size,
        half_width: cam.half_width,
        half_height: cam.half_height,
        inv_transform: rows012(&inv),
    })
}

/// One context per GPU of the host, created on first use.  RT_HIP_DEVICES lists the ordinals to use ("0,2,3"),
/// RT_HIP_DEVICE names a single one; without either every device rt_device_count() reports is used, and a frame is
/// rendered by all of them at once (rt_render_multi: GPU g renders the rows r with (r / 8) % n == g).
struct Gpus(Vec<*mut RtContext>);
unsafe impl Send for Gpus {}

impl Drop for Gpus {
    fn drop(&mut self) {
        for &ctx in &self.0 {
            unsafe { rt_context_destroy(ctx) }
        }
    }
}

lazy_static::lazy_static! {
    static ref GPU: parking_lot::Mutex<Option<Gpus>> = parking_lot::Mutex::new(None);
}

fn device_list() -> Result<Vec<i32>, String> {
    if let Ok(list) = std::env::var("RT_HIP_DEVICES") {
        let devices: Vec<i32> = list.split(',').filter_map(|s| s.trim().parse().ok()).collect();
        if devices.is_empty() {
            return Err("RT_HIP_DEVICES names no device".into());
        }
        return Ok(devices);
    }
    if let Some(device) = std::env::var("RT_HIP_DEVICE").ok().and_then(|s| s.parse().ok()) {
        return Ok(vec![device]);
    }
    let n = unsafe { rt_device_count() };
    if n <= 0 {
        return Err("no GPU visible (rt_device_count)".into());
    }
    Ok((0..n.min(RT_MULTI_MAX as i32)).collect())
}

/// The contexts of this process, creating them on the first call.
fn contexts(guard: &mut Option<Gpus>) -> Result<Vec<*mut RtContext>, String> {
    if guard.is_none() {
        let mut gpus = Gpus(Vec::new());
        for device in device_list()? {
            let mut ctx = ptr::null_mut();
            check(ptr::null_mut(), unsafe { rt_context_create(device, &mut ctx) }, "rt_context_create")?;
            gpus.0.push(ctx);
        }
        *guard = Some(gpus);
    }
    Ok(guard.as_ref().unwrap().0.clone())
}

/// The flattened world in every context's HBM; `f` runs with the scene handles, which are destroyed afterwards.
fn with_scenes<T>(
    ctxs: &[*mut RtContext],
    scene: &RtScene,
    f: impl FnOnce(&[*const RtDeviceScene]) -> Result<T, String>,
) -> Result<T, String> {
    let mut scenes: Vec<*mut RtDeviceScene> = Vec::with_capacity(ctxs.len());
    let mut result = Ok(());
    for &ctx in ctxs {
        let mut d = ptr::null_mut();
        result = check(ctx, unsafe { rt_scene_create(ctx, scene, &mut d) }, "rt_scene_create");
        if result.is_err() {
            break;
        }
        scenes.push(d);
    }
    let out = match result {
        Ok(()) => {
            let views: Vec<*const RtDeviceScene> = scenes.iter().map(|&d| d as *const RtDeviceScene).collect();
            f(&views)
        }
        Err(e) => Err(e),
    };
    for (k, &d) in scenes.iter().enumerate() {
        unsafe { rt_scene_destroy(ctxs[k], d) };
    }
    out
}

fn check(ctx: *mut RtContext, rc: i32, what: &str) -> Result<(), String> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = if ctx.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned()
    };
    Err(format!("{what} failed ({rc}): {msg}"))
}

/// The body of `Camera::render` (camera.rs:66-84): one ray per pixel through `World::color_at`, written to a new canvas.
pub fn render(cam: &Camera, world: &World) -> Result<Canvas, String> {
    if unsafe { rt_abi_version() } != RT_ABI_VERSION {
        return Err("librt_hip.so has another ABI version than raytracer-hip-sys".into());
    }
    let flat = Flat::of(world);
    let scene = flat.view();
    let rc = rt_camera(cam)?;
    // Area lights draw their jitter from thread_rng in the reference (lights.rs:114-119): any stream of uniform f32 is
    // as good as another, so the device's counter-based generator is seeded from the same source.
    let opts = RtRenderOptions {
        jitter_mode: RT_JITTER_COUNTER,
        jitter_value: 0.5,
        seed: rand::random(),
        tile_h: 8,
        n_parts: 1,
        part: 0,
        flags: 0,
    };
    let mut canvas = Canvas::new(cam.hsize, cam.vsize);
    let mut guard = GPU.lock();
    let ctxs = contexts(&mut guard)?;
    // Rgb32FImage is row-major f32 RGB (canvas.rs:10-31): exactly what rt_render_multi writes, unclamped; with one
    // GPU it is rt_render byte for byte, with several every GPU copies its stripes of rows into this buffer itself
    let out: &mut [f32] = canvas.buffer_mut();
    debug_assert_eq!(out.len(), cam.hsize as usize * cam.vsize as usize * 3);
    with_scenes(&ctxs, &scene, |scenes| {
        check(
            ctxs[0],
            unsafe {
                rt_render_multi(ctxs.as_ptr(), scenes.as_ptr(), ctxs.len() as u32, &rc, &opts, out.as_mut_ptr(), ptr::null_mut())
            },
            "rt_render_multi",
        )
    })?;
    Ok(canvas)
}

/// `Camera::render` followed by `Canvas::apply_dithering` and the 8-bit quantisation of `Canvas::to_ppm`
/// (scene.rs:215-224, canvas.rs:37-55, :73-75) in one call: RGB u8, row-major.  `dither` is one of RT_DITHER_*.
pub fn render_u8(cam: &Camera, world: &World, dither: u32) -> Result<Vec<u8>, String> {
    let flat = Flat::of(world);
    let scene = flat.view();
    let rc = rt_camera(cam)?;
    let opts = RtRenderOptions {
        jitter_mode: RT_JITTER_COUNTER,
        jitter_value: 0.5,
        seed: rand::random(),
        tile_h: 8,
        n_parts: 1,
        part: 0,
        flags: 0,
    };
    let mut out = vec![0u8; cam.hsize as usize * cam.vsize as usize * 3];
    let mut guard = GPU.lock();
    // (the 8-bit frame is a quarter of the bytes: one GPU's link carries it; rt_render_u8 has no multi-GPU form)
    let ctxs = contexts(&mut guard)?;
    let ctx = ctxs[0];
    with_scenes(&ctxs[..1], &scene, |scenes| {
        check(
            ctx,
            unsafe { rt_render_u8(ctx, scenes[0], &rc, &opts, dither, out.as_mut_ptr(), ptr::null_mut(), ptr::null_mut()) },
            "rt_render_u8",
        )
    })?;
    Ok(out)
}
