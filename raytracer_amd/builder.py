"""ctypes binding of the C world-builder API (include/rt_builder.h).

The same API is exported by two libraries under different prefixes: ``rth_`` (the product
host library, GPU render path) and ``orc_`` (the CPU oracle, test infrastructure).  This
module is prefix-agnostic plumbing; it computes nothing.

The class mirrors the reference's constructors (file:line in include/rt_builder.h):
``World`` (src/world.rs:12-18), shapes (src/shapes.rs:38-158), ``Group``
(src/groups.rs:44-98), ``Csg`` (src/csg.rs:14-24), lights (src/lights.rs:42-103),
``Material`` (src/materials.rs:35-79), patterns (src/patterns.rs) and
``Camera::render`` (src/camera.rs:66-84).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

# --- enums (include/rt_builder.h) -------------------------------------------------------
XF_TRANSLATE, XF_SCALE, XF_ROTATE_X, XF_ROTATE_Y, XF_ROTATE_Z, XF_SHEAR, XF_MATRIX = range(7)
SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE = range(7)
PAT_STRIPED, PAT_GRADIENT, PAT_RING, PAT_CHECKER, PAT_XYZ_RGB, PAT_TEXTURE_MAP, PAT_CUBE_MAP = range(7)
UVMAP_SPHERICAL, UVMAP_PLANAR, UVMAP_CYLINDRICAL = range(3)
CSG_UNION, CSG_INTERSECT, CSG_DIFFERENCE = range(3)
CHILD_SHAPE, CHILD_CSG = range(2)
JITTER_CONST, JITTER_COUNTER, JITTER_SEQUENCE = range(3)

SHAPE_KINDS = {
    "sphere": SPHERE, "plane": PLANE, "cube": CUBE, "cylinder": CYLINDER, "cone": CONE,
    "triangle": TRIANGLE, "smooth-triangle": SMOOTH_TRIANGLE,
}
XF_KINDS = {
    "translate": XF_TRANSLATE, "scale": XF_SCALE, "rotate-x": XF_ROTATE_X, "rotate-y": XF_ROTATE_Y,
    "rotate-z": XF_ROTATE_Z, "shear": XF_SHEAR,
}


class XfOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("a", C.c_float * 6), ("m", C.c_float * 16)]


class MaterialC(C.Structure):
    _fields_ = [
        ("color", C.c_float * 3), ("ambient", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float),
        ("shininess", C.c_float), ("reflective", C.c_float), ("transparency", C.c_float),
        ("refractive_index", C.c_float), ("pattern", C.c_int32),
    ]


class CameraC(C.Structure):
    _fields_ = [
        ("hsize", C.c_uint32), ("vsize", C.c_uint32), ("field_of_view", C.c_float), ("use_view", C.c_int32),
        ("from_", C.c_float * 3), ("to", C.c_float * 3), ("up", C.c_float * 3),
        ("ops", C.POINTER(XfOp)), ("nops", C.c_int32),
    ]


class RenderOptionsC(C.Structure):
    _fields_ = [
        ("jitter_mode", C.c_int32), ("jitter_value", C.c_float), ("seed", C.c_uint64), ("threads", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32), ("row_step", C.c_int32), ("col_step", C.c_int32),
    ]


class AuxC(C.Structure):
    _fields_ = [("hit_mask", C.POINTER(C.c_uint8)), ("object_id", C.POINTER(C.c_int32)),
                ("hit_t", C.POINTER(C.c_float))]


class StatsC(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("reflect", C.c_uint64),
                ("refract", C.c_uint64), ("overflow", C.c_uint64), ("seconds", C.c_double)]


# --- plain-Python descriptions ------------------------------------------------------------
@dataclass
class Material:
    """Material::default() (src/materials.rs:19-33)."""
    color: Sequence[float] = (1.0, 1.0, 1.0)
    ambient: float = 0.1
    diffuse: float = 0.9
    specular: float = 0.9
    shininess: float = 200.0
    reflective: float = 0.0
    transparency: float = 0.0
    refractive_index: float = 1.0
    pattern: int = -1

    def to_c(self) -> MaterialC:
        m = MaterialC()
        m.color[:] = [float(x) for x in self.color]
        m.ambient, m.diffuse, m.specular = self.ambient, self.diffuse, self.specular
        m.shininess, m.reflective, m.transparency = self.shininess, self.reflective, self.transparency
        m.refractive_index, m.pattern = self.refractive_index, int(self.pattern)
        return m


@dataclass
class Camera:
    """Camera::new(hsize, vsize, fov).with_transform(...) (src/camera.rs:19-43)."""
    hsize: int
    vsize: int
    field_of_view: float
    from_: Optional[Sequence[float]] = None
    to: Optional[Sequence[float]] = None
    up: Optional[Sequence[float]] = None
    ops: Sequence = field(default_factory=list)  # used when from_/to/up are None


@dataclass
class RenderOptions:
    jitter_mode: int = JITTER_CONST
    jitter_value: float = 0.5
    seed: int = 0
    threads: int = 0
    row_begin: int = 0
    row_end: int = 0
    row_step: int = 1
    col_step: int = 1

    def to_c(self) -> RenderOptionsC:
        return RenderOptionsC(self.jitter_mode, self.jitter_value, self.seed, self.threads, self.row_begin,
                              self.row_end, self.row_step, self.col_step)


@dataclass
class RenderResult:
    rgb: np.ndarray            # (vsize, hsize, 3) float32, unclamped
    hit_mask: Optional[np.ndarray]
    object_id: Optional[np.ndarray]
    hit_t: Optional[np.ndarray]
    stats: dict


def make_ops(ops: Sequence) -> "C.Array[XfOp]":
    """ops: sequence of (kind, *args); kind is an XF_* int or the scene-file name."""
    arr = (XfOp * max(len(ops), 1))()
    for i, op in enumerate(ops):
        kind = op[0]
        if isinstance(kind, str):
            kind = XF_KINDS[kind]
        arr[i].op = kind
        args = [float(x) for x in op[1:]]
        if kind == XF_MATRIX:
            arr[i].m[:] = args
        else:
            for j, a in enumerate(args):
                arr[i].a[j] = a
    return arr


def _f3(v) -> "C.Array[C.c_float]":
    return (C.c_float * 3)(*[float(x) for x in v])


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class WorldBuilder:
    """One world behind the C builder API of ``lib`` with symbol prefix ``prefix``."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib, self.prefix = lib, prefix
        self._declare()
        self.w = self._fn("world_new")()
        if not self.w:
            raise RuntimeError("world_new failed")

    # -- plumbing ---------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _declare(self):
        P, I, F = C.c_void_p, C.c_int, C.c_float
        FP, IP, XP = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(XfOp)
        MP = C.POINTER(MaterialC)
        sig = {
            "world_new": (P, []),
            "world_free": (None, [P]),
            "last_error": (C.c_char_p, [P]),
            "uv_checker": (I, [P, F, F, FP, FP]),
            "uv_align_check": (I, [P, FP]),
            "uv_image": (I, [P, C.c_uint32, C.c_uint32, FP]),
            "pattern_new": (I, [P, I, FP, XP, I, I, IP]),
            "shape_new": (I, [P, I, XP, I, MP, I, FP, I]),
            "group_new": (I, [P, XP, I]),
            "group_set_transform": (I, [P, I, XP, I]),
            "group_add_shape": (I, [P, I, I]),
            "group_add_child": (I, [P, I, I]),
            "group_add_mesh": (I, [P, I, MP, FP, I, FP, I, IP, IP, I]),
            "group_divide": (I, [P, I, I]),
            "csg_new": (I, [P, I, I, I, I, I]),
            "world_add_shape": (I, [P, I]),
            "world_add_group": (I, [P, I, I]),
            "world_add_csg": (I, [P, I]),
            "world_add_point_light": (I, [P, FP, FP]),
            "world_add_area_light": (I, [P, FP, FP, I, FP, I, FP]),
            "world_set_recursion_limit": (I, [P, I]),
            "render": (I, [P, C.POINTER(CameraC), C.POINTER(RenderOptionsC), FP, C.POINTER(AuxC),
                           C.POINTER(StatsC)]),
            "color_at": (I, [P, C.POINTER(RenderOptionsC), FP, FP, I, I, FP, FP, IP, FP]),
            "ray_for_pixel": (I, [P, C.POINTER(CameraC), C.c_uint32, C.c_uint32, FP, FP]),
        }
        for name, (res, args) in sig.items():
            fn = self._fn(name)
            fn.restype, fn.argtypes = res, args

    def close(self):
        if self.w:
            self._fn("world_free")(self.w)
            self.w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> int:
        if rc < 0:
            msg = self._fn("last_error")(self.w)
            raise RuntimeError(f"{self.prefix}{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    # -- patterns ---------------------------------------------------------------------
    def uv_checker(self, width, height, a, b) -> int:
        return self._check(self._fn("uv_checker")(self.w, width, height, _f3(a), _f3(b)), "uv_checker")

    def uv_align_check(self, main, ul, ur, bl, br) -> int:
        flat = (C.c_float * 15)(*[float(x) for c in (main, ul, ur, bl, br) for x in c])
        return self._check(self._fn("uv_align_check")(self.w, flat), "uv_align_check")

    def uv_image(self, rgb: np.ndarray) -> int:
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, _ = rgb.shape
        return self._check(self._fn("uv_image")(self.w, w, h, _fp(rgb)), "uv_image")

    def pattern(self, kind: int, colors=None, ops=(), uv_mapping: int = 0, uvpats=()) -> int:
        cols = (C.c_float * 6)(*([float(x) for c in colors for x in c] if colors else [0.0] * 6))
        uv = (C.c_int32 * 6)(*(list(uvpats) + [-1] * (6 - len(uvpats))))
        arr = make_ops(ops)
        return self._check(self._fn("pattern_new")(self.w, kind, cols, arr, len(ops), uv_mapping, uv), "pattern_new")

    # -- shapes -----------------------------------------------------------------------
    def shape(self, kind, ops=(), material: Optional[Material] = None, shadow: bool = True, params=None) -> int:
        if isinstance(kind, str):
            kind = SHAPE_KINDS[kind]
        arr = make_ops(ops)
        m = material.to_c() if material is not None else None
        if params is not None:
            p = (C.c_float * len(params))(*[float(x) for x in params])
            n = len(params)
        else:
            p, n = None, 0
        return self._check(
            self._fn("shape_new")(self.w, kind, arr, len(ops), C.byref(m) if m is not None else None,
                                  1 if shadow else 0, p, n), "shape_new")

    # -- groups -----------------------------------------------------------------------
    def group(self, ops=()) -> int:
        return self._check(self._fn("group_new")(self.w, make_ops(ops), len(ops)), "group_new")

    def group_set_transform(self, g: int, ops=()):
        self._check(self._fn("group_set_transform")(self.w, g, make_ops(ops), len(ops)), "group_set_transform")

    def group_add_shape(self, g: int, shape: int):
        self._check(self._fn("group_add_shape")(self.w, g, shape), "group_add_shape")

    def group_add_child(self, g: int, child: int):
        self._check(self._fn("group_add_child")(self.w, g, child), "group_add_child")

    def group_add_mesh(self, g: int, material: Optional[Material], positions, normals, vidx, nidx) -> int:
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        vidx = np.ascontiguousarray(vidx, dtype=np.int32).reshape(-1, 3)
        if normals is not None and len(normals) and nidx is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
            nidx = np.ascontiguousarray(nidx, dtype=np.int32).reshape(-1, 3)
            np_, ni_, nn = _fp(normals), _ip(nidx), len(normals)
        else:
            np_, ni_, nn = None, None, 0
        m = material.to_c() if material is not None else None
        return self._check(
            self._fn("group_add_mesh")(self.w, g, C.byref(m) if m is not None else None, _fp(positions),
                                       len(positions), np_, nn, _ip(vidx), ni_, len(vidx)), "group_add_mesh")

    def group_divide(self, g: int, threshold: int):
        self._check(self._fn("group_divide")(self.w, g, threshold), "group_divide")

    # -- csg --------------------------------------------------------------------------
    def csg(self, op: int, left_kind: int, left: int, right_kind: int, right: int) -> int:
        return self._check(self._fn("csg_new")(self.w, op, left_kind, left, right_kind, right), "csg_new")

    # -- world ------------------------------------------------------------------------
    def add_shape(self, shape: int) -> int:
        return self._check(self._fn("world_add_shape")(self.w, shape), "world_add_shape")

    def add_group(self, g: int, cache_bounds: bool = True):
        self._check(self._fn("world_add_group")(self.w, g, 1 if cache_bounds else 0), "world_add_group")

    def add_csg(self, c: int):
        self._check(self._fn("world_add_csg")(self.w, c), "world_add_csg")

    def add_point_light(self, position, intensity) -> int:
        return self._check(self._fn("world_add_point_light")(self.w, _f3(position), _f3(intensity)),
                           "world_add_point_light")

    def add_area_light(self, corner, full_uvec, usteps, full_vvec, vsteps, intensity) -> int:
        return self._check(
            self._fn("world_add_area_light")(self.w, _f3(corner), _f3(full_uvec), usteps, _f3(full_vvec), vsteps,
                                             _f3(intensity)), "world_add_area_light")

    def set_recursion_limit(self, limit: int):
        self._check(self._fn("world_set_recursion_limit")(self.w, limit), "world_set_recursion_limit")

    # -- camera / render --------------------------------------------------------------
    @staticmethod
    def _camera_c(cam: Camera):
        c = CameraC()
        c.hsize, c.vsize, c.field_of_view = cam.hsize, cam.vsize, cam.field_of_view
        keep = None
        if cam.from_ is not None:
            c.use_view = 1
            c.from_[:] = [float(x) for x in cam.from_]
            c.to[:] = [float(x) for x in cam.to]
            c.up[:] = [float(x) for x in cam.up]
            c.nops = 0
        else:
            c.use_view = 0
            keep = make_ops(cam.ops)
            c.ops = C.cast(keep, C.POINTER(XfOp))
            c.nops = len(cam.ops)
        return c, keep

    def render(self, cam: Camera, opts: Optional[RenderOptions] = None, aux: bool = False) -> RenderResult:
        """Camera::render(&World) -> Canvas (src/camera.rs:66-84)."""
        c, _keep = self._camera_c(cam)
        o = (opts or RenderOptions()).to_c()
        n = cam.hsize * cam.vsize
        rgb = np.zeros((cam.vsize, cam.hsize, 3), dtype=np.float32)
        st = StatsC()
        auxc, mask, oid, ht = None, None, None, None
        if aux:
            mask = np.zeros(n, dtype=np.uint8)
            oid = np.full(n, -1, dtype=np.int32)
            ht = np.zeros(n, dtype=np.float32)
            auxc = AuxC(mask.ctypes.data_as(C.POINTER(C.c_uint8)), _ip(oid), _fp(ht))
        rc = self._fn("render")(self.w, C.byref(c), C.byref(o), _fp(rgb), C.byref(auxc) if aux else None,
                                C.byref(st))
        self._check(rc, "render")
        stats = {k: getattr(st, k) for k, _ in StatsC._fields_}
        shape2 = (cam.vsize, cam.hsize)
        return RenderResult(rgb, mask.reshape(shape2) if aux else None, oid.reshape(shape2) if aux else None,
                            ht.reshape(shape2) if aux else None, stats)

    def color_at(self, origins, directions, remaining: int, opts: Optional[RenderOptions] = None):
        """World::color_at + Intersections::hit for a batch of rays."""
        origins = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        directions = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(origins)
        rgb = np.zeros((n, 3), dtype=np.float32)
        t = np.zeros(n, dtype=np.float32)
        ids = np.zeros(n, dtype=np.int32)
        uv = np.zeros((n, 2), dtype=np.float32)
        o = (opts or RenderOptions()).to_c()
        rc = self._fn("color_at")(self.w, C.byref(o), _fp(origins), _fp(directions), n, remaining, _fp(rgb), _fp(t),
                                  _ip(ids), _fp(uv))
        self._check(rc, "color_at")
        return rgb, t, ids, uv

    def ray_for_pixel(self, cam: Camera, px: int, py: int):
        c, _keep = self._camera_c(cam)
        out = (C.c_float * 6)()
        ps = C.c_float()
        self._check(self._fn("ray_for_pixel")(self.w, C.byref(c), px, py, out, C.byref(ps)), "ray_for_pixel")
        return np.array(out[:3], dtype=np.float32), np.array(out[3:], dtype=np.float32), ps.value
