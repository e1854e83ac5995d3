"""ctypes mirror of include/rt_abi.h (struct layouts + prototypes of librt_hip.so)."""
from __future__ import annotations

import ctypes as C

import numpy as np

RT_ABI_VERSION = 2
RT_OK, RT_E_ABI, RT_E_DEVICE, RT_E_OOM, RT_E_SCENE, RT_E_OVERFLOW, RT_E_LIMIT = 0, -1, -2, -3, -4, -5, -6
RT_STATIC_MAX_RAYS = 65536
RT_SHAPE_KIND_MASK, RT_SHAPE_SHADOW, RT_SHAPE_CLOSED = 0xF, 0x10, 0x20
RT_VCLASS_NEVER = 0xFFFFFFFF
RT_NODE_ALL_TRIS, RT_NODE_PRUNABLE = 1, 2
RT_CSG_LEFT_IS_SHAPE, RT_CSG_RIGHT_IS_SHAPE = 1, 2
RT_JITTER_CONST, RT_JITTER_COUNTER = 0, 1
RT_DITHER_NONE, RT_DITHER_BAYER2, RT_DITHER_BAYER4, RT_DITHER_BAYER8, RT_DITHER_BAYER16, RT_DITHER_BAYER_COLOR = range(6)
RT_RENDER_NO_PRUNE, RT_RENDER_COUNT_WORK, RT_RENDER_NO_ACCEL, RT_RENDER_WORK_MAP, RT_RENDER_SHORT_STACK = 1, 2, 4, 8, 16

# numpy dtypes of the device records (must match the C structs byte for byte)
XFORM = np.dtype([("m", "<f4", 12)])
SHAPE = np.dtype([("kind_flags", "<u4"), ("xform", "<u4"), ("material", "<u4"), ("parent", "<i4"),
                  ("vclass", "<u4"), ("prim", "<u4"), ("mn", "<f4"), ("mx", "<f4")])
TRI_GEOM = np.dtype([("p1", "<f4", 3), ("pad0", "<f4"), ("e1", "<f4", 3), ("pad1", "<f4"),
                     ("e2", "<f4", 3), ("pad2", "<f4")])
TRI_NORMALS = np.dtype([("n1", "<f4", 3), ("pad0", "<f4"), ("n2", "<f4", 3), ("pad1", "<f4"),
                        ("n3", "<f4", 3), ("pad2", "<f4")])
NODE = np.dtype([("bmin", "<f4", 3), ("xform", "<u4"), ("bmax", "<f4", 3), ("flags", "<u4"),
                 ("first_shape", "<u4"), ("n_shapes", "<u4"), ("skip", "<u4"), ("parent", "<i4"),
                 ("first_tri", "<u4"), ("next_xf", "<i4"), ("idrun", "<u4"), ("n_children", "<u4")])
CSG_NODE = np.dtype([("bmin", "<f4", 3), ("op", "<u4"), ("bmax", "<f4", 3), ("flags", "<u4"),
                     ("left", "<i4"), ("right", "<i4"), ("left_lo", "<u4"), ("left_hi", "<u4")])
CSG_ROOT = np.dtype([("first", "<u4"), ("count", "<u4")])
MATERIAL = np.dtype([("color", "<f4", 3), ("ambient", "<f4"), ("diffuse", "<f4"), ("specular", "<f4"),
                     ("shininess", "<f4"), ("reflective", "<f4"), ("transparency", "<f4"),
                     ("refractive_index", "<f4"), ("pattern", "<i4"), ("pad", "<u4")])
PATTERN = np.dtype([("kind", "<u4"), ("xform", "<u4"), ("uv_mapping", "<u4"), ("pad", "<u4"),
                    ("a", "<f4", 3), ("pad_a", "<f4"), ("b", "<f4", 3), ("pad_b", "<f4"),
                    ("uv", "<i4", 6), ("pad2", "<u4", 2)])
UVPATTERN = np.dtype([("kind", "<u4"), ("width", "<f4"), ("height", "<f4"), ("image", "<i4"),
                      ("c", "<f4", (5, 3)), ("pad", "<u4")])
IMAGE = np.dtype([("width", "<u4"), ("height", "<u4"), ("offset", "<u8")])
LIGHT = np.dtype([("kind", "<u4"), ("usteps", "<u4"), ("vsteps", "<u4"), ("pad", "<u4"),
                  ("intensity", "<f4", 3), ("pad_i", "<f4"), ("position", "<f4", 3), ("pad_p", "<f4"),
                  ("corner", "<f4", 3), ("pad_c", "<f4"), ("uvec", "<f4", 3), ("pad_u", "<f4"),
                  ("vvec", "<f4", 3), ("pad_v", "<f4")])

assert XFORM.itemsize == 48 and SHAPE.itemsize == 32 and TRI_GEOM.itemsize == 48 and NODE.itemsize == 64
assert CSG_NODE.itemsize == 48 and MATERIAL.itemsize == 48 and PATTERN.itemsize == 80 and UVPATTERN.itemsize == 80
assert IMAGE.itemsize == 16 and LIGHT.itemsize == 96


class RtScene(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("recursion_limit", C.c_uint32),
        ("n_xforms", C.c_uint32), ("n_shapes", C.c_uint32), ("n_tris", C.c_uint32), ("n_nodes", C.c_uint32),
        ("n_csg_nodes", C.c_uint32), ("n_csg_roots", C.c_uint32),
        ("n_materials", C.c_uint32), ("n_patterns", C.c_uint32), ("n_uvpatterns", C.c_uint32),
        ("n_images", C.c_uint32), ("n_lights", C.c_uint32), ("n_loose_shapes", C.c_uint32),
        ("n_texels", C.c_uint64),
        ("xforms", C.c_void_p), ("shapes", C.c_void_p), ("tri_geom", C.c_void_p), ("tri_normals", C.c_void_p),
        ("nodes", C.c_void_p), ("csg_nodes", C.c_void_p), ("csg_roots", C.c_void_p), ("materials", C.c_void_p),
        ("patterns", C.c_void_p), ("uvpatterns", C.c_void_p), ("images", C.c_void_p), ("texels", C.c_void_p),
        ("lights", C.c_void_p),
    ]


class RtCamera(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("pixel_size", C.c_float),
                ("half_width", C.c_float), ("half_height", C.c_float), ("inv_transform", C.c_float * 12)]


class RtRenderOptions(C.Structure):
    _fields_ = [("jitter_mode", C.c_uint32), ("jitter_value", C.c_float), ("seed", C.c_uint64),
                ("tile_h", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32), ("flags", C.c_uint32)]


class RtAux(C.Structure):
    _fields_ = [("hit_mask", C.c_void_p), ("object_id", C.c_void_p), ("hit_t", C.c_void_p)]


class RtStats(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("reflect", C.c_uint64), ("refract", C.c_uint64),
                ("overflow", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shape_tests", C.c_uint64), ("shaded_hits", C.c_uint64), ("node_xforms", C.c_uint64),
                ("aux_tests", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("rays_reference", C.c_uint64)]


class RtAccelView(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_tris", C.c_uint64), ("n_roots", C.c_uint64), ("nodes", C.c_void_p),
                ("tris", C.c_void_p), ("root_of", C.c_void_p), ("gate_next", C.c_void_p), ("owner", C.c_void_p)]


# 4-wide node (rt_accel.h Layout): eight {xyz, word}: b[2c] = min of child c, b[2c+1] = max of child c; words 0-3 the child
# references (index << 4 | leaf triangle count, 0xFFFFFFFF = no such child), words 4-7 the normal cone as float bits.
# An entry record uses b[0] = {centre, radius bits} and b[1] = {lmax, amax, top node as uint bits | first shape}.
AUX_NODE = np.dtype([("v", "<f4", 3), ("w", "<u4")])
AUX_NODE = np.dtype([("b", AUX_NODE, 8)])
AUX_TRI = np.dtype([("p1", "<f4", 3), ("shape", "<u4"), ("e1", "<f4", 3), ("owner", "<u4"), ("e2", "<f4", 3),
                    ("kind_flags", "<u4")])
assert AUX_NODE.itemsize == 128 and AUX_TRI.itemsize == 48

EXPORTS = ["rt_host_alloc", "rt_host_free", "rt_accel_build", "rt_accel_build_device", "rt_accel_free", "rt_abi_version", "rt_context_create", "rt_context_destroy", "rt_last_error", "rt_scene_create",
           "rt_scene_destroy", "rt_part_rows", "rt_render", "rt_render_u8", "rt_canvas_to_u8_device", "rt_render_device", "rt_check", "rt_color_at", "rt_color_at_static", "rt_device_count", "rt_render_multi",
           "rt_render_multi_device"]


def declare(lib: C.CDLL) -> None:
    P = C.c_void_p
    lib.rt_abi_version.restype, lib.rt_abi_version.argtypes = C.c_uint32, []
    lib.rt_context_create.restype, lib.rt_context_create.argtypes = C.c_int, [C.c_int, C.POINTER(P)]
    lib.rt_context_destroy.restype, lib.rt_context_destroy.argtypes = None, [P]
    lib.rt_last_error.restype, lib.rt_last_error.argtypes = C.c_char_p, [P]
    lib.rt_scene_create.restype, lib.rt_scene_create.argtypes = C.c_int, [P, C.POINTER(RtScene), C.POINTER(P)]
    lib.rt_scene_destroy.restype, lib.rt_scene_destroy.argtypes = None, [P, P]
    lib.rt_host_alloc.restype, lib.rt_host_alloc.argtypes = P, [P, C.c_uint64]
    lib.rt_host_free.restype, lib.rt_host_free.argtypes = None, [P, P]
    lib.rt_accel_build.restype, lib.rt_accel_build.argtypes = C.c_int, [C.POINTER(RtScene), C.POINTER(RtAccelView)]
    lib.rt_accel_build_device.restype, lib.rt_accel_build_device.argtypes = C.c_int, [C.c_void_p, C.POINTER(RtScene), C.POINTER(RtAccelView)]
    lib.rt_accel_free.restype, lib.rt_accel_free.argtypes = None, [C.POINTER(RtAccelView)]
    lib.rt_part_rows.restype, lib.rt_part_rows.argtypes = C.c_uint32, [C.POINTER(RtCamera), C.POINTER(RtRenderOptions)]
    lib.rt_render.restype = C.c_int
    lib.rt_render.argtypes = [P, P, C.POINTER(RtCamera), C.POINTER(RtRenderOptions), P, C.POINTER(RtAux),
                              C.POINTER(RtStats)]
    lib.rt_render_u8.restype = C.c_int
    lib.rt_render_u8.argtypes = [P, P, C.POINTER(RtCamera), C.POINTER(RtRenderOptions), C.c_uint32, P, C.POINTER(RtAux),
                                 C.POINTER(RtStats)]
    lib.rt_canvas_to_u8_device.restype = C.c_int
    lib.rt_canvas_to_u8_device.argtypes = [P, C.POINTER(RtCamera), C.POINTER(RtRenderOptions), C.c_uint32, P, P, P]
    lib.rt_render_device.restype = C.c_int
    lib.rt_render_device.argtypes = [P, P, C.POINTER(RtCamera), C.POINTER(RtRenderOptions), P, C.POINTER(RtAux), P,
                                     C.POINTER(RtStats)]
    lib.rt_check.restype, lib.rt_check.argtypes = C.c_int, [P, P]
    lib.rt_device_count.restype, lib.rt_device_count.argtypes = C.c_int, []
    lib.rt_render_multi.restype = C.c_int
    lib.rt_render_multi.argtypes = [C.POINTER(P), C.POINTER(P), C.c_uint32, C.POINTER(RtCamera), C.POINTER(RtRenderOptions), P,
                                    C.POINTER(RtStats)]
    lib.rt_render_multi_device.restype = C.c_int
    lib.rt_render_multi_device.argtypes = lib.rt_render_multi.argtypes
    lib.rt_color_at.restype = C.c_int
    lib.rt_color_at.argtypes = [P, P, C.POINTER(RtRenderOptions), P, P, C.c_uint32, C.c_uint32, P, P, P, P,
                                C.POINTER(RtStats)]
    lib.rt_color_at_static.restype = C.c_int
    lib.rt_color_at_static.argtypes = lib.rt_color_at.argtypes


def _view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def accel_arrays(lib: C.CDLL, s: RtScene, ctx=None) -> dict:
    """The auxiliary BVHs of a flattened scene as numpy arrays (copies); with a context: the big trees built on its
    GPU (rt_accel_build_device), as rt_scene_create builds them."""
    v = RtAccelView()
    rc = lib.rt_accel_build_device(ctx, C.byref(s), C.byref(v)) if ctx else lib.rt_accel_build(C.byref(s), C.byref(v))
    if rc != RT_OK:
        raise RuntimeError(f"rt_accel_build failed ({rc})")
    try:
        return {"nodes": _view(v.nodes, v.n_nodes, AUX_NODE), "tris": _view(v.tris, v.n_tris, AUX_TRI),
                "root_of": _view(v.root_of, v.n_roots, np.dtype("<u4")),
                "gate_next": _view(v.gate_next, v.n_roots, np.dtype("<u4"))}
    finally:
        lib.rt_accel_free(C.byref(v))


def scene_arrays(s: RtScene) -> dict:
    return {
        "recursion_limit": s.recursion_limit, "n_loose_shapes": s.n_loose_shapes,
        "xforms": _view(s.xforms, s.n_xforms, XFORM), "shapes": _view(s.shapes, s.n_shapes, SHAPE),
        "tri_geom": _view(s.tri_geom, s.n_tris, TRI_GEOM), "tri_normals": _view(s.tri_normals, s.n_tris, TRI_NORMALS),
        "nodes": _view(s.nodes, s.n_nodes, NODE), "csg_nodes": _view(s.csg_nodes, s.n_csg_nodes, CSG_NODE),
        "csg_roots": _view(s.csg_roots, s.n_csg_roots, CSG_ROOT),
        "materials": _view(s.materials, s.n_materials, MATERIAL), "patterns": _view(s.patterns, s.n_patterns, PATTERN),
        "uvpatterns": _view(s.uvpatterns, s.n_uvpatterns, UVPATTERN), "images": _view(s.images, s.n_images, IMAGE),
        "texels": _view(s.texels, s.n_texels, np.dtype("<f4")), "lights": _view(s.lights, s.n_lights, LIGHT),
    }
