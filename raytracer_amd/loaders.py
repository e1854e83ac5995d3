"""ctypes binding of include/rt_loaders.h (librt_host.so): OBJ in, PPM in, PPM out, native and threaded.

The formats on either side of the render path: parse_obj (src/obj.rs:8-41),
Canvas::from_ppm (src/canvas.rs:86-107) and Canvas::to_ppm (src/canvas.rs:57-84) of the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import host
from .builder import Material, MaterialC

_declared = False


class RtlObj(C.Structure):
    _fields_ = [("n_positions", C.c_uint32), ("n_normals", C.c_uint32), ("n_triangles", C.c_uint32),
                ("n_groups", C.c_uint32), ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("vidx", C.POINTER(C.c_int32)), ("nidx", C.POINTER(C.c_int32)), ("group_first", C.POINTER(C.c_uint32)),
                ("group_names", C.POINTER(C.c_char_p)), ("owner", C.c_void_p)]


EXPORTS = ["rtl_last_error", "rtl_group_add_obj", "rtl_obj_parse", "rtl_obj_free", "rtl_ppm_decode", "rtl_uv_image_ppm",
           "rtl_ppm_encode", "rtl_ppm_write", "rtl_free"]


def lib() -> C.CDLL:
    global _declared
    L = host.host_lib()
    if not _declared:
        L.rtl_last_error.restype = C.c_char_p
        L.rtl_last_error.argtypes = []
        L.rtl_group_add_obj.restype = C.c_int
        L.rtl_group_add_obj.argtypes = [C.c_void_p, C.c_int, C.POINTER(MaterialC), C.c_char_p, C.c_size_t, C.c_int]
        L.rtl_obj_parse.restype = C.c_int
        L.rtl_obj_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(RtlObj)]
        L.rtl_obj_free.restype = None
        L.rtl_obj_free.argtypes = [C.POINTER(RtlObj)]
        L.rtl_ppm_decode.restype = C.c_int
        L.rtl_ppm_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.POINTER(C.c_float))]
        L.rtl_uv_image_ppm.restype = C.c_int
        L.rtl_uv_image_ppm.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.rtl_ppm_encode.restype = C.c_int
        L.rtl_ppm_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_size_t)]
        L.rtl_ppm_write.restype = C.c_int
        L.rtl_ppm_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.rtl_free.restype = None
        L.rtl_free.argtypes = [C.c_void_p]
        _declared = True
    return L


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise RuntimeError(f"{what}: {lib().rtl_last_error().decode()}")
    return rc


def parse_obj_bytes(data: bytes, threads: int = 0) -> List[dict]:
    """The groups of an OBJ file in the form raytracer_amd/obj.py returns them (same keys, same arrays)."""
    L = lib()
    o = RtlObj()
    _check(L.rtl_obj_parse(data, len(data), threads, C.byref(o)), "rtl_obj_parse")
    try:
        pos = np.ctypeslib.as_array(o.positions, shape=(o.n_positions, 3)).copy() if o.n_positions else np.zeros((0, 3), np.float32)
        nrm = np.ctypeslib.as_array(o.normals, shape=(o.n_normals, 3)).copy() if o.n_normals else None
        vidx = np.ctypeslib.as_array(o.vidx, shape=(o.n_triangles, 3)).copy() if o.n_triangles else np.zeros((0, 3), np.int32)
        nidx = np.ctypeslib.as_array(o.nidx, shape=(o.n_triangles, 3)).copy() if o.n_triangles else np.zeros((0, 3), np.int32)
        first = [int(o.group_first[g]) for g in range(o.n_groups + 1)]
        out = []
        for g in range(o.n_groups):
            out.append({"name": o.group_names[g].decode(), "positions": pos, "normals": nrm,
                        "vidx": vidx[first[g]:first[g + 1]], "nidx": nidx[first[g]:first[g + 1]]})
        return out
    finally:
        L.rtl_obj_free(C.byref(o))


def group_add_obj(world, group: int, material: Optional[Material], data: bytes, threads: int = 0) -> int:
    """parse_obj (src/obj.rs:8-41) into `group` of a HostWorld: one child group per OBJ group; returns their number."""
    m = material.to_c() if material is not None else None
    return _check(lib().rtl_group_add_obj(world.w, group, C.byref(m) if m is not None else None, data, len(data), threads),
                  "rtl_group_add_obj")


def ppm_decode(data: bytes) -> np.ndarray:
    """Canvas::from_ppm (src/canvas.rs:86-107): (h, w, 3) f32."""
    L = lib()
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    _check(L.rtl_ppm_decode(data, len(data), C.byref(w), C.byref(h), C.byref(p)), "rtl_ppm_decode")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        L.rtl_free(p)


def uv_image_ppm(world, data: bytes) -> int:
    return _check(lib().rtl_uv_image_ppm(world.w, data, len(data)), "rtl_uv_image_ppm")


def ppm_encode(rgb_u8: np.ndarray, threads: int = 0) -> bytes:
    """Canvas::to_ppm (src/canvas.rs:57-84) of an (h, w, 3) uint8 frame."""
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, _ = a.shape
    L = lib()
    text, n = C.c_void_p(), C.c_size_t()
    _check(L.rtl_ppm_encode(a.ctypes.data, w, h, threads, C.byref(text), C.byref(n)), "rtl_ppm_encode")
    try:
        return C.string_at(text.value, n.value)
    finally:
        L.rtl_free(text)


def ppm_write(path: str, rgb_u8: np.ndarray, threads: int = 0) -> None:
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, _ = a.shape
    _check(lib().rtl_ppm_write(path.encode(), a.ctypes.data, w, h, threads), "rtl_ppm_write")
