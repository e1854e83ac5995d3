"""Loader for the product libraries: librt_host.so (world model + flatten) and librt_hip.so
(C ABI + gfx950 kernels).  There is no CPU render fallback: if the libraries are missing or no
GPU is visible, rendering raises.

``new_world()`` returns a :class:`raytracer_amd.builder.WorldBuilder` bound to the ``rth_``
prefix, i.e. the C++ mirror of the reference's World/Camera interface whose ``render`` is the
MI355X path (src/camera.rs:66-84 replaced).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi
from .builder import CameraC, WorldBuilder

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HIP_LIB = os.path.join(CSRC, "librt_hip.so")
HOST_LIB = os.path.join(CSRC, "librt_host.so")

_hip = None
_host = None


def hip_lib() -> C.CDLL:
    """librt_hip.so — the drop-in boundary (include/rt_abi.h)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hip = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        abi.declare(_hip)
    return _hip


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = C.CDLL(HOST_LIB)
        _host.rth_flatten.restype = C.POINTER(abi.RtScene)
        _host.rth_flatten.argtypes = [C.c_void_p]
        _host.rth_set_device.restype = C.c_int
        _host.rth_set_device.argtypes = [C.c_void_p, C.c_int]
        _host.rth_set_devices.restype = C.c_int
        _host.rth_set_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        _host.rth_make_camera.restype = C.c_int
        _host.rth_make_camera.argtypes = [C.c_void_p, C.POINTER(CameraC), C.POINTER(abi.RtCamera)]
        _host.rth_device_scene.restype = C.c_int
        _host.rth_device_scene.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    return _host


class HostWorld(WorldBuilder):
    """World on the product path, plus access to the flattened scene and the C-ABI handles."""

    def __init__(self, device=0):
        """device: a HIP ordinal, or a list of them: frames rendered without aux outputs are then shared between those
        GPUs (rt_render_multi); an ordinal listed twice means two contexts on that GPU."""
        super().__init__(host_lib(), "rth_")
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*device)
            self._check(self.lib.rth_set_devices(self.w, arr, len(device)), "set_devices")
        else:
            self.lib.rth_set_device(self.w, device)

    def flatten(self) -> "abi.RtScene":
        p = self.lib.rth_flatten(self.w)
        if not p:
            raise RuntimeError("flatten failed")
        return p.contents

    def flat_arrays(self) -> dict:
        """The flattened scene as numpy structured arrays (copies), for host-logic tests."""
        return abi.scene_arrays(self.flatten())

    def rt_camera(self, cam) -> "abi.RtCamera":
        c, _keep = self._camera_c(cam)
        out = abi.RtCamera()
        self._check(self.lib.rth_make_camera(self.w, C.byref(c), C.byref(out)), "make_camera")
        return out

    def device_scene(self):
        """(RtContext*, RtDeviceScene*) with the scene resident in HBM."""
        ctx, sc = C.c_void_p(), C.c_void_p()
        self._check(self.lib.rth_device_scene(self.w, C.byref(ctx), C.byref(sc)), "device_scene")
        return ctx, sc


def new_world(device=0) -> HostWorld:
    return HostWorld(device)
