"""Camera::render over several GPUs behind the C ABI (include/rt_abi.h rt_render_multi / rt_render_multi_device).
This box has one GPU: n > 1 is exercised with several contexts on that GPU (the partition, the strided copies into the
host canvas, the gather layout and the de-interleave kernel are the same code); RCCL itself is rehearsed with a
communicator of one rank (RT_MULTI_FORCE_GATHER).  Real multi-GPU runs stay unmeasured until a node has more than one."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path, small_dragons
from raytracer_amd import abi, host
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


class Multi:
    def __init__(self, scene, size, n):
        import torch
        self.torch = torch
        self.lib = host.hip_lib()
        self.worlds = [scene.build(host.new_world(0)) for _ in range(n)]  # n contexts on GPU 0, the same world in each
        pairs = [w.device_scene() for w in self.worlds]
        self.ctxs = (C.c_void_p * n)(*[p[0].value for p in pairs])
        self.scenes = (C.c_void_p * n)(*[p[1].value for p in pairs])
        self.n = n
        self.cam = scene.camera(*size)
        self.rcam = self.worlds[0].rt_camera(self.cam)

    def opts(self, tile_h=8, flags=0):
        return abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, tile_h, 77, 5, flags)  # n_parts / part are ignored

    def check(self, rc):
        assert rc == abi.RT_OK, self.lib.rt_last_error(self.ctxs[0]).decode()

    def single(self):
        out = np.zeros((self.cam.vsize, self.cam.hsize, 3), dtype=np.float32)
        o = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
        st = abi.RtStats()
        self.check(self.lib.rt_render(self.ctxs[0], self.scenes[0], C.byref(self.rcam), C.byref(o), out.ctypes.data, None, C.byref(st)))
        return out, st

    def host(self, tile_h=8, pinned=False):
        n_px = self.cam.vsize * self.cam.hsize * 3
        stats = (abi.RtStats * self.n)()
        o = self.opts(tile_h)
        if pinned:
            p = self.lib.rt_host_alloc(self.ctxs[0], n_px * 4)
            assert p
            self.check(self.lib.rt_render_multi(self.ctxs, self.scenes, self.n, C.byref(self.rcam), C.byref(o), p, stats))
            out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n_px,)).reshape(self.cam.vsize, self.cam.hsize, 3).copy()
            self.lib.rt_host_free(self.ctxs[0], p)
        else:
            out = np.full((self.cam.vsize, self.cam.hsize, 3), np.nan, dtype=np.float32)
            self.check(self.lib.rt_render_multi(self.ctxs, self.scenes, self.n, C.byref(self.rcam), C.byref(o), out.ctypes.data, stats))
        return out, stats

    def device(self, tile_h=8):
        torch = self.torch
        out = torch.full((self.cam.vsize, self.cam.hsize, 3), float("nan"), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        stats = (abi.RtStats * self.n)()
        o = self.opts(tile_h)
        self.check(self.lib.rt_render_multi_device(self.ctxs, self.scenes, self.n, C.byref(self.rcam), C.byref(o), out.data_ptr(), stats))
        return out.cpu().numpy(), stats


def _rays(st):
    return st.primary + st.shadow + st.reflect + st.refract


@pytest.mark.parametrize("n", [1, 2, 3, 8])
@pytest.mark.parametrize("name,size", [("cover", (640, 360)), ("dragons", (333, 187))])
def test_multi_equals_single(name, size, n):
    """n contexts share a frame: the host canvas (pageable and page-locked) and the gathered device canvas are the
    bytes of rt_render on one context, whatever the stripe height, ragged last stripes included; the parts' rays add up."""
    scene = small_dragons(32, 24) if name == "dragons" else Scene.from_file(scene_path(name))
    m = Multi(scene, size, n)
    want, st = m.single()
    for tile_h in (8, 5):
        got, stats = m.host(tile_h)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"host canvas, n={n} tile_h={tile_h}"
        assert sum(_rays(s) for s in stats) == _rays(st) and all(s.overflow == 0 for s in stats)
        got, stats = m.device(tile_h)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"device canvas, n={n} tile_h={tile_h}"
        assert sum(_rays(s) for s in stats) == _rays(st)
    got, _ = m.host(8, pinned=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"page-locked host canvas, n={n}"


def test_multi_at_4k_and_through_rccl(monkeypatch):
    """dragons at 3840x2160: four contexts against one (host and device canvases), and the exchange step itself with
    RCCL linked in at run time: a communicator of one rank sends the part to itself (RT_MULTI_FORCE_GATHER), then the
    de-interleave kernel."""
    from raytracer_amd import synth
    scene = Scene.from_file(scene_path("dragons"), meshes={"dragon.obj": synth.synth_dragon(220, 220)})
    m = Multi(scene, (3840, 2160), 4)
    want, st = m.single()
    got, stats = m.host()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert sum(_rays(s) for s in stats) == _rays(st)
    got, _ = m.device()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    one = Multi(scene, (3840, 2160), 1)
    monkeypatch.setenv("RT_MULTI_FORCE_GATHER", "1")
    got, _ = one.device()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_multi_through_the_world_interface():
    """The C++ twin of the Rust host (librt_host.so): a world bound to several devices renders through rt_render_multi."""
    scene = Scene.from_file(scene_path("cover"))
    cam = scene.camera(320, 320)
    one = scene.build(host.new_world(0)).render(cam)
    three = scene.build(host.new_world([0, 0, 0])).render(cam)
    assert np.array_equal(one.rgb.view(np.uint32), three.rgb.view(np.uint32))
    assert all(one.stats[k] == three.stats[k] for k in ("primary", "shadow", "reflect", "refract"))


def test_multi_argument_errors():
    scene = Scene.from_file(scene_path("basic_scene"))
    m = Multi(scene, (64, 36), 2)
    lib, out = m.lib, np.zeros((36, 64, 3), dtype=np.float32)
    assert lib.rt_device_count() >= 1
    o = m.opts()
    assert lib.rt_render_multi(m.ctxs, m.scenes, 0, C.byref(m.rcam), C.byref(o), out.ctypes.data, None) == abi.RT_E_ABI
    assert lib.rt_render_multi(m.ctxs, m.scenes, 17, C.byref(m.rcam), C.byref(o), out.ctypes.data, None) == abi.RT_E_ABI
    assert lib.rt_render_multi(m.ctxs, m.scenes, 2, C.byref(m.rcam), C.byref(o), None, None) == abi.RT_E_ABI
    twice = (C.c_void_p * 2)(m.ctxs[0], m.ctxs[0])
    assert lib.rt_render_multi(twice, m.scenes, 2, C.byref(m.rcam), C.byref(o), out.ctypes.data, None) == abi.RT_E_ABI
    swapped = (C.c_void_p * 2)(m.scenes[1], m.scenes[0])
    assert lib.rt_render_multi(m.ctxs, swapped, 2, C.byref(m.rcam), C.byref(o), out.ctypes.data, None) == abi.RT_E_ABI
    assert b"another context" in lib.rt_last_error(m.ctxs[0])
