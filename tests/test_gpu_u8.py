"""The device's 8-bit output path through the C ABI: rt_render_u8 and rt_canvas_to_u8_device against the
oracle's restatement of Canvas::apply_dithering + the quantisation of to_ppm (tests/test_kat_canvas.py pins it to the
reference's known answers).  Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path, small_dragons
from raytracer_amd import abi, host, tiles
from raytracer_amd.scene import Scene
from test_kat_canvas import to_u8

pytestmark = pytest.mark.gpu


def _ctx(scene, size):
    hw = scene.build(host.new_world(0))
    ctx, ds = hw.device_scene()
    return hw, ctx, ds, hw.rt_camera(scene.camera(*size))


@pytest.mark.parametrize("name,size", [("basic_scene", (333, 187)), ("dragons", (640, 256)), ("cover", (3840, 2160))])
def test_render_u8_is_the_quantised_render(name, size):
    lib = host.hip_lib()
    scene = small_dragons(32, 24) if name == "dragons" else Scene.from_file(scene_path(name))
    hw, ctx, ds, rcam = _ctx(scene, size)
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
    w, h = size
    rgb = np.zeros((h, w, 3), dtype=np.float32)
    assert lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), rgb.ctypes.data, None, None) == abi.RT_OK
    for dither in range(6):
        out = np.full((h, w, 3), 77, dtype=np.uint8)
        st = abi.RtStats()
        rc = lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), dither, out.ctypes.data, None, C.byref(st))
        assert rc == abi.RT_OK, lib.rt_last_error(ctx)
        assert st.primary == w * h
        want = to_u8(rgb, dither)
        assert np.array_equal(out, want), f"{name} dither {dither}: {(out != want).sum()} bytes differ"
        if dither in (1, 2, 3, 4):
            assert set(np.unique(out)) <= {0, 255}
    assert lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), 6, out.ctypes.data, None, None) == abi.RT_E_ABI


def test_canvas_to_u8_device_special_values_and_parts():
    """NaN, infinities, negatives, values beyond 1 and exact half-way cases through the device kernel; a frame split into
    interleaved parts dithers like the whole frame (the Bayer matrix is indexed by the row in the image); unaligned
    buffers and pixel counts that are no multiple of four take the scalar path."""
    import torch
    lib = host.hip_lib()
    hw, ctx, ds, _ = _ctx(Scene.from_file(scene_path("basic_scene")), (16, 16))
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    w, h = 61, 47
    rgb = rng.uniform(-0.2, 1.2, size=(h, w, 3)).astype(np.float32)
    k = np.arange(256, dtype=np.float32)
    specials = np.concatenate([[np.nan, np.inf, -np.inf, -0.0, 1e-30, 2.0], (k + 0.5) / 255, k / 255]).astype(np.float32)
    rgb.reshape(-1)[: len(specials)] = specials
    cam = abi.RtCamera(w, h, 0.01, 1.0, 1.0)
    d_rgb = torch.from_numpy(rgb).to(dev)
    for dither in range(6):
        d_out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert lib.rt_canvas_to_u8_device(ctx, C.byref(cam), None, dither, d_rgb.data_ptr(), d_out.data_ptr(), None) == abi.RT_OK
        assert lib.rt_check(ctx, None) == abi.RT_OK
        want = to_u8(rgb, dither)
        assert np.array_equal(d_out.cpu().numpy(), want), f"dither {dither}"
        # three interleaved parts of 4-row stripes
        for p in range(3):
            rows = tiles.part_rows(h, 4, 3, p)
            po = abi.RtRenderOptions(0, 0.5, 0, 4, 3, p, 0)
            part_in = torch.from_numpy(rgb[rows]).to(dev)
            part_out = torch.zeros((len(rows), w, 3), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            assert lib.rt_canvas_to_u8_device(ctx, C.byref(cam), C.byref(po), dither, part_in.data_ptr(), part_out.data_ptr(), None) == abi.RT_OK
            assert lib.rt_check(ctx, None) == abi.RT_OK
            assert np.array_equal(part_out.cpu().numpy(), want[rows]), f"dither {dither} part {p}"
    # an input pointer that is not 16-byte aligned, an output pointer that is not 4-byte aligned
    big_in = torch.zeros(h * w * 3 + 8, dtype=torch.float32, device=dev)
    big_in[1: 1 + h * w * 3] = torch.from_numpy(rgb.reshape(-1)).to(dev)
    big_out = torch.zeros(h * w * 3 + 8, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert lib.rt_canvas_to_u8_device(ctx, C.byref(cam), None, 5, big_in.data_ptr() + 4, big_out.data_ptr() + 1, None) == abi.RT_OK
    assert lib.rt_check(ctx, None) == abi.RT_OK
    assert np.array_equal(big_out.cpu().numpy()[1: 1 + h * w * 3].reshape(h, w, 3), to_u8(rgb, 5))


def test_page_locked_host_buffers_receive_the_same_frames():
    """rt_host_alloc / rt_host_free (include/rt_abi.h): rt_render and rt_render_u8 into page-locked memory give the bytes
    they give into pageable memory."""
    import ctypes as C
    from raytracer_amd import abi, host
    from raytracer_amd.scene import Scene
    from conftest import scene_path
    lib = host.hip_lib()
    scene = Scene.from_file(scene_path("basic_scene"))
    hw = scene.build(host.new_world(0))
    ctx, ds = hw.device_scene()
    cam = scene.camera(320, 200)
    rcam = hw.rt_camera(cam)
    opts = abi.RtRenderOptions(abi.RT_JITTER_CONST, 0.5, 0, 1, 1, 0, 0)
    n = 320 * 200 * 3
    pf, pu = lib.rt_host_alloc(ctx, n * 4), lib.rt_host_alloc(ctx, n)
    assert pf and pu
    assert lib.rt_host_alloc(ctx, 0) is None
    try:
        f = np.zeros(n, dtype=np.float32)
        u = np.zeros(n, dtype=np.uint8)
        assert lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), f.ctypes.data, None, None) == abi.RT_OK
        assert lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), pf, None, None) == abi.RT_OK
        assert lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), abi.RT_DITHER_BAYER4, u.ctypes.data, None, None) == abi.RT_OK
        assert lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), abi.RT_DITHER_BAYER4, pu, None, None) == abi.RT_OK
        assert np.array_equal(np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_float)), shape=(n,)).view(np.uint32), f.view(np.uint32))
        assert np.array_equal(np.ctypeslib.as_array(C.cast(pu, C.POINTER(C.c_uint8)), shape=(n,)), u)
        assert f.max() > 0.1 and u.max() > 0
    finally:
        lib.rt_host_free(ctx, pf)
        lib.rt_host_free(ctx, pu)
