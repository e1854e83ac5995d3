"""rt_color_at_static (include/rt_abi.h): the allocation-free variant of rt_color_at, through the C ABI.

Its rays are staged through a fixed device array that every call on the GPU shares and that keeps the last call's
contents, so the test makes calls in a row — the second with a smaller batch — and holds each against rt_color_at on the
same rays, bit for bit (colours, hit t, ids, (u, v), ray counters): a stale tail or a counter that was not cleared would
show.  Collected last (test_zz_*): it must pass alone in a fresh process and after every other GPU test."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path, small_dragons
from raytracer_amd import abi, host
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


def _rays(rng, n):
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.2
    tgt = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.ascontiguousarray(o), np.ascontiguousarray(d.astype(np.float32))


def _call(fn, ctx, ds, o, d, remaining, flags=0):
    n = len(o)
    rgb = np.full((n, 3), np.nan, dtype=np.float32)
    t = np.full(n, np.nan, dtype=np.float32)
    ids = np.full(n, -77, dtype=np.int32)
    uv = np.full((n, 2), np.nan, dtype=np.float32)
    st = abi.RtStats()
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 9, 1, 1, 0, flags)
    rc = fn(ctx, ds, C.byref(opts), o.ctypes.data, d.ctypes.data, n, remaining, rgb.ctypes.data, t.ctypes.data,
            ids.ctypes.data, uv.ctypes.data, C.byref(st))
    return rc, rgb, t, ids, uv, st


def _same(a, b):
    for x, y in zip(a[1:5], b[1:5]):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    for k in ("primary", "shadow", "reflect", "refract", "overflow"):
        assert getattr(a[5], k) == getattr(b[5], k), k


@pytest.mark.parametrize("which", ["dragons", "soft_shadows"])
def test_static_entry_point_equals_rt_color_at(which):
    """dragons: groups and auxiliary trees (k_color_at_jobs, the work queue in the static counter block);
    soft_shadows: group-less with an area light (k_color_at_direct)."""
    lib = host.hip_lib()
    scene = small_dragons(32, 24) if which == "dragons" else Scene.from_file(scene_path("soft_shadows"))
    hw = scene.build(host.new_world(0))
    ctx, ds = hw.device_scene()
    rng = np.random.RandomState(21)
    for n in (50000, 700, abi.RT_STATIC_MAX_RAYS, 1):  # a large batch, then smaller ones over its remains, then the maximum
        o, d = _rays(rng, n)
        got = _call(lib.rt_color_at_static, ctx, ds, o, d, 4)
        want = _call(lib.rt_color_at, ctx, ds, o, d, 4)
        assert got[0] == abi.RT_OK and want[0] == abi.RT_OK, lib.rt_last_error(ctx)
        _same(got, want)
        assert got[5].primary == n and (n == 1 or (got[3] >= 0).sum() > 0)
    # the instrumented kernels through the same scratch: work counters equal too
    o, d = _rays(rng, 3000)
    got = _call(lib.rt_color_at_static, ctx, ds, o, d, 2, abi.RT_RENDER_COUNT_WORK)
    want = _call(lib.rt_color_at, ctx, ds, o, d, 2, abi.RT_RENDER_COUNT_WORK)
    _same(got, want)
    # (the group-less kernel, k_color_at_direct, has no instrumented instantiation: zeros on both sides there)
    assert got[5].rays_reference == want[5].rays_reference and got[5].shape_tests == want[5].shape_tests
    assert got[5].node_visits == want[5].node_visits and got[5].tri_tests == want[5].tri_tests
    if which == "dragons":
        assert got[5].rays_reference > 0 and got[5].node_visits > 0


def test_static_entry_point_rejects_what_does_not_fit():
    lib = host.hip_lib()
    hw = Scene.from_file(scene_path("basic_scene")).build(host.new_world(0))
    ctx, ds = hw.device_scene()
    o, d = _rays(np.random.RandomState(1), abi.RT_STATIC_MAX_RAYS + 1)
    rc = _call(lib.rt_color_at_static, ctx, ds, o, d, 4)[0]
    assert rc == abi.RT_E_LIMIT and b"RT_STATIC_MAX_RAYS" in lib.rt_last_error(ctx)
    assert _call(lib.rt_color_at_static, ctx, ds, o[:0], d[:0], 4)[0] == abi.RT_OK  # nothing to do
    assert _call(lib.rt_color_at_static, ctx, ds, o[:10], d[:10], 99)[0] == abi.RT_E_SCENE  # deeper than any ray stack
    # deeper than the ordinary kernels' ray stack: the slow-path kernels keep lists in device memory that this entry
    # point may not allocate
    rc = _call(lib.rt_color_at_static, ctx, ds, o[:10], d[:10], 10)[0]
    assert rc == abi.RT_E_LIMIT and b"may not" in lib.rt_last_error(ctx)
    got, want = _call(lib.rt_color_at_static, ctx, ds, o[:10], d[:10], 4), _call(lib.rt_color_at, ctx, ds, o[:10], d[:10], 4)
    assert got[0] == abi.RT_OK
    _same(got, want)
