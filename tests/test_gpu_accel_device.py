"""The auxiliary trees built on the device (raytracer_amd/csrc/rt_accel_device.h) against the host
builder's (rt_accel.h TreeBuilder): the same level-by-level binned SAH on the same arithmetic must give the same tree
-- every box, every child reference, every leaf triangle in the same place; only the normal cones may differ, in the
last bits (the host sums the unit normals in double in triangle order, the device in 2^-32 fixed point in any order)."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _dragons(nu, nv):
    return Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon(nu, nv)})


def _soup_world(seed, n):
    """Random triangles of very different sizes, some degenerate (no usable normal), in one group."""
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-4, 4, (n, 3))
    size = 10.0 ** rng.uniform(-3, 0, (n, 1))
    pos = (centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size[:, None, :]).astype(np.float32)
    pos[::97, 2] = pos[::97, 1]  # degenerate: two equal corners
    pos[5::41, :, 1] = 0.25      # many centroids with the same y: bins on that axis are lopsided
    w = host.new_world(0)
    g = w.group()
    vidx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    w.group_add_mesh(g, None, pos.reshape(-1, 3), None, vidx, None)
    w.add_group(g)
    w.add_point_light((-5, 8, -9), (1, 1, 1))
    return w


def _compare(lib, ctx, fs, min_tris, exact_order=True):
    h = abi.accel_arrays(lib, fs)
    d = abi.accel_arrays(lib, fs, ctx)
    assert b"built on the host" not in lib.rt_last_error(ctx), lib.rt_last_error(ctx).decode()
    assert len(d["tris"]) == len(h["tris"]) >= min_tris and len(d["nodes"]) == len(h["nodes"])
    assert np.array_equal(d["root_of"], h["root_of"]) and np.array_equal(d["gate_next"], h["gate_next"])
    hb, db = h["nodes"]["b"], d["nodes"]["b"]
    entry = np.zeros(len(hb), dtype=bool)
    entry[h["root_of"][h["root_of"] != 0xFFFFFFFF]] = True
    # leaf triangles: the same triangles in every leaf.  (The same order too, except below a node that was split at the
    # object median: std::nth_element leaves the order inside the halves unspecified, the device keeps the old one.)
    refs = hb["w"][~entry][:, :4].ravel()
    refs = np.sort(refs[(refs != 0xFFFFFFFF) & ((refs & 15) != 0)])
    assert (refs & 15).sum() == len(h["tris"]) and np.array_equal((refs >> 4)[1:], ((refs >> 4) + (refs & 15))[:-1])
    leaf = np.repeat(refs >> 4, refs & 15)

    def by_leaf(tris):
        words = tris.view("<u4").reshape(len(tris), 12)
        return words[np.lexsort(tuple(words[:, k] for k in range(11, -1, -1)) + (leaf,))]
    assert np.array_equal(by_leaf(d["tris"]), by_leaf(h["tris"]))
    if exact_order:
        assert d["tris"].tobytes() == h["tris"].tobytes()
    # boxes (and the entry records' centres, radii, lmax, amax): the same values
    assert np.array_equal(hb["v"], db["v"])
    # words 0-3: child references (entry records: radius bits / first shape)
    assert np.array_equal(hb["w"][:, :4], db["w"][:, :4])
    # words 4-7: the cone, {axis / kalpha, cos w}
    hc, dc = hb["w"][:, 4:].copy().view("<f4"), db["w"][:, 4:].copy().view("<f4")
    assert np.array_equal(hc[entry], dc[entry])
    assert np.array_equal(hc[~entry, 3] < -1.0, dc[~entry, 3] < -1.0)  # "no cone" on the same nodes
    np.testing.assert_allclose(dc[~entry], hc[~entry], rtol=0, atol=2e-6)
    return h


def test_device_tree_equals_host_tree_on_the_dragon():
    """dragons.yaml with a 96 800-triangle dragon: six instances, one shared tree, built on the device."""
    scene = _dragons(220, 220)
    hw = scene.build(host.new_world(0))
    fs = hw.flatten()
    ctx, _ = hw.device_scene()
    h = _compare(host.hip_lib(), ctx, fs, 96_000)
    assert (h["root_of"] != 0xFFFFFFFF).sum() == 6


def test_device_tree_equals_host_tree_on_the_bench_mesh():
    """The 871 200-triangle dragon of bench.py: deep in the tree some nodes find no split that leaves room for the
    levels below and take the object median (k_median on the device)."""
    scene = _dragons(660, 660)
    hw = scene.build(host.new_world(0))
    fs = hw.flatten()
    ctx, _ = hw.device_scene()
    _compare(host.hip_lib(), ctx, fs, 871_200, exact_order=False)


@pytest.mark.parametrize("seed,n", [(1, 20_000), (2, 70_001), (3, 2_000_000)])
def test_device_tree_equals_host_tree_on_triangle_soup(seed, n):
    """(2 M triangles: close to what one tree of 20 levels holds, so that already high in the tree only nearly even
    splits leave room for the levels below and large nodes take the object median: k_mcount / k_mupdate.)"""
    w = _soup_world(seed, n)
    fs = w.flatten()
    ctx, _ = w.device_scene()
    _compare(host.hip_lib(), ctx, fs, n, exact_order=n < 1_000_000)


def test_small_trees_stay_on_the_host():
    """Below 16384 triangles a tree is built by the host builder in either view."""
    scene = _dragons(32, 24)
    hw = scene.build(host.new_world(0))
    fs = hw.flatten()
    ctx, _ = hw.device_scene()
    lib = host.hip_lib()
    h, d = abi.accel_arrays(lib, fs), abi.accel_arrays(lib, fs, ctx)
    assert h["nodes"].tobytes() == d["nodes"].tobytes() and h["tris"].tobytes() == d["tris"].tobytes()


def test_frames_do_not_depend_on_where_the_trees_were_built(monkeypatch):
    """A 1920x1080 frame of dragons.yaml from a scene whose trees the device built and from one whose trees the host
    built (RT_HOST_ACCEL_BUILD): same bytes, same ids, same ray counts."""
    import torch
    scene = _dragons(220, 220)
    lib = host.hip_lib()

    def frame():
        hw = scene.build(host.new_world(0))
        ctx, ds = hw.device_scene()
        cam = hw.rt_camera(scene.camera(1920, 1080))
        out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
        oid = torch.full((1080, 1920), -7, dtype=torch.int32, device="cuda:0")
        aux = abi.RtAux(None, oid.data_ptr(), None)
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
        st = abi.RtStats()
        torch.cuda.synchronize()
        rc = lib.rt_render_device(ctx, ds, C.byref(cam), C.byref(opts), out.data_ptr(), C.byref(aux), None, C.byref(st))
        assert rc == abi.RT_OK, lib.rt_last_error(ctx).decode()
        return out.cpu().numpy(), oid.cpu().numpy(), (st.primary, st.shadow, st.reflect, st.refract, st.overflow)

    dev = frame()
    monkeypatch.setenv("RT_HOST_ACCEL_BUILD", "1")
    hst = frame()
    assert np.array_equal(dev[0].view(np.uint32), hst[0].view(np.uint32)) and np.array_equal(dev[1], hst[1])
    assert dev[2] == hst[2] and dev[2][4] == 0
