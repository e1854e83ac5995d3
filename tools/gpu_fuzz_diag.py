"""Diagnose a failing seed of gpu_fuzz.py (mixed worlds): which rays differ, between which shapes, under which flags."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
argv = sys.argv
import gpu_fuzz as F  # noqa: E402
from oracle import oracle  # noqa: E402
from raytracer_amd import abi, host  # noqa: E402

P = C.c_void_p
for seed in [int(a) for a in argv[1:]]:
    rng = np.random.RandomState(1000 + seed)
    rng.randint(4, 28), rng.randint(0, 24)
    gw, ow = host.new_world(0), oracle.new_world()
    F.mixed_world(gw, np.random.RandomState(seed))
    F.mixed_world(ow, np.random.RandomState(seed))
    span, n = 12.0, 40000
    o = rng.uniform(-span, span, size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(-span / 2, span / 2, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10
    d[:k] *= rng.uniform(1e-3, 1e3, size=(k, 1)).astype(np.float32)
    d[k:2 * k, int(rng.randint(3))] = 0.0
    d[2 * k:3 * k, int(rng.randint(3))] = 3e-5
    _, gt, gid, _ = gw.color_at(o, d, 0)
    _, rt, rid, _ = ow.color_at(o, d, 0)
    arr = abi.scene_arrays(gw.flatten())
    shapes = arr["shapes"]
    bad = np.nonzero((gid != rid) | ((rid >= 0) & (gt.view(np.uint32) != rt.view(np.uint32))))[0]
    ctx, ds = gw.device_scene()
    lib = host.hip_lib()
    print(f"seed {seed}: {len(bad)} rays differ; shapes {len(shapes)} nodes {len(arr['nodes'])} csg nodes {len(arr.get('csg_nodes', []))}")
    for i in bad:
        def desc(s):
            if s < 0:
                return "miss"
            r = shapes[s]
            return f"{s}:kind{int(r['kind_flags']) & 15}/parent{int(r['parent'])}/flags{int(r['kind_flags']) >> 4:x}"
        res = []
        for flags in (0, abi.RT_RENDER_NO_PRUNE, abi.RT_RENDER_NO_ACCEL, abi.RT_RENDER_NO_PRUNE | abi.RT_RENDER_NO_ACCEL):
            opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, flags)
            rgb = np.zeros(3, np.float32); t = np.zeros(1, np.float32); ids = np.zeros(1, np.int32); uv = np.zeros(2, np.float32)
            oo, dd = np.ascontiguousarray(o[i]), np.ascontiguousarray(d[i])
            rc = lib.rt_color_at(ctx, ds, C.byref(opts), oo.ctypes.data_as(P), dd.ctypes.data_as(P), 1, 0, rgb.ctypes.data_as(P),
                                 t.ctypes.data_as(P), ids.ctypes.data_as(P), uv.ctypes.data_as(P), None)
            res.append(f"flags{flags}: id {ids[0]} t {t[0]!r}")
        print(f" ray {i}: o {o[i]} d {d[i]}\n   gpu {desc(gid[i])} t {gt[i]!r}   oracle {desc(rid[i])} t {rt[i]!r}\n   " + " | ".join(res), flush=True)
