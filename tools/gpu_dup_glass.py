"""Transparent solids listed more than once (n1/n2 of Computations::prepare walk the deduplicated list,
intersections.rs:141-170): frames and rays, device against oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracer_amd import host, builder as B
from raytracer_amd.canvas import to_u8
from oracle import oracle

from test_gpu_parity import _dup_glass_world as build

bad = 0
cam = B.Camera(96, 72, 0.8, (0.5, 1.2, -5.0), (0, 0, 0), (0, 1, 0))
for kind in ("sphere", "cube", "cylinder"):
    for outer in (1, 2, 3):
        for inner in (0, 1, 2):
            for grouped in (False, True):
                g = build(host.new_world(0), outer, inner, grouped, kind).render(cam, aux=True)
                r = build(oracle.new_world(), outer, inner, grouped, kind).render(cam, aux=True)
                d = np.abs(to_u8(g.rgb).astype(int) - to_u8(r.rgb).astype(int)).max(axis=2)
                ok = np.array_equal(g.object_id, r.object_id) and (d <= 1).mean() >= 0.999 and all(g.stats[k] == r.stats[k] for k in ("primary", "shadow", "reflect", "refract"))
                bad += not ok
                print(kind, outer, inner, grouped, "ok" if ok else f"FAIL ids {np.array_equal(g.object_id, r.object_id)} within 1 LSB {(d <= 1).mean():.4f} rays {[g.stats[k] for k in ('shadow','reflect','refract')]} vs {[r.stats[k] for k in ('shadow','reflect','refract')]}", flush=True)
print("failures:", bad)
