"""Ids of solids listed three times, loose and in a group, device against oracle (the probe that showed the rule for
repeated triangles to be a rule for every kind; tests/test_gpu_parity.py test_equal_solids_listed_more_than_once)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracer_amd import host
from oracle import oracle
def build(w, kind, grouped):
    params = [-1.0, 1.0, 1.0] if kind in ("cylinder", "cone") else None
    shapes = [w.shape(kind, [("translate", 0, 0, 0)], None, params=params) for _ in range(3)]
    if grouped:
        g = w.group([("translate", 0.5, 0, 0)])
        for s in shapes: w.group_add_shape(g, s)
        w.add_group(g)
    else:
        for s in shapes: w.add_shape(s)
    w.add_point_light((0, 5, -10), (1, 1, 1))
    return w
o = np.array([[0.1, 0.2, -5.0], [0.6, 0.1, -5.0]], dtype=np.float32)
d = np.array([[0, 0, 1.0], [0, 0, 1.0]], dtype=np.float32)
for kind in ("sphere", "cube", "cylinder", "cone", "plane"):
    for grouped in (False, True):
        if kind == "plane":
            dd = np.array([[0, -1.0, 1.0], [0, -1.0, 1.0]], dtype=np.float32)
        else:
            dd = d
        _, gt, gid, _ = build(host.new_world(0), kind, grouped).color_at(o, dd, 0)
        _, rt, rid, _ = build(oracle.new_world(), kind, grouped).color_at(o, dd, 0)
        print(kind, grouped, "device", list(gid), "oracle", list(rid), "t equal", np.array_equal(gt.view(np.uint32), rt.view(np.uint32)))
