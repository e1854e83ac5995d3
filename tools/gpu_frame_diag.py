import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_fuzz as F
from oracle import oracle
from raytracer_amd import abi, host, builder as B
seed = int(sys.argv[1])
rng = np.random.RandomState(5000 + seed)
gw, ow = host.new_world(0), oracle.new_world()
F.mixed_world(gw, np.random.RandomState(seed), True, bool(seed % 2))
F.mixed_world(ow, np.random.RandomState(seed), True, bool(seed % 2))
eye = rng.uniform(-11, 11, 3); eye[1] = abs(eye[1]) + 1.0
cam = B.Camera(int(rng.randint(17, 230)), int(rng.randint(9, 140)), float(rng.uniform(0.5, 1.6)), tuple(eye), tuple(rng.uniform(-2, 2, 3)), (0, 1, 0))
ro = B.RenderOptions(jitter_mode=B.JITTER_COUNTER if seed % 3 else B.JITTER_CONST, seed=seed)
g = gw.render(cam, ro, aux=True); r = ow.render(cam, ro, aux=True)
bad = np.argwhere(g.object_id != r.object_id)
print("size", cam.hsize, cam.vsize, "bad pixels", len(bad), bad[:10].tolist())
arr = abi.scene_arrays(gw.flatten()); shapes = arr["shapes"]
def desc(s):
    if s < 0: return "miss"
    q = shapes[s]; return f"{s}:kind{int(q['kind_flags']) & 15}/parent{int(q['parent'])}/prim{int(q['prim'])}/vclass{int(q['vclass'])}"
lib = host.hip_lib(); ctx, ds = gw.device_scene(); P = C.c_void_p
for (y, x) in bad[:6]:
    print(" pixel", x, y, "gpu", desc(int(g.object_id[y, x])), "t", repr(g.hit_t[y, x]), "oracle", desc(int(r.object_id[y, x])), "t", repr(r.hit_t[y, x]))
    o, d, _ = gw.ray_for_pixel(cam, int(x), int(y))
    for flags in (0, abi.RT_RENDER_NO_PRUNE, abi.RT_RENDER_NO_ACCEL, abi.RT_RENDER_NO_PRUNE | abi.RT_RENDER_NO_ACCEL, abi.RT_RENDER_SHORT_STACK):
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, flags)
        rgb = np.zeros(3, np.float32); t = np.zeros(1, np.float32); ids = np.zeros(1, np.int32); uv = np.zeros(2, np.float32)
        oo, dd = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        lib.rt_color_at(ctx, ds, C.byref(opts), oo.ctypes.data_as(P), dd.ctypes.data_as(P), 1, 0, rgb.ctypes.data_as(P), t.ctypes.data_as(P), ids.ctypes.data_as(P), uv.ctypes.data_as(P), None)
        print(f"   color_at flags {flags}: id {ids[0]} t {t[0]!r}")
    print("   ray", o, d)

    tg = arr["tri_geom"]
    par = int(shapes[int(g.object_id[y, x])]["parent"])
    same = [i for i in range(len(shapes)) if int(shapes[i]["parent"]) == par and (int(shapes[i]["kind_flags"]) & 15) >= 5]
    of, df = o.astype(np.float32), d.astype(np.float32)
    hits = []
    for i in same:
        q = tg[int(shapes[i]["prim"])]
        p1, e1, e2 = q["p1"].astype(np.float32), q["e1"].astype(np.float32), q["e2"].astype(np.float32)
        dxe2 = np.cross(df, e2).astype(np.float32); det = np.float32(np.dot(e1, dxe2))
        if abs(det) < 1e-4: continue
        f = np.float32(1.0) / det; p1o = (of - p1).astype(np.float32); u = np.float32(f * np.float32(np.dot(p1o, dxe2)))
        if u < 0 or u > 1: continue
        oxe1 = np.cross(p1o, e1).astype(np.float32); v = np.float32(f * np.float32(np.dot(df, oxe1)))
        if v < 0 or u + v > 1: continue
        hits.append((float(np.float32(f * np.float32(np.dot(e2, oxe1)))), i, int(shapes[i]["vclass"])))
    print("   triangles of the group hit (numpy f32, t / shape / class):", sorted(hits)[:6])
