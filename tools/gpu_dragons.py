import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from raytracer_amd import host, synth, builder as B
from raytracer_amd.scene import Scene
from raytracer_amd.canvas import to_u8
mesh = synth.synth_dragon()
sc = Scene.from_file("scenes/dragons.yaml", meshes={"dragon.obj": mesh})
t = time.time(); gw = sc.build(host.new_world(0)); print("host build", time.time() - t, flush=True)
sizes = [(300, 120), (1200, 480)]
if "--4k" in sys.argv: sizes.append((3840, 2160))
res = {}
for wh in sizes:
    cam = sc.camera(*wh)
    t = time.time(); g = gw.render(cam, aux=True); print(wh, "gpu", time.time() - t, g.stats, flush=True)
    t = time.time(); g = gw.render(cam, aux=True); print(wh, "gpu again", time.time() - t, g.stats, flush=True)
    res[wh] = g
if "--oracle" in sys.argv:
    t = time.time(); ow = sc.build(oracle.new_world()); print("oracle build", time.time() - t, flush=True)
    for wh in sizes[:2 if "--oracle-native" in sys.argv else 1]:
        cam = sc.camera(*wh)
        t = time.time(); r = ow.render(cam, aux=True); print(wh, "oracle", time.time() - t, r.stats, flush=True)
        g = res[wh]
        d = np.abs(to_u8(g.rgb).astype(int) - to_u8(r.rgb).astype(int)).max(axis=2)
        print(wh, f"mask_eq={np.array_equal(g.hit_mask, r.hit_mask)} id_mismatch={(g.object_id != r.object_id).sum()} exact8={(d==0).mean()*100:.4f}% within1={(d<=1).mean()*100:.4f}% hitfrac={g.hit_mask.mean():.3f}", flush=True)
