import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from raytracer_amd import host
from raytracer_amd.scene import Scene
from raytracer_amd.canvas import to_u8
objs = ["scenes/obj/space_ship.obj", "scenes/obj/teapot_low.obj"]
for name, wh in [("basic_scene", None), ("cover", None), ("csg", None), ("checkered_cube", None), ("checkered_sphere", None),
                 ("checkered_plane", None), ("checkered_cylinder", None), ("space_teapot", None), ("soft_shadows", (320, 180))]:
    sc = Scene.from_file(f"scenes/{name}.yaml", obj_files=objs)
    cam = sc.camera(*wh) if wh else sc.camera()
    t = time.time(); g = sc.build(host.new_world(0)).render(cam, aux=True); tg = time.time() - t
    t = time.time(); r = sc.build(oracle.new_world()).render(cam, aux=True); tr = time.time() - t
    d = np.abs(to_u8(g.rgb).astype(int) - to_u8(r.rgb).astype(int)).max(axis=2)
    fd = np.abs(g.rgb - r.rgb).max()
    print(f"{name}: gpu {tg:.2f}s cpu {tr:.2f}s mask_eq={np.array_equal(g.hit_mask, r.hit_mask)} id_eq={np.array_equal(g.object_id, r.object_id)} "
          f"id_mismatch={(g.object_id != r.object_id).sum()} exact8={(d==0).mean()*100:.4f}% within1={(d<=1).mean()*100:.4f}% maxf32diff={fd:.3g} "
          f"gpu_stats={g.stats} ref_stats={r.stats}", flush=True)
