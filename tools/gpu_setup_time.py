import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_TIMING"] = "1"
from raytracer_amd import host, synth
from raytracer_amd.scene import Scene
t = time.time(); mesh = synth.synth_dragon(); print("synth", round(time.time() - t, 2), flush=True)
sc = Scene.from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "dragons.yaml"), meshes={"dragon.obj": mesh})
for k in range(2):
    t0 = time.time(); hw = sc.build(host.new_world(0)); t1 = time.time(); hw.flatten(); t2 = time.time(); hw.device_scene(); t3 = time.time()
    print(f"build {t1 - t0:.3f}  flatten {t2 - t1:.3f}  device_scene {t3 - t2:.3f}  total {t3 - t0:.3f}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "one_material":
    # six instances of the mesh with ONE material: every triangle has five equal shapes (rt_scene_create links them)
    from raytracer_amd import builder as B
    m = mesh[0]
    for k in range(2):
        t0 = time.time()
        w2 = host.new_world(0)
        for j in range(6):
            g = w2.group([("scale", 0.25, 0.25, 0.25), ("translate", (j % 3 - 1) * 2.2, (j // 3) * 1.6 - 0.8, 0.0)])
            w2.group_add_mesh(g, B.Material(color=(0.8, 0.3, 0.2)), m["positions"], m["normals"], m["vidx"], m["nidx"])
            w2.group_divide(g, 100)
            w2.add_group(g)
        w2.add_point_light((-5, 8, -10), (1, 1, 1))
        t1 = time.time(); w2.flatten(); t2 = time.time(); w2.device_scene(); t3 = time.time()
        print(f"one material: build {t1 - t0:.3f}  flatten {t2 - t1:.3f}  device_scene {t3 - t2:.3f}  total {t3 - t0:.3f}", flush=True)
