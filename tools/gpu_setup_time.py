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
