"""What the pass over the ids costs (rt_hip.hip k_copies_pixels): dragons.yaml at 4K with every fourth triangle of the
mesh listed twice, ids requested.  Run under `rocprofv3 --kernel-trace --stats` and read the kernel's line."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import host, synth
from raytracer_amd.scene import Scene
mesh = synth.synth_dragon()
g = mesh[0]
rep = np.ones(len(g["vidx"]), dtype=np.int64)
rep[::4] = 2
g["vidx"] = np.repeat(g["vidx"], rep, axis=0)
g["nidx"] = np.repeat(g["nidx"], rep, axis=0)
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": mesh})
w = sc.build(host.new_world(0))
cam = sc.camera(3840, 2160)
for k in range(4):
    r = w.render(cam, aux=True)
ids = r.object_id
print("triangles per mesh", len(g["vidx"]), "pixels hit", int((ids >= 0).sum()), "frame ms", r.stats.get("kernel_ms"), flush=True)
