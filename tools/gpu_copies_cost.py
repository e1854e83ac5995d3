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

# second case: six instances of one mesh with ONE material (every triangle has five equal shapes under other
# transforms: each is intersected once per hit, none is an entry at the hit's t)
from raytracer_amd import builder as B
mesh = synth.synth_dragon()[0]
w2 = host.new_world(0)
for k in range(6):
    g = w2.group([("scale", 0.25, 0.25, 0.25), ("translate", (k % 3 - 1) * 2.2, (k // 3) * 1.6 - 0.8, 0.0)])
    w2.group_add_mesh(g, B.Material(color=(0.8, 0.3, 0.2)), mesh["positions"], mesh["normals"], mesh["vidx"], mesh["nidx"])
    w2.group_divide(g, 100)
    w2.add_group(g)
w2.add_point_light((-5, 8, -10), (1, 1, 1))
cam2 = B.Camera(3840, 2160, 0.9, (0, 0.5, -7), (0, 0, 0), (0, 1, 0))
for k in range(4):
    r2 = w2.render(cam2, aux=True)
print("six instances of one material: pixels hit", int((r2.object_id >= 0).sum()), flush=True)
