"""Kernel time of a few workloads on the GPU (experiments).  usage: gpu_time.py [label] [scene:WxH ...]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""
specs = sys.argv[2:] or ["dragons:3840x2160", "cover:3840x2160", "soft_shadows:1920x1080", "csg:3840x2160"]
lib = host.hip_lib()
dev = torch.device("cuda", 0)
objs = [os.path.join(ROOT, "scenes", "obj", f) for f in ("space_ship.obj", "teapot_low.obj")]
for spec in specs:
    name, wh = spec.split(":")
    W, H = (int(x) for x in wh.split("x"))
    meshes = {"dragon.obj": synth.synth_dragon()} if name == "dragons" else None
    sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), obj_files=objs, meshes=meshes)
    hw = sc.build(host.new_world(0))
    ctx, ds = hw.device_scene()
    rcam = hw.rt_camera(sc.camera(W, H))
    out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
    ms, st = [], abi.RtStats()
    for _ in range(6):
        torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
        rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st))
        assert rc in (0, -5), lib.rt_last_error(ctx).decode()  # -5: list overflow (experiments with small caps)
        ms.append(st.kernel_ms)
    rays = st.primary + st.shadow + st.reflect + st.refract
    print(f"{label:>12} {spec:<24} min {min(ms):8.3f} ms  med {sorted(ms)[len(ms) // 2]:8.3f} ms  "
          f"{rays / min(ms) / 1e3:9.1f} Mrays/s  checksum {float(out.double().sum()):.6f}", flush=True)
