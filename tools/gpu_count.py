"""Work counters of a render (RT_RENDER_COUNT_WORK).  usage: gpu_count.py scene:WxH ..."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

lib = host.hip_lib()
dev = torch.device("cuda", 0)
objs = [os.path.join(ROOT, "scenes", "obj", f) for f in ("space_ship.obj", "teapot_low.obj")]
for spec in sys.argv[1:]:
    name, wh = spec.split(":")
    W, H = (int(x) for x in wh.split("x"))
    meshes = {"dragon.obj": synth.synth_dragon()} if name == "dragons" else None
    sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), obj_files=objs, meshes=meshes)
    hw = sc.build(host.new_world(0))
    ctx, ds = hw.device_scene()
    rcam = hw.rt_camera(sc.camera(W, H))
    out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    for flags in (0, abi.RT_RENDER_COUNT_WORK):
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, flags)
        st = abi.RtStats()
        for _ in range(2):
            torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
            rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st))
            assert rc in (0, -5), lib.rt_last_error(ctx).decode()
        print(spec, "flags", flags, {f: getattr(st, f) for f, _ in st._fields_}, flush=True)
