"""The ray queue at the end of a launch (rt_device.h tail_donate), on and off: same bytes, kernel times.
usage: gpu_tail_ab.py [scene:WxH ...]   (whole frames and the eight parts of a frame)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

specs = sys.argv[1:] or ["dragons:3840x2160", "cover:3840x2160", "dragons:1920x1080"]
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
    outs = {}
    for n_parts, parts in ((1, [0]), (8, range(8))):
        for flags, label in ((abi.RT_RENDER_NO_TAIL, "off"), (0, "on")):
            times = []
            for part in parts:
                opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 8, n_parts, part, flags)
                rows = int(lib.rt_part_rows(C.byref(rcam), C.byref(opts)))
                out = torch.zeros((rows, W, 3), dtype=torch.float32, device=dev)
                ms, st = [], abi.RtStats()
                for _ in range(14):  # (every counter block of the ring gets its tail buffers on first use: 8 launches)
                    torch.cuda.synchronize()
                    rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st))
                    assert rc == 0, lib.rt_last_error(ctx).decode()
                    ms.append(st.kernel_ms)
                times.append(min(ms[9:]))
                key = (n_parts, part)
                if key in outs:
                    same = torch.equal(outs[key].view(torch.int32), out.view(torch.int32))
                    if not same:
                        d = (outs[key] != out).any(dim=2)
                        print(f"   MISMATCH {spec} parts {n_parts} part {part}: {int(d.sum())} pixels differ", flush=True)
                else:
                    outs[key] = out
            print(f"{spec:<20} parts {n_parts} tail {label:>3}: " + " ".join(f"{t:6.3f}" for t in times) + " ms", flush=True)
