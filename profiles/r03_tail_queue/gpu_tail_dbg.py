"""Statistics of the tail queue (library built with -DRT_TAIL_DEBUG prints them per synchronous launch).
usage: gpu_tail_dbg.py scene:WxH [n_parts]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

name, wh = sys.argv[1].split(":")
n_parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = (int(x) for x in wh.split("x"))
lib = host.hip_lib()
dev = torch.device("cuda", 0)
meshes = {"dragon.obj": synth.synth_dragon()} if name == "dragons" else None
sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), meshes=meshes)
hw = sc.build(host.new_world(0))
ctx, ds = hw.device_scene()
rcam = hw.rt_camera(sc.camera(W, H))
for flags in (abi.RT_RENDER_NO_TAIL, 0):
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 8, n_parts, 0, flags)
    rows = int(lib.rt_part_rows(C.byref(rcam), C.byref(opts)))
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device=dev)
    st = abi.RtStats()
    for k in range(12):
        torch.cuda.synchronize()
        sys.stderr.flush()
        rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st))
        assert rc == 0, lib.rt_last_error(ctx).decode()
        print(f"{sys.argv[1]} parts {n_parts} flags {flags}: kernel {st.kernel_ms * 1e3:.1f} us", file=sys.stderr, flush=True)
