"""GPU A/B: the auxiliary-BVH render must equal the linear (reference-order) render bit for bit.
usage: gpu_accel_ab.py [W H]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 480)
mesh = synth.synth_dragon()
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": mesh})
t = time.time()
hw = sc.build(host.new_world(0))
print("host build", round(time.time() - t, 2), flush=True)
t = time.time()
ctx, ds = hw.device_scene()
print("device scene (incl. aux build)", round(time.time() - t, 2), flush=True)
lib = host.hip_lib()
cam = sc.camera(W, H)
rcam = hw.rt_camera(cam)
dev = torch.device("cuda", 0)


def render(flags):
    opts = abi.RtRenderOptions(abi.RT_JITTER_CONST, 0.5, 1, 1, 1, 0, flags)
    out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    oid = torch.zeros((H, W), dtype=torch.int32, device=dev)
    ht = torch.zeros((H, W), dtype=torch.float32, device=dev)
    aux = abi.RtAux(None, oid.data_ptr(), ht.data_ptr())
    st = abi.RtStats()
    torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
    rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), C.byref(aux), None, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"rc={rc} {lib.rt_last_error(ctx).decode()}")
    stats = {k: getattr(st, k) for k, _ in abi.RtStats._fields_}
    return out.cpu().numpy(), oid.cpu().numpy(), ht.cpu().numpy(), stats


for name, flags in [("accel", 0), ("accel", 0), ("accel+count", abi.RT_RENDER_COUNT_WORK),
                    ("linear", abi.RT_RENDER_NO_ACCEL)]:
    r = render(flags)
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r[3].items()}, flush=True)
    if name == "accel":
        a = r
b = r
print("rgb bit-identical:", np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)),
      "ids identical:", np.array_equal(a[1], b[1]),
      "t identical:", np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)),
      "ndiff rgb:", int((a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2).sum()), flush=True)
