"""Work counters of single pixels of dragons (diagnostics).  usage: gpu_pixel.py W H px py [px py ...]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
W, H = int(sys.argv[1]), int(sys.argv[2])
pix = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(3, len(sys.argv), 2)]
lib = host.hip_lib(); torch.zeros(1, device="cuda")
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
cam = sc.camera(W, H)
P = C.c_void_p
for px, py in pix:
    o, d, _ = hw.ray_for_pixel(cam, px, py)
    for rem in (0, 1, 2, 3, 4):
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, abi.RT_RENDER_COUNT_WORK)
        rgb = np.zeros(3, np.float32); t = np.zeros(1, np.float32); ids = np.zeros(1, np.int32); uv = np.zeros(2, np.float32)
        st = abi.RtStats()
        rc = lib.rt_color_at(ctx, ds, C.byref(opts), o.ctypes.data_as(P), d.ctypes.data_as(P), 1, rem,
                             rgb.ctypes.data_as(P), t.ctypes.data_as(P), ids.ctypes.data_as(P), uv.ctypes.data_as(P), C.byref(st))
        print((px, py), "rem", rem, "rc", rc, "o", o, "d", d, "rgb", rgb, "t", t[0], "id", ids[0],
              {f: getattr(st, f) for f, _ in st._fields_ }, flush=True)
