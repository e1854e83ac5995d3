import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
lib = host.hip_lib()
sc = Scene.from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
cam = sc.camera(3840, 2160); rcam = hw.rt_camera(cam)
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
n = 3840 * 2160 * 3
pf, pu = lib.rt_host_alloc(ctx, n * 4), lib.rt_host_alloc(ctx, n)
f = np.empty(n, np.float32); u = np.empty(n, np.uint8)
for bands in ("1", "2", "3", "4"):
    os.environ["RT_BANDS"] = bands
    res = []
    for what, fn in (("f32 pageable", lambda: lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), f.ctypes.data, None, None)),
                     ("f32 pinned", lambda: lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), pf, None, None)),
                     ("u8 pageable", lambda: lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), 0, u.ctypes.data, None, None)),
                     ("u8 pinned", lambda: lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), 0, pu, None, None))):
        ts = []
        for _ in range(6):
            t = time.perf_counter(); assert fn() == 0; ts.append((time.perf_counter() - t) * 1e3)
        res.append(f"{what} {min(ts[1:]):.2f}")
    print("bands", bands, " | ".join(res), flush=True)
