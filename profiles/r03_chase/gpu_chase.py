"""rt_render / rt_render_u8 into host memory: the single launch with the copy behind it (RT_CHASE_BANDS) against the
bands as launches of their own (RT_NO_CHASE), and the bytes against rt_render_device.  usage: gpu_chase.py [scene]"""
import ctypes as C, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "dragons"
lib = host.hip_lib()
objs = [os.path.join(ROOT, "scenes", "obj", f) for f in ("space_ship.obj", "teapot_low.obj")]
sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), obj_files=objs, meshes={"dragon.obj": synth.synth_dragon()} if name == "dragons" else None)
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
W, H = 3840, 2160
rcam = hw.rt_camera(sc.camera(W, H))
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
n = W * H * 3
dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), dev.data_ptr(), None, None, None) == 0
torch.cuda.synchronize()
want = dev.cpu().numpy().ravel()
pf, pu = lib.rt_host_alloc(ctx, n * 4), lib.rt_host_alloc(ctx, n)
f = np.empty(n, np.float32); u = np.empty(n, np.uint8)
vf = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_float)), shape=(n,))
vu = np.ctypeslib.as_array(C.cast(pu, C.POINTER(C.c_uint8)), shape=(n,))
u_ref = None
st = abi.RtStats()
for mode in ("nochase", "4", "8", "12", "16"):
    os.environ.pop("RT_NO_CHASE", None); os.environ.pop("RT_CHASE_BANDS", None)
    if mode == "nochase":
        os.environ["RT_NO_CHASE"] = "1"
    else:
        os.environ["RT_CHASE_BANDS"] = mode
    res = []
    for what, fn, got in (("f32 pageable", lambda: lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), f.ctypes.data, None, None), lambda: f),
                          ("f32 pinned", lambda: lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), pf, None, C.byref(st)), lambda: vf),
                          ("u8 pageable", lambda: lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), 0, u.ctypes.data, None, None), lambda: u),
                          ("u8 pinned", lambda: lib.rt_render_u8(ctx, ds, C.byref(rcam), C.byref(opts), 0, pu, None, None), lambda: vu)):
        ts = []
        for _ in range(8):
            got()[:] = 0
            t = time.perf_counter(); assert fn() == 0, lib.rt_last_error(ctx); ts.append((time.perf_counter() - t) * 1e3)
            if what.startswith("f32"):
                assert got().tobytes() == want.tobytes(), (mode, what, "bytes differ from rt_render_device")
            else:
                if u_ref is None:
                    u_ref = got().copy()
                assert np.array_equal(got(), u_ref), (mode, what, "8-bit bytes differ")
        res.append(f"{what} {min(ts[1:]):.2f}" + (f" (kernel {st.kernel_ms:.2f})" if what == "f32 pinned" else ""))
    print(f"{name} {mode:>8}", " | ".join(res), flush=True)
