"""Wall time of the host-buffer entry point rt_render (what the Rust shim calls): kernel + copy back."""
import ctypes as C, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
lib = host.hip_lib(); torch.zeros(1, device="cuda")
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
W, H = 3840, 2160
rcam = hw.rt_camera(sc.camera(W, H))
out = np.zeros((H, W, 3), np.float32)
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
st = abi.RtStats()
for i in range(6):
    t0 = time.perf_counter()
    rc = lib.rt_render(ctx, ds, C.byref(rcam), C.byref(opts), out.ctypes.data_as(C.c_void_p), None, C.byref(st))
    dt = (time.perf_counter() - t0) * 1e3
    assert rc == 0
    print(f"rt_render 4K dragons: wall {dt:.2f} ms (kernel {st.kernel_ms:.2f} ms, total_ms {st.total_ms:.2f}) checksum {float(out.astype(np.float64).sum()):.3f}", flush=True)
