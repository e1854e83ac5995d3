"""How much of a launch is tail, whole frame and eighths (run with a -DRT_PHASE_TIMERS build of librt_hip.so: the
library prints when the waves of each launch end).  usage: gpu_tail.py [scene]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragons"
lib = host.hip_lib(); dev = torch.device("cuda", 0)
sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), meshes={"dragon.obj": synth.synth_dragon()} if name == "dragons" else None)
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
rcam = hw.rt_camera(sc.camera(3840, 2160))
out = torch.zeros((2160, 3840, 3), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for n, part in ((1, 0), (8, 0), (8, 3), (8, 7), (2, 1)):
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 8, n, part, 0)
    st = abi.RtStats()
    for rep in range(3):
        sys.stderr.flush()
        if rep == 2:
            print(f"--- part {part} of {n}", file=sys.stderr, flush=True)
        assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st)) == 0
    print(f"part {part} of {n}: {st.kernel_ms:.3f} ms", file=sys.stderr, flush=True)
