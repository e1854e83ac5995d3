"""How much of a frame is tail?  Frames submitted alternately to two streams let the head of one frame fill the idle
CUs at the end of the other; the throughput gain over one stream is the tail's share.  (Experiment only: bench.py
times frames one after another on one stream.)"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

lib = host.hip_lib()
dev = torch.device("cuda", 0)
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0))
ctx, ds = hw.device_scene()
W, H = 3840, 2160
rcam = hw.rt_camera(sc.camera(W, H))
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 8, 1, 0, 0)
for n_streams in (1, 2, 3):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    outs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        n = 24
        t = time.perf_counter()
        for k in range(n):
            s = streams[k % n_streams]
            rc = lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), outs[k % n_streams].data_ptr(), None,
                                      C.c_void_p(s.cuda_stream), None)
            assert rc == 0
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    assert lib.rt_check(ctx, None) == 0
    print(f"{n_streams} stream(s): {best:.3f} ms per frame", flush=True)
ref = outs[0].clone()
print("frames equal:", all(torch.equal(ref, o) for o in outs))
