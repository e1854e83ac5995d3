"""Rehearsal of bench.py's gather pipeline on one GPU (RCCL, world 1): every assembled canvas against a synchronous render."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, tiles  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, H = 1920, 1080
sc = Scene.from_file(os.path.join(ROOT, "scenes", "cover.yaml"))
hw = sc.build(host.new_world(0))
ctx, ds = hw.device_scene()
lib = host.hip_lib()
rcam = hw.rt_camera(sc.camera(W, H))
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 8, 1, 0, 0)
stream = torch.cuda.current_stream(dev)
assert stream.cuda_stream != 0
ref = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), ref.data_ptr(), None, C.c_void_p(stream.cuda_stream), None) == 0
torch.cuda.synchronize()
pipe = tiles.CanvasPipeline(H, W, 8, dev)
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_bad = 0
for k in range(n_frames + 1):
    if k < n_frames:
        mine = pipe.next_local()
        mine.fill_(float(k))  # stale data must not survive
        assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), mine.data_ptr(), None, C.c_void_p(stream.cuda_stream), None) == 0
        out = pipe.submit()
    else:
        out = pipe.finish()
    if out is not None:
        bad = int((out != ref).any(dim=2).sum())
        n_bad += bad
        print(f"frame {k - 1}: slot {(k - 1) % 2} mismatching pixels {bad}, local ok {bool(torch.equal(pipe.local[(k - 1) % 2], ref))}, "
              f"gathered ok {bool(torch.equal(pipe.gathered[(k - 1) % 2], ref))}", flush=True)
dist.destroy_process_group()
sys.exit(1 if n_bad else 0)
