"""Kernel time of dragons 4K rendered as n interleaved parts and at several sizes (tail-effect check)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
lib = host.hip_lib(); dev = torch.device("cuda", 0)
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
def run(W, H, n, part, tile_h=2):
    rcam = hw.rt_camera(sc.camera(W, H))
    out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, tile_h, n, part, 0)
    st = abi.RtStats(); ms = []
    for _ in range(4):
        torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
        assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), None, None, C.byref(st)) == 0
        ms.append(st.kernel_ms)
    return min(ms), st.primary + st.shadow + st.reflect + st.refract
for W, H in ((3840, 2160), (7680, 4320), (1920, 1080), (960, 540)):
    t, r = run(W, H, 1, 0)
    print(f"{W}x{H} full: {t:.3f} ms  {r / t / 1e3:.0f} Mrays/s", flush=True)
for n in (2, 8):
    ts = [run(3840, 2160, n, p)[0] for p in range(n)]
    print(f"4K as {n} parts: each {[round(x, 3) for x in ts]} sum {sum(ts):.3f} ms", flush=True)
for th in (8, 16, 32):
    for n in (2, 8):
        ts = [run(3840, 2160, n, p, th)[0] for p in range(n)]
        print(f"tile_h {th}: 4K as {n} parts: each {[round(x, 3) for x in ts]} max {max(ts):.3f} ms", flush=True)
