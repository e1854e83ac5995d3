"""Per-pixel traversal work map of dragons (diagnostics): where is the heaviest 8x8 tile and what does it do."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host, synth
from raytracer_amd.scene import Scene
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
lib = host.hip_lib(); dev = torch.device("cuda", 0)
sc = Scene.from_file(os.path.join(ROOT, "scenes", "dragons.yaml"), meshes={"dragon.obj": synth.synth_dragon()})
hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
rcam = hw.rt_camera(sc.camera(W, H))
out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
wm = torch.zeros((H, W), dtype=torch.float32, device=dev)
aux = abi.RtAux(None, None, wm.data_ptr())
opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, abi.RT_RENDER_COUNT_WORK | abi.RT_RENDER_WORK_MAP)
st = abi.RtStats()
torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), C.byref(aux), None, C.byref(st)) == 0
w = wm.cpu().numpy()
for bit, nm in ((0x10, "shadow"), (0x20, "collect")):
    o2 = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, abi.RT_RENDER_COUNT_WORK | abi.RT_RENDER_WORK_MAP | bit)
    assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(o2), out.data_ptr(), C.byref(aux), None, C.byref(st)) == 0
    w2 = wm.cpu().numpy()
    print("%s work: total %.3g of %.3g (%.1f%%); in the 1%% heaviest pixels %.1f%%" % (
        nm, w2.sum(), w.sum(), 100 * w2.sum() / w.sum(), 100 * w2[w >= np.percentile(w, 99)].sum() / w[w >= np.percentile(w, 99)].sum()))
print("per-pixel work: mean %.0f  median %.0f  p99 %.0f  max %.0f" % (w.mean(), np.median(w), np.percentile(w, 99), w.max()))
t = w[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
tmax = t.max(axis=(1, 3)); tmean = t.mean(axis=(1, 3))
print("per-tile max-lane work: mean %.0f  p50 %.0f p90 %.0f p99 %.0f  max %.0f" % (tmax.mean(), np.median(tmax), np.percentile(tmax, 90), np.percentile(tmax, 99), tmax.max()))
print("lane efficiency (sum of tile mean / sum of tile max): %.3f" % (tmean.sum() / tmax.sum()))
iy, ix = np.unravel_index(np.argmax(tmax), tmax.shape)
print("heaviest tile at tile (x=%d,y=%d) of (%d,%d): pixel work rows:" % (ix, iy, W // 8, H // 8))
print(np.array2string(t[iy, :, ix, :].astype(int), max_line_width=200))
# coarse map: 16 columns x 9 rows of mean tile-max
cm = tmax[: tmax.shape[0] // 9 * 9, : tmax.shape[1] // 16 * 16].reshape(9, tmax.shape[0] // 9, 16, tmax.shape[1] // 16).max(axis=(1, 3))
print("coarse map of max tile work (9 x 16):"); print(np.array2string(cm.astype(int), max_line_width=250))
