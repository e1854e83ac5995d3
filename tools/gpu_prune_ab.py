"""Default render vs RT_RENDER_NO_PRUNE (no cull spheres, no distance pruning) on the group-less scenes at 4K: the
culling must not change a bit.  usage: gpu_prune_ab.py [scene ...]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracer_amd import abi, host
from raytracer_amd.scene import Scene
lib = host.hip_lib(); dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["cover", "csg", "checkered_cube", "checkered_sphere", "soft_shadows", "basic_scene"]
for name in names:
    sc = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"))
    hw = sc.build(host.new_world(0)); ctx, ds = hw.device_scene()
    W, H = (1920, 1080) if name == "soft_shadows" else (3840, 2160)
    rcam = hw.rt_camera(sc.camera(W, H))
    res = {}
    for flags in (0, abi.RT_RENDER_NO_PRUNE):
        out = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        oid = torch.zeros((H, W), dtype=torch.int32, device=dev)
        ht = torch.zeros((H, W), dtype=torch.float32, device=dev)
        aux = abi.RtAux(None, oid.data_ptr(), ht.data_ptr())
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, flags)
        st = abi.RtStats(); ms = []
        for _ in range(4):
            torch.cuda.synchronize()  # torch's fills ran on the null stream, the render uses the context's stream
            assert lib.rt_render_device(ctx, ds, C.byref(rcam), C.byref(opts), out.data_ptr(), C.byref(aux), None, C.byref(st)) == 0
            ms.append(st.kernel_ms)
        res[flags] = (out.cpu().numpy(), oid.cpu().numpy(), ht.cpu().numpy(), min(ms))
    a, b = res[0], res[abi.RT_RENDER_NO_PRUNE]
    same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and \
        np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    print(f"{name} {W}x{H}: identical {same}  default {a[3]:.3f} ms  no-prune {b[3]:.3f} ms", flush=True)
    assert same, name
