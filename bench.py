#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X render path on the headline workload.

A "step" is one full frame of the workload: every rank renders its interleaved scanline tiles
with the HIP kernels (scene already resident in HBM), then (N > 1) the canvases are all-gathered
over RCCL and de-interleaved; frames are double-buffered, so the all-gather of one frame runs beside the
render of the next (every frame is assembled on every rank before the clock stops).  value = rays of the whole frame / mean step time, in Mrays/s, where a
ray is one World::intersect-equivalent query (primary + shadow + reflect + refract).

Default workload: samples/scenes/dragons.yaml at 3840x2160 (BASELINE.json: "frame time 4K dragons")
with the deterministic synthetic stand-in for dragon.obj (the real file is not redistributable
here; set RT_ASSETS=/dir/with/dragon.obj to use it).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import abi, host, synth, tiles  # noqa: E402
from raytracer_amd.obj import parse_obj  # noqa: E402
from raytracer_amd.scene import Scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_NO_FMA = 39.3  # TFLOP/s: 256 CUs x 64 lanes x 2.4 GHz, one mul or add per lane per cycle (no FMA: SURVEY.md Q11)


def load_workload(name: str):
    meshes, label = {}, ""
    if name == "dragons":
        assets = os.environ.get("RT_ASSETS", "")
        real = os.path.join(assets, "dragon.obj") if assets else ""
        if real and os.path.exists(real):
            mesh, label = parse_obj(real), "real dragon.obj"
        else:
            mesh, label = synth.synth_dragon(), "synthetic dragon"
        label += " sha256=" + synth.mesh_sha256(mesh)[:16]
        meshes["dragon.obj"] = mesh
    objs = [os.path.join(ROOT, "scenes", "obj", f) for f in ("space_ship.obj", "teapot_low.obj")]
    scene = Scene.from_file(os.path.join(ROOT, "scenes", f"{name}.yaml"), obj_files=objs, meshes=meshes)
    return scene, label


def main_multi_abi(args, json_fd) -> int:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rc = 0
    if rank == 0:
        devices = [int(x) for x in os.environ["RT_BENCH_DEVICES"].split(",")] if os.environ.get("RT_BENCH_DEVICES") else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but RT_BENCH_DEVICES names {len(devices)} devices")
        lib = host.hip_lib()
        scene, mesh_label = load_workload(args.workload)
        cam = scene.camera(args.width, args.height)
        t0 = time.time()
        worlds = [scene.build(host.new_world(d)) for d in devices]
        pairs = [w.device_scene() for w in worlds]
        setup_s = time.time() - t0
        n = len(devices)
        ctxs = (C.c_void_p * n)(*[p[0].value for p in pairs])
        scenes = (C.c_void_p * n)(*[p[1].value for p in pairs])
        rcam = worlds[0].rt_camera(cam)
        dev0 = torch.device("cuda", devices[0])
        canvas = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.float32, device=dev0)
        torch.cuda.synchronize(dev0)
        opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, 1, 0, 0)
        stats = (abi.RtStats * n)()

        def frame():
            r = lib.rt_render_multi_device(ctxs, scenes, n, C.byref(rcam), C.byref(opts), canvas.data_ptr(), stats)
            if r != abi.RT_OK:
                raise RuntimeError(f"rt_render_multi_device failed ({r}): {lib.rt_last_error(ctxs[0]).decode()}")

        for _ in range(args.warmup):
            frame()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame()  # (synchronous: the canvas is complete in GPU 0's HBM when it returns)
        elapsed = time.perf_counter() - t0
        rays = sum(s.primary + s.shadow + s.reflect + s.refract for s in stats)
        ms = elapsed / args.steps * 1e3
        out = {"metric": "Mrays/s (primary+shadow+secondary)", "value": round(rays / ms / 1e3, 2), "unit": "Mrays/s",
               "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if "synthetic" in mesh_label or args.workload != "dragons" else "real",
               "config": {"workload": f"{args.workload}.yaml {cam.hsize}x{cam.vsize} recursion_limit 4 {mesh_label}".strip(),
                          "tile_h": args.tile_h, "parts": n, "devices": devices, "rays_per_frame": int(rays),
                          "entry_point": "rt_render_multi_device (one process, RCCL gather into the first GPU's HBM)",
                          "part_kernel_ms": [round(s.kernel_ms, 3) for s in stats], "scene_setup_s": round(setup_s, 2)},
               "roofline": None, "cpu_baseline": None}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dragons")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--tile-h", type=int, default=8)  # a multiple of 8 keeps every 8x8 wave tile inside one stripe
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # only the timed loop and the roofline launches: no host-frame / eighths side measurements (tools/gpu_profile.sh uses
    # it so that every dispatch of the render kernel in a profile is a full frame of the workload)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--force-gather", action="store_true")  # rehearse the N > 1 exchange step on one GPU (RCCL, world 1)
    # N GPUs driven by ONE process through the C ABI (rt_render_multi_device: RCCL gather into GPU 0's HBM), as a host
    # in the reference's language would do it; under torchrun rank 0 drives them and the other ranks only wait.
    # RT_BENCH_DEVICES=0,0 rehearses N = 2 on a box with one GPU (two contexts on it).
    ap.add_argument("--multi-abi", action="store_true")
    ap.add_argument("--cpu-sample-pixels", type=int, default=900000)  # ~15 s of oracle work on 16 threads (the box's CPU share)
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: libraries that print there (RCCL's version banner) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.multi_abi:
        raise SystemExit(main_multi_abi(args, json_fd))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: start one rank per GPU (fresh child processes, nothing here has
        # touched the GPU yet) and relay rank 0's JSON line; under torchrun the ranks already exist
        from raytracer_amd.launch import launch_ranks
        os.dup2(json_fd, 1)
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU render path in the product")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one stream of our own for the render launches, torch's copies and RCCL's ordering: torch's default stream is
    # the null stream, whose handle 0 rt_render_device reads as "the context's stream" (include/rt_abi.h)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    gather = world > 1 or args.force_gather
    if gather:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene, mesh_label = load_workload(args.workload)
    cam = scene.camera(args.width, args.height)
    t0 = time.time()
    hw = scene.build(host.new_world(local_rank))
    ctx, dscene = hw.device_scene()
    setup_s = time.time() - t0
    lib = host.hip_lib()
    rcam = hw.rt_camera(cam)
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, world, rank, 0)
    rows = int(lib.rt_part_rows(C.byref(rcam), C.byref(opts)))
    max_rows = tiles.max_part_rows(cam.vsize, args.tile_h, world)
    # N > 1: frames are double-buffered, the all-gather of one frame (RCCL's stream) runs beside the next render
    pipe = tiles.CanvasPipeline(cam.vsize, cam.hsize, args.tile_h, dev) if gather else None
    local = pipe.local[0] if gather else torch.zeros((max_rows, cam.hsize, 3), dtype=torch.float32, device=dev)
    assert local.shape[0] == max_rows
    stream = torch.cuda.current_stream(dev)

    def check(rc, what):
        if rc != abi.RT_OK:
            raise RuntimeError(f"{what} failed ({rc}): {lib.rt_last_error(ctx).decode()}")

    def step():
        mine = pipe.next_local() if gather else local
        check(lib.rt_render_device(ctx, dscene, C.byref(rcam), C.byref(opts), mine.data_ptr(), None,
                                   C.c_void_p(stream.cuda_stream), None), "rt_render_device")
        return pipe.submit() if gather else mine  # the canvas of the frame before (None for the first) when gathering

    def sync():  # every frame submitted so far is assembled on every rank when this returns
        if gather:
            pipe.finish()
        torch.cuda.synchronize(dev)
        if gather:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # rays of this rank's part (deterministic) + work counters, from one instrumented launch
    copts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, world, rank, abi.RT_RENDER_COUNT_WORK)
    cst = abi.RtStats()
    check(lib.rt_render_device(ctx, dscene, C.byref(rcam), C.byref(copts), local.data_ptr(), None,
                               C.c_void_p(stream.cuda_stream), C.byref(cst)), "rt_render_device(count)")
    my_rays = cst.primary + cst.shadow + cst.reflect + cst.refract

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if gather:  # outside the timed region: this rank's rows of the last assembled canvas are the rows it rendered
        last = (pipe.frame - 1) % 2
        mine_rows = torch.from_numpy(tiles.part_rows(cam.vsize, args.tile_h, world, rank)).to(dev)
        if not torch.equal(pipe.canvas[last].index_select(0, mine_rows), pipe.local[last][: len(mine_rows)]):
            raise RuntimeError("gathered canvas does not hold this rank's rows")
    tot = torch.tensor([elapsed, float(my_rays), float(cst.rays_reference)], dtype=torch.float64, device=dev)
    if gather:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_rays, ref_rays = float(tot[1]), float(tot[2])
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays / (ms_per_step * 1e-3) / 1e6

    # dominant kernel: average launch duration from HIP events on the render stream
    kms = []
    st = abi.RtStats()
    for _ in range(max(3, args.steps)):
        check(lib.rt_render_device(ctx, dscene, C.byref(rcam), C.byref(opts), local.data_ptr(), None,
                                   C.c_void_p(stream.cuda_stream), C.byref(st)), "rt_render_device(stats)")
        kms.append(st.kernel_ms)
    kernel_ms = float(np.mean(kms))
    # Work of one launch from the instrumented kernel (RT_RENDER_COUNT_WORK), priced with the per-unit figures of
    # SURVEY.md §8(d) / DESIGN.md "Measurement".  These bytes are what the algorithm reads per lane; the walk fetches
    # wave-uniform records once per wave and the scene is cache resident, so they are NOT HBM traffic: they are reported
    # as `algorithmic` (cache-level), never as the roofline fraction.
    alg_bytes = (cst.node_visits * 32 + cst.tri_tests * 36 + cst.shaded_hits * 36 + cst.node_xforms * 48
                 + rows * cam.hsize * 12)
    alg_flops = (cst.node_visits * 12 + cst.node_xforms * 33 + cst.aux_tests * 18 + cst.tri_tests * 24
                 + cst.shape_tests * 30 + cst.shaded_hits * 140 + cst.primary * 36)
    ksec = kernel_ms * 1e-3
    valu_tflops = alg_flops / ksec / 1e12
    # HBM bytes per launch: the rocprofv3 PMC passes of this same command committed under profiles/ (FETCH_SIZE
    # doubled: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md "HBM"); null when no pass matches this
    # workload and kernel (bench.py cannot run the profiler on itself)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("workload") == f"{args.workload}.yaml {cam.hsize}x{cam.vsize}" and tj.get("parts") == world:
            traffic = int(2 * tj["fetch_kb"] * 1024 + tj["write_kb"] * 1024)
            traffic_src = tj.get("source", "profiles/traffic.json")
    hbm_gbs = traffic / ksec / 1e9 if traffic else None
    hbm_frac = hbm_gbs / HBM_PEAK_GBS if traffic else 0.0
    valu_frac = valu_tflops / VALU_PEAK_NO_FMA
    # The kernel is bound by instruction issue and memory latency (lane utilisation, dependent fetches), far below both
    # roofs.  The fraction reported is the one of USEFUL work: algorithmic flops against the VALU roof.  The measured HBM
    # traffic is given beside it (`hbm`), not as the fraction: in the four-wave kernel two thirds of it are spilled
    # registers going to scratch and back, bytes the algorithm does not ask for.
    roofline = {"bound": "valu", "achieved": round(valu_tflops, 3), "peak": VALU_PEAK_NO_FMA, "unit": "TFLOP/s",
                "frac": round(valu_frac, 5)}
    roofline.update({
        "traffic": traffic, "traffic_source": traffic_src,
        "kernel": "k_render_jobs", "kernel_ms": round(kernel_ms, 3),
        "limiter": "issue/latency: divergent traversal, see profiles/ (SQ passes) and DESIGN.md section 4",
        "hbm": {"achieved": round(hbm_gbs, 2) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 5) if traffic else None},
        "valu": {"achieved": round(valu_tflops, 3), "unit": "TFLOP/s", "peak_no_fma_scalar": VALU_PEAK_NO_FMA,
                 "peak_fma_packed": 157.3, "frac": round(valu_frac, 5), "algorithmic_flops": int(alg_flops),
                 "note": "bit-exact parity forbids FMA contraction (SURVEY.md Q11): 39.3 TFLOP/s is the ceiling of "
                         "scalar-lane mul/add on 256 CUs at 2.4 GHz"},
        "algorithmic": {"bytes": int(alg_bytes), "gbps": round(alg_bytes / ksec / 1e9, 2),
                        "frac_of_hbm_peak": round(alg_bytes / ksec / 1e9 / HBM_PEAK_GBS, 5),
                        "note": "SURVEY.md 8(d) units only (node visits 32 B, triangle tests 36 B, hit normals 36 B, "
                                "node transforms 48 B, output 12 B/pixel), per lane: served by scalar loads and "
                                "L1/L2, not by HBM"},
        "counters": {"node_visits": cst.node_visits, "aux_box_tests": cst.aux_tests, "tri_tests": cst.tri_tests,
                     "shape_tests": cst.shape_tests, "shaded_hits": cst.shaded_hits, "node_xforms": cst.node_xforms}})

    # the same frame through rt_render: host buffers out (pageable memory), the PCIe-inclusive figure (never `value`)
    host_frame_ms = None
    host_frame_u8_ms = None
    host_pinned_ms = None
    eighths_ms = None
    setup_again_s = None
    if world == 1 and rank == 0 and not args.no_extras:
        # the same world built and uploaded a second time: set-up without what a process pays once (HIP start-up, the
        # context, the code object)
        t0 = time.time()
        hw2 = scene.build(host.new_world(local_rank))
        hw2.device_scene()
        setup_again_s = round(time.time() - t0, 3)
        del hw2
        hbuf = np.empty((cam.vsize, cam.hsize, 3), dtype=np.float32)
        fopts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
        hms = []
        for _ in range(4):
            th = time.perf_counter()
            check(lib.rt_render(ctx, dscene, C.byref(rcam), C.byref(fopts), hbuf.ctypes.data, None, None), "rt_render")
            hms.append((time.perf_counter() - th) * 1e3)
        host_frame_ms = round(min(hms[1:]), 3)
        # and through rt_render_u8: quantised on the device (canvas.rs:73-75), a quarter of the bytes over PCIe
        ubuf = np.empty((cam.vsize, cam.hsize, 3), dtype=np.uint8)
        ums = []
        for _ in range(4):
            th = time.perf_counter()
            check(lib.rt_render_u8(ctx, dscene, C.byref(rcam), C.byref(fopts), abi.RT_DITHER_NONE, ubuf.ctypes.data, None,
                                   None), "rt_render_u8")
            ums.append((time.perf_counter() - th) * 1e3)
        host_frame_u8_ms = round(min(ums[1:]), 3)
        # the same two calls into page-locked host buffers (rt_host_alloc): the copy is a DMA beside the next band
        npx = cam.vsize * cam.hsize * 3
        pf, pu = lib.rt_host_alloc(ctx, npx * 4), lib.rt_host_alloc(ctx, npx)
        if pf and pu:
            pms, qms = [], []
            for _ in range(4):
                th = time.perf_counter()
                check(lib.rt_render(ctx, dscene, C.byref(rcam), C.byref(fopts), pf, None, None), "rt_render(pinned)")
                pms.append((time.perf_counter() - th) * 1e3)
            for _ in range(4):
                th = time.perf_counter()
                check(lib.rt_render_u8(ctx, dscene, C.byref(rcam), C.byref(fopts), abi.RT_DITHER_NONE, pu, None, None),
                      "rt_render_u8(pinned)")
                qms.append((time.perf_counter() - th) * 1e3)
            host_pinned_ms = [round(min(pms[1:]), 3), round(min(qms[1:]), 3)]
            pin_u8 = np.ctypeslib.as_array(C.cast(pu, C.POINTER(C.c_uint8)), shape=(npx,))
            if not np.array_equal(pin_u8, ubuf.reshape(-1)):
                raise RuntimeError("page-locked and pageable 8-bit frames differ")
        lib.rt_host_free(ctx, pf)
        lib.rt_host_free(ctx, pu)
        # strong-scaling preview: kernel time of each eighth of the frame (what each of 8 GPUs would render)
        # (the host-frame measurements above leave the GPU idle during their copies: a few whole frames first, so that the
        # parts are timed at the clocks a GPU that renders part after part runs at)
        for _ in range(5):
            check(lib.rt_render_device(ctx, dscene, C.byref(rcam), C.byref(opts), local.data_ptr(), None,
                                       C.c_void_p(stream.cuda_stream), None), "rt_render_device(warm)")
        eighths_ms = []
        for p in range(8):
            po = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, 8, p, 0)
            best = 1e9
            for _ in range(5):
                check(lib.rt_render_device(ctx, dscene, C.byref(rcam), C.byref(po), local.data_ptr(), None,
                                           C.c_void_p(stream.cuda_stream), C.byref(st)), "rt_render_device(part)")
                best = min(best, st.kernel_ms)
            eighths_ms.append(round(best, 3))

    # the other BASELINE.json configurations at 3840x2160, a few frames each, outside the timed region: the driver's
    # line then carries every config's frame time (BASELINE.md section 4 is filled from it)
    other = None
    if world == 1 and rank == 0 and not args.no_extras and args.workload == "dragons":
        other = {}
        for name in ("cover", "csg", "checkered_sphere", "soft_shadows"):
            osc, _ = load_workload(name)
            ocam = osc.camera(args.width, args.height)
            ohw = osc.build(host.new_world(local_rank))
            octx, odsc = ohw.device_scene()
            orcam = ohw.rt_camera(ocam)
            obuf = torch.empty((ocam.vsize, ocam.hsize, 3), dtype=torch.float32, device=dev)
            oo = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, 1, 0, 0)
            oc = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, args.tile_h, 1, 0, abi.RT_RENDER_COUNT_WORK)
            ost = abi.RtStats()

            def ocheck(rc, what):
                if rc != abi.RT_OK:
                    raise RuntimeError(f"{what} failed ({rc}): {lib.rt_last_error(octx).decode()}")

            ocheck(lib.rt_render_device(octx, odsc, C.byref(orcam), C.byref(oc), obuf.data_ptr(), None,
                                        C.c_void_p(stream.cuda_stream), C.byref(ost)), f"rt_render_device({name}, count)")
            orays = ost.primary + ost.shadow + ost.reflect + ost.refract
            for _ in range(3):
                ocheck(lib.rt_render_device(octx, odsc, C.byref(orcam), C.byref(oo), obuf.data_ptr(), None,
                                            C.c_void_p(stream.cuda_stream), None), f"rt_render_device({name})")
            torch.cuda.synchronize(dev)
            nfr = 10
            th = time.perf_counter()
            for _ in range(nfr):
                ocheck(lib.rt_render_device(octx, odsc, C.byref(orcam), C.byref(oo), obuf.data_ptr(), None,
                                            C.c_void_p(stream.cuda_stream), None), f"rt_render_device({name})")
            ocheck(lib.rt_check(octx, C.c_void_p(stream.cuda_stream)), f"rt_check({name})")
            fms = (time.perf_counter() - th) / nfr * 1e3
            other[f"{name}.yaml {ocam.hsize}x{ocam.vsize}"] = {
                "frame_ms": round(fms, 3), "mrays_s": round(orays / fms / 1e3, 1), "rays_per_frame": int(orays),
                "rays_reference": int(ost.rays_reference), "frames": nfr}
            del ohw, obuf

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        from raytracer_amd.builder import JITTER_COUNTER, RenderOptions
        stepn = max(1, int(round((cam.hsize * cam.vsize / args.cpu_sample_pixels) ** 0.5)))
        # the threads this process may use: its CPU affinity, and on a shared box the share the harness states in
        # OMP_NUM_THREADS (os.cpu_count() there is the whole machine, many times this job's share); both counts and the
        # CPU model go into the line
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if os.environ.get("OMP_NUM_THREADS", "").isdigit():
            threads = max(1, min(threads, int(os.environ["OMP_NUM_THREADS"])))
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        ow = scene.build(oracle.new_world())
        ro = RenderOptions(jitter_mode=JITTER_COUNTER, seed=1, threads=threads, row_step=stepn, col_step=stepn)
        r = ow.render(cam, ro)
        rays = r.stats["primary"] + r.stats["shadow"] + r.stats["reflect"] + r.stats["refract"]
        cpu_baseline = {"value": round(rays / r.stats["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": threads,
                        "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
                        "sample": f"every {stepn}th row and column of the same frame "
                                  f"({r.stats['primary']} primary rays, {r.stats['seconds']:.1f} s); C++ restatement "
                                  f"of the reference (oracle/), not the Rust reference itself"}

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+shadow+secondary)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            # the same frame through rt_render (the boundary: canvas in page-locked HOST memory, PCIe included); `value`
            # is the frame left in HBM (rt_render_device)
            "value_at_boundary": round(total_rays / (host_pinned_ms[0] * 1e-3) / 1e6, 2) if host_pinned_ms else None,
            "value_at_boundary_u8": round(total_rays / (host_pinned_ms[1] * 1e-3) / 1e6, 2) if host_pinned_ms else None,
            "data": "synthetic" if "synthetic" in mesh_label or args.workload != "dragons" else "real",
            "config": {"workload": f"{args.workload}.yaml {cam.hsize}x{cam.vsize} recursion_limit 4 {mesh_label}".strip(),
                       "tile_h": args.tile_h, "parts": world, "rays_per_frame": int(total_rays),
                       "rays_traced": int(total_rays),
                       # what the reference would cast: it re-traces secondary rays once per light
                       "rays_reference": int(ref_rays),
                       "frame_ms": round(ms_per_step, 3), "host_frame_ms": host_frame_ms, "host_frame_u8_ms": host_frame_u8_ms,
                       "host_frame_pinned_ms": host_pinned_ms[0] if host_pinned_ms else None,
                       "host_frame_u8_pinned_ms": host_pinned_ms[1] if host_pinned_ms else None,
                       "eighths_kernel_ms": eighths_ms, "scene_setup_s": round(setup_s, 2), "scene_setup_again_s": setup_again_s,
                       "other_workloads": other},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
