"""Multi-GPU partition logic on the CPU: interleaved scanline tiles and the
all-gather + row permutation that reassembles the canvas, run with world_size-2/3 gloo processes.
Each simulated rank takes its rows from an oracle render (pixels are independent, src/camera.rs:73-77);
the GPU version of the same check is tests/test_gpu_parity.py::test_parts_reassemble."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import scene_path
from raytracer_amd import abi, host, tiles


@pytest.mark.parametrize("vsize,tile_h,n", [(2160, 2, 1), (2160, 2, 2), (2160, 2, 4), (2160, 2, 8), (480, 2, 8),
                                            (101, 3, 4), (7, 2, 8), (1, 1, 2), (16, 16, 3)])
def test_partition_covers_every_row_once(vsize, tile_h, n):
    rows = [tiles.part_rows(vsize, tile_h, n, p) for p in range(n)]
    allr = np.concatenate(rows)
    assert sorted(allr.tolist()) == list(range(vsize))
    m = tiles.max_part_rows(vsize, tile_h, n)
    perm = tiles.row_permutation(vsize, tile_h, n)
    assert len(set(perm.tolist())) == vsize and perm.max() < n * m
    for p in range(n):
        assert np.array_equal(perm[rows[p]], p * m + np.arange(len(rows[p])))
        # the C ABI computes the same row count (host-only call)
        cam = abi.RtCamera(64, vsize, 0.01, 1.0, 1.0)
        opts = abi.RtRenderOptions(0, 0.5, 0, tile_h, n, p, 0)
        assert host.hip_lib().rt_part_rows(C.byref(cam), C.byref(opts)) == len(rows[p])


def _worker(rank, world, port, full_path, tile_h, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.load(full_path))
        vsize = full.shape[0]
        mine = tiles.part_rows(vsize, tile_h, world, rank)
        m = tiles.max_part_rows(vsize, tile_h, world)
        local = torch.zeros((m,) + tuple(full.shape[1:]), dtype=full.dtype)
        local[: len(mine)] = full[torch.from_numpy(mine)]
        canvas = tiles.gather_canvas(local, vsize, tile_h)
        ok = torch.equal(canvas, full)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            np.save(out_path, np.array([int(flag.item())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_h,size", [(2, 2, (48, 32)), (3, 1, (40, 25)), (2, 4, (32, 30))])
def test_gather_canvas_gloo(tmp_path, world, tile_h, size):
    from oracle import oracle
    from raytracer_amd.scene import Scene
    scene = Scene.from_file(scene_path("basic_scene"))
    full = scene.build(oracle.new_world()).render(scene.camera(*size)).rgb
    fp, op = str(tmp_path / "full.npy"), str(tmp_path / "ok.npy")
    np.save(fp, full)
    port = 29600 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, fp, tile_h, op), nprocs=world, join=True)
    assert np.load(op)[0] == 1


def _pipeline_worker(rank, world, port, full_path, tile_h, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.load(full_path))
        vsize, hsize = full.shape[:2]
        mine = torch.from_numpy(tiles.part_rows(vsize, tile_h, world, rank))
        pipe = tiles.CanvasPipeline(vsize, hsize, tile_h, torch.device("cpu"))
        ok, done = True, []
        for k in range(5):  # frame k = the render scaled by k + 1: every frame must come back as itself, in order
            local = pipe.next_local()
            local.zero_()
            local[: len(mine)] = full[mine] * float(k + 1)
            out = pipe.submit()
            if out is not None:
                done.append(out.clone())
        done.append(pipe.finish().clone())
        ok = len(done) == 5 and all(torch.equal(c, full * float(k + 1)) for k, c in enumerate(done)) and pipe.finish() is None
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            np.save(out_path, np.array([int(flag.item())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_h,size", [(2, 8, (48, 37)), (3, 2, (40, 25))])
def test_canvas_pipeline_gloo(tmp_path, world, tile_h, size):
    """bench.py's N > 1 path: double-buffered frames, the gather of one frame overlapping the next render."""
    rng = np.random.RandomState(5)
    full = rng.uniform(0, 1, size=(size[1], size[0], 3)).astype(np.float32)
    fp, op = str(tmp_path / "full.npy"), str(tmp_path / "ok.npy")
    np.save(fp, full)
    port = 29950 + (os.getpid() % 300) + world
    mp.spawn(_pipeline_worker, args=(world, port, fp, tile_h, op), nprocs=world, join=True)
    assert np.load(op)[0] == 1


def _u8_worker(rank, world, port, full_path, tile_h, dither, out_path):
    from test_kat_canvas import to_u8
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = np.load(full_path)
        vsize, hsize = full.shape[:2]
        mine = tiles.part_rows(vsize, tile_h, world, rank)
        pipe = tiles.CanvasPipeline(vsize, hsize, tile_h, torch.device("cpu"), dtype=torch.uint8)
        local = pipe.next_local()
        # what rt_canvas_to_u8_device makes of this rank's f32 rows: dithered by their row in the whole image
        local[: len(mine)] = torch.from_numpy(to_u8(full[mine], dither, mine))
        pipe.submit()
        canvas = pipe.finish()
        ok = canvas.dtype == torch.uint8 and torch.equal(canvas, torch.from_numpy(to_u8(full, dither)))
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            np.save(out_path, np.array([int(flag.item())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_h,dither", [(2, 8, 0), (3, 2, 5), (2, 2, 3)])
def test_u8_rows_gather_to_the_quantised_frame(tmp_path, world, tile_h, dither):
    """The 8-bit exchange: every rank quantises / dithers its own rows, the all-gather moves a quarter of
    the bytes, and the assembled canvas is the whole frame's apply_dithering + to_ppm bytes."""
    rng = np.random.RandomState(8)
    full = rng.uniform(-0.1, 1.1, size=(45, 52, 3)).astype(np.float32)
    fp, op = str(tmp_path / "full.npy"), str(tmp_path / "ok.npy")
    np.save(fp, full)
    port = 30300 + (os.getpid() % 300) + world
    mp.spawn(_u8_worker, args=(world, port, fp, tile_h, dither, op), nprocs=world, join=True)
    assert np.load(op)[0] == 1
