"""Interleaved scanline tiles: how one frame is split over N GPUs and put back together.

Part ``p`` of ``n`` renders the rows ``r`` with ``(r // tile_h) % n == p``, compacted
in increasing order — the same rule as RtRenderOptions.tile_h/n_parts/part (include/rt_abi.h).
Pixels are independent (src/camera.rs:73-77), so the assembled canvas is byte-identical for any n.
The exchange step is one all-gather of equal-sized row blocks (RCCL over xGMI on GPUs, gloo in
the CPU tests) followed by one row permutation.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def part_rows(vsize: int, tile_h: int, n_parts: int, part: int) -> np.ndarray:
    r = np.arange(vsize)
    return r[(r // tile_h) % n_parts == part]


def max_part_rows(vsize: int, tile_h: int, n_parts: int) -> int:
    return max(len(part_rows(vsize, tile_h, n_parts, p)) for p in range(n_parts))


def row_permutation(vsize: int, tile_h: int, n_parts: int) -> np.ndarray:
    """perm[g] = position of global row g inside the padded all-gather buffer
    (part-major, each part padded to max_part_rows)."""
    m = max_part_rows(vsize, tile_h, n_parts)
    perm = np.empty(vsize, dtype=np.int64)
    for p in range(n_parts):
        rows = part_rows(vsize, tile_h, n_parts, p)
        perm[rows] = p * m + np.arange(len(rows))
    return perm


def gather_canvas(local: torch.Tensor, vsize: int, tile_h: int, group=None, perm: torch.Tensor = None,
                  gathered: torch.Tensor = None) -> torch.Tensor:
    """All-gather every rank's compacted rows ``local`` (max_part_rows, hsize, 3; trailing rows of
    short parts are padding) and return the full (vsize, hsize, 3) canvas on every rank."""
    n = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((n * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    flat = gathered.view(n * local.shape[0], *local.shape[1:])  # concatenated form: accepted by RCCL and gloo
    dist.all_gather_into_tensor(flat, local.contiguous(), group=group)
    if perm is None:
        perm = torch.from_numpy(row_permutation(vsize, tile_h, n)).to(local.device)
    return flat.index_select(0, perm)


class CanvasPipeline:
    """Frames in flight: the all-gather of frame k runs beside the render of frame k+1.

    Two sets of buffers (this rank's rows, the gathered blocks, the assembled canvas), no allocation per frame.
    ``dtype=torch.uint8`` exchanges the 8-bit rows that rt_canvas_to_u8_device makes of a part (a quarter of the bytes:
    3.1 MB per GPU instead of 12.4 MB at 4K over 8 GPUs).
    Per frame: render into ``next_local()``, then ``submit()``, which starts that frame's all-gather without
    waiting for it and returns the assembled canvas of the frame before (None for the first); ``finish()`` returns
    the last one.  A returned canvas stays valid until the second ``submit()`` after it.  On GPUs the collective
    runs on RCCL's own stream; the waits are stream-side, so the host runs ahead."""

    def __init__(self, vsize: int, hsize: int, tile_h: int, device, group=None, dtype=torch.float32):
        self.group = group
        n = dist.get_world_size(group)
        m = max_part_rows(vsize, tile_h, n)
        self.local = [torch.zeros((m, hsize, 3), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [torch.empty((n * m, hsize, 3), dtype=dtype, device=device) for _ in range(2)]
        self.canvas = [torch.empty((vsize, hsize, 3), dtype=dtype, device=device) for _ in range(2)]
        self.perm = torch.from_numpy(row_permutation(vsize, tile_h, n)).to(device)
        self.pending = None
        self.frame = 0

    def next_local(self) -> torch.Tensor:
        return self.local[self.frame % 2]

    def submit(self):
        slot = self.frame % 2
        work = dist.all_gather_into_tensor(self.gathered[slot], self.local[slot], group=self.group, async_op=True)
        done = self.finish()
        self.pending = (work, slot)
        self.frame += 1
        return done

    def finish(self):
        if self.pending is None:
            return None
        work, slot = self.pending
        self.pending = None
        work.wait()
        torch.index_select(self.gathered[slot], 0, self.perm, out=self.canvas[slot])
        return self.canvas[slot]
