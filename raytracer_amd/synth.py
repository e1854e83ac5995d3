"""Deterministic synthetic stand-in for dragon.obj (the real file is not in the reference tree:
samples/scenes/dragons.yaml:1 tells the user to download it; there is no network here).

``synth_dragon()`` is a closed, bumpy, smooth-normal torus-knot tube of 660 x 660 x 2 = 871 200
triangles (the real dragon has ~871 k), scaled to fill the scene's raw bounding box
x in [-3.9863, 3.4804], y in [-0.1217, 5.0473], z in [-1.1820, 2.0746] (dragons.yaml:39-43),
one OBJ group, per-vertex normals (=> SmoothTriangle, src/obj.rs:24-33).  Results quoted on it
say "synthetic dragon" and carry the mesh sha256.
"""
from __future__ import annotations

import hashlib

import numpy as np

RAW_BBOX = ((-3.9863, 3.4804), (-0.1217, 5.0473), (-1.1820, 2.0746))


def synth_dragon(nu: int = 660, nv: int = 660, seed: int = 0xD7A60) -> list:
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    u = (np.arange(nu, dtype=np.float64) / nu) * 2 * np.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2 * np.pi
    U, V = np.meshgrid(u, v, indexing="ij")
    p, q = 2.0, 3.0

    def curve(t):
        r = 2.0 + np.cos(q * t)
        return np.stack([r * np.cos(p * t), np.sin(q * t) * 1.2, r * np.sin(p * t)], axis=-1)

    eps = 1e-4
    c = curve(U)
    t = curve(U + eps) - curve(U - eps)
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=-1, keepdims=True)
    n2 = np.cross(t, n1)
    radius = 0.42 * (1.0 + 0.18 * np.sin(9 * V + ph[0]) * np.sin(14 * U + ph[1])
                     + 0.07 * np.sin(31 * U + ph[2]) * np.cos(23 * V + ph[3]))
    pos = c + radius[..., None] * (np.cos(V)[..., None] * n1 + np.sin(V)[..., None] * n2)
    # scale to the raw bbox
    lo, hi = pos.reshape(-1, 3).min(axis=0), pos.reshape(-1, 3).max(axis=0)
    tgt_lo = np.array([b[0] for b in RAW_BBOX])
    tgt_hi = np.array([b[1] for b in RAW_BBOX])
    pos = (pos - lo) / (hi - lo) * (tgt_hi - tgt_lo) + tgt_lo
    # vertex normals from central differences on the closed grid
    du = np.roll(pos, -1, axis=0) - np.roll(pos, 1, axis=0)
    dv = np.roll(pos, -1, axis=1) - np.roll(pos, 1, axis=1)
    nrm = np.cross(dv, du)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    # orient outwards (away from the tube centre line)
    centre = (c - lo) / (hi - lo) * (tgt_hi - tgt_lo) + tgt_lo
    flip = np.sum(nrm * (pos - centre), axis=-1) < 0
    nrm[flip] *= -1
    positions = pos.reshape(-1, 3).astype(np.float32)
    normals = nrm.reshape(-1, 3).astype(np.float32)
    iu = np.arange(nu)[:, None]
    iv = np.arange(nv)[None, :]
    a = (iu * nv + iv)
    b = (((iu + 1) % nu) * nv + iv)
    cc = (((iu + 1) % nu) * nv + (iv + 1) % nv)
    d = (iu * nv + (iv + 1) % nv)
    tri1 = np.stack([a, b, cc], axis=-1).reshape(-1, 3)
    tri2 = np.stack([a, cc, d], axis=-1).reshape(-1, 3)
    vidx = np.empty((tri1.shape[0] * 2, 3), dtype=np.int32)
    vidx[0::2] = tri1
    vidx[1::2] = tri2
    return [{"name": "synth_dragon", "positions": positions, "normals": normals, "vidx": vidx,
             "nidx": vidx.copy()}]


def mesh_sha256(groups: list) -> str:
    h = hashlib.sha256()
    for g in groups:
        for k in ("positions", "normals", "vidx", "nidx"):
            if g[k] is not None:
                h.update(np.ascontiguousarray(g[k]).tobytes())
    return h.hexdigest()
