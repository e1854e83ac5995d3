"""Wavefront OBJ reader for the harness (restates src/obj.rs:8-41 + the `wavefront` 0.2.3
crate's documented behaviour; the crate's source is not in the reference tree).

* ``v x y z`` / ``vn x y z`` build the position / normal tables (1-based, negative = relative);
* ``f`` polygons are fan-triangulated around their first vertex (pinned by obj.rs:82-123);
* ``g name`` starts (or re-opens) a group; faces before any ``g`` go to a default group;
  ``o``, ``s``, ``usemtl``, ``mtllib``, ``vt`` are ignored for grouping.  Groups are returned
  in order of first appearance — the reference's own test does not pin the order
  (obj.rs:142-158 uses ``contains``; SURVEY.md Q9: "parity unpinned" for inter-group tie order);
* a triangle is smooth only if all three corners carry a normal index (obj.rs:24-33).

Returns a list of groups, each a dict of numpy arrays:
``positions`` (P,3) f32, ``normals`` (N,3) f32 or None, ``vidx`` (T,3) i32, ``nidx`` (T,3) i32 (-1 = none).
Groups share the file-global position/normal tables.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def parse_obj_text(text: str) -> List[dict]:
    positions: List[List[float]] = []
    normals: List[List[float]] = []
    groups: Dict[str, dict] = {}
    order: List[str] = []
    current = ""

    def group(name):
        if name not in groups:
            groups[name] = {"v": [], "n": []}
            order.append(name)
        return groups[name]

    for line in text.splitlines():
        parts = line.split()
        if not parts or parts[0].startswith("#"):
            continue
        tag = parts[0]
        if tag == "v":
            positions.append([float(parts[1]), float(parts[2]), float(parts[3])])
        elif tag == "vn":
            normals.append([float(parts[1]), float(parts[2]), float(parts[3])])
        elif tag == "g":
            current = parts[1] if len(parts) > 1 else ""
        elif tag == "f":
            corners = []
            for tok in parts[1:]:
                fields = tok.split("/")
                vi = int(fields[0])
                vi = vi - 1 if vi > 0 else len(positions) + vi
                ni = -1
                if len(fields) >= 3 and fields[2] != "":
                    ni = int(fields[2])
                    ni = ni - 1 if ni > 0 else len(normals) + ni
                corners.append((vi, ni))
            g = group(current)
            for k in range(1, len(corners) - 1):
                tri = (corners[0], corners[k], corners[k + 1])
                g["v"].append([c[0] for c in tri])
                g["n"].append([c[1] for c in tri])
    pos = np.asarray(positions, dtype=np.float64).astype(np.float32).reshape(-1, 3)
    nrm = np.asarray(normals, dtype=np.float64).astype(np.float32).reshape(-1, 3) if normals else None
    out = []
    for name in order:
        g = groups[name]
        out.append({
            "name": name,
            "positions": pos,
            "normals": nrm,
            "vidx": np.asarray(g["v"], dtype=np.int32).reshape(-1, 3),
            "nidx": np.asarray(g["n"], dtype=np.int32).reshape(-1, 3),
        })
    return out


def parse_obj(path: str) -> List[dict]:
    with open(path, "r", errors="replace") as f:
        return parse_obj_text(f.read())


def write_obj(path: str, positions: np.ndarray, normals: np.ndarray, vidx: np.ndarray, nidx: np.ndarray) -> None:
    """Write a mesh as OBJ (used to export the synthetic dragon stand-in for inspection)."""
    with open(path, "w") as f:
        for p in positions:
            f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for n in normals:
            f.write(f"vn {n[0]:.9g} {n[1]:.9g} {n[2]:.9g}\n")
        for tv, tn in zip(vidx, nidx):
            f.write("f " + " ".join(f"{a + 1}//{b + 1}" for a, b in zip(tv, tn)) + "\n")
