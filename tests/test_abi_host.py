"""CPU-side checks of the boundary: the C-ABI libraries load without a GPU and export every symbol
the headers declare, the ctypes/numpy mirrors match the C struct sizes, the flattened scene obeys
the layout rules of include/rt_abi.h, and the library-private auxiliary BVHs are well formed.
No compute call is made on the GPU library here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path, small_dragons
from oracle import oracle
from raytracer_amd import abi, builder as B, host
from raytracer_amd.scene import Scene


def _declared(header, prefix_macro=None):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    if prefix_macro:
        return sorted(set(re.findall(r"RTB_PREFIX\((\w+)\)\(", text)))
    return sorted(set(re.findall(r"\b(rt_\w+)\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = host.hip_lib()
    names = _declared("rt_abi.h")
    assert {"rt_render", "rt_render_device", "rt_scene_create", "rt_color_at", "rt_context_create"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"librt_hip.so does not export {n}"
    assert sorted(abi.EXPORTS) == names, "abi.EXPORTS is out of date with include/rt_abi.h"
    assert lib.rt_abi_version() == abi.RT_ABI_VERSION


def test_builder_libraries_export_every_declared_symbol():
    names = _declared("rt_builder.h", prefix_macro=True)
    assert len(names) >= 20
    hl, ol = host.host_lib(), oracle.lib()
    for n in names:
        assert hasattr(hl, "rth_" + n), f"librt_host.so does not export rth_{n}"
        assert hasattr(ol, "orc_" + n), f"librt_oracle.so does not export orc_{n}"


def test_struct_sizes_match_the_headers(tmp_path):
    structs = {"RtXform": abi.XFORM.itemsize, "RtShape": abi.SHAPE.itemsize, "RtTriGeom": abi.TRI_GEOM.itemsize,
               "RtTriNormals": abi.TRI_NORMALS.itemsize, "RtNode": abi.NODE.itemsize, "RtCsgNode": abi.CSG_NODE.itemsize,
               "RtCsgRoot": abi.CSG_ROOT.itemsize, "RtMaterial": abi.MATERIAL.itemsize, "RtPattern": abi.PATTERN.itemsize,
               "RtUvPattern": abi.UVPATTERN.itemsize, "RtImage": abi.IMAGE.itemsize, "RtLight": abi.LIGHT.itemsize,
               "RtScene": C.sizeof(abi.RtScene), "RtCamera": C.sizeof(abi.RtCamera),
               "RtRenderOptions": C.sizeof(abi.RtRenderOptions), "RtAux": C.sizeof(abi.RtAux),
               "RtStats": C.sizeof(abi.RtStats), "RtAccelView": C.sizeof(abi.RtAccelView),
               "RtbXfOp": C.sizeof(B.XfOp), "RtbMaterial": C.sizeof(B.MaterialC), "RtbCamera": C.sizeof(B.CameraC),
               "RtbRenderOptions": C.sizeof(B.RenderOptionsC), "RtbAux": C.sizeof(B.AuxC), "RtbStats": C.sizeof(B.StatsC)}
    src = tmp_path / "sz.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in structs)
    src.write_text(f'#include <stdio.h>\n#include "rt_abi.h"\n#include "rt_builder.h"\nint main(void) {{\n{body}\n  return 0; }}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, size in structs.items():
        assert int(out[n]) == size, f"{n}: C says {out[n]}, Python mirror says {size}"


def _flat(name, obj_files=()):
    scene = small_dragons() if name == "small_dragons" else Scene.from_file(scene_path(name), obj_files=obj_files)
    hw = scene.build(host.new_world(0))
    return scene, hw, hw.flatten()


def test_flatten_layout_rules(obj_files):
    scene, hw, fs = _flat("small_dragons")
    a = abi.scene_arrays(fs)
    nodes, shapes = a["nodes"], a["shapes"]
    assert len(nodes) > 100 and len(shapes) > 200
    # depth-first order with skip pointers; parents precede children
    assert np.all(nodes["skip"] > np.arange(len(nodes))) and np.all(nodes["skip"] <= len(nodes))
    assert np.all(nodes["parent"] < np.arange(len(nodes)))
    # a group's own shapes are contiguous, in insertion order, and point back at it
    for i in np.flatnonzero(nodes["n_shapes"])[:200]:
        f, n = nodes["first_shape"][i], nodes["n_shapes"][i]
        assert np.all(shapes["parent"][f:f + n] == i)
        if nodes["flags"][i] & abi.RT_NODE_ALL_TRIS:
            assert np.array_equal(shapes["prim"][f:f + n], nodes["first_tri"][i] + np.arange(n))
    # loose shapes come last (src/world.rs:56-59)
    nl = a["n_loose_shapes"]
    assert np.all(shapes["parent"][len(shapes) - nl:] == -1)
    # xforms[0] is the identity
    assert np.array_equal(a["xforms"]["m"][0], np.eye(4, dtype=np.float32)[:3].reshape(-1))


def test_flatten_matches_oracle_ranks(obj_files):
    """The device object id of a primary hit is the shape's index in the flattened scene; the oracle
    numbers shapes independently (its own insertion rank).  Same ids on the same rays <=> same order."""
    scene = Scene.from_file(scene_path("cover"))
    cam = scene.camera(64, 64)
    ow = scene.build(oracle.new_world())
    r = ow.render(cam, aux=True)
    a = abi.scene_arrays(scene.build(host.new_world(0)).flatten())
    ids = np.unique(r.object_id[r.object_id >= 0])
    assert ids.max() < len(a["shapes"])
    # cover.yaml: 1 plane + 1 sphere + 17 cubes = 19 loose shapes, all hit by some ray at 64x64? at least kinds agree
    kinds = a["shapes"]["kind_flags"] & abi.RT_SHAPE_KIND_MASK
    assert sorted(np.bincount(kinds, minlength=7).tolist(), reverse=True)[:3] == [17, 1, 1]
    assert a["n_loose_shapes"] == 19 and len(a["lights"]) == 1 and a["recursion_limit"] == 4


def test_group_boxes_equal_the_oracle(obj_files):
    """Copied reference boxes (Q6): the flattened top-level group box equals the oracle's cached bounds."""
    import orc_probe as P
    scene = Scene.from_file(scene_path("space_teapot"), obj_files=obj_files)
    a = abi.scene_arrays(scene.build(host.new_world(0)).flatten())
    ow = scene.build(oracle.new_world())
    tops = np.flatnonzero(a["nodes"]["parent"] == -1)
    assert len(tops) >= 1
    # the oracle creates groups in the same order as the host; its top-level groups are those without parent
    otops = [g for g in range(P.group_count(ow.w)) if P.group_parent(ow.w, g) < 0 and sum(P.group_counts(ow, g)) > 0]
    boxes = {tuple(np.concatenate(P.group_bounds(ow, g)).tolist()) for g in otops}
    for i in tops:
        key = tuple(np.concatenate([a["nodes"]["bmin"][i], a["nodes"]["bmax"][i]]).tolist())
        assert key in boxes


def test_aux_bvh_is_well_formed(obj_files):
    """rt_accel.h: one BVH per maximal subtree of identity-transform triangle groups; every triangle of
    the subtree sits in exactly one leaf with its shape index and owning group; boxes nest and hold
    their triangles; gate chains only skip groups whose box test is implied by nesting."""
    scene, hw, fs = _flat("small_dragons")
    a = abi.scene_arrays(fs)
    acc = abi.accel_arrays(host.hip_lib(), fs)
    has = _check_aux(a, acc)
    assert len(has) == 6  # dragons.yaml: six dragon.obj instances, each one covered subtree


def test_oversized_subtrees_are_covered_piecewise(monkeypatch):
    """A subtree with more triangles than one auxiliary tree can hold (3.1 M; RT_AUX_MAX_TRIS lowers the limit for
    this test) is split: its child groups become roots of their own, its own shapes keep the reference's tests."""
    scene, hw, fs = _flat("small_dragons")
    a = abi.scene_arrays(fs)
    whole = abi.accel_arrays(host.hip_lib(), fs)
    n_whole = len(whole["tris"])
    monkeypatch.setenv("RT_AUX_MAX_TRIS", "600")
    acc = abi.accel_arrays(host.hip_lib(), fs)
    has = _check_aux(a, acc, one_tree=False)
    nodes = a["nodes"]
    assert len(has) > 6
    below = [int(nodes["n_shapes"][int(R):int(nodes["skip"][R])].sum()) for R in has]
    assert max(below) <= 600
    assert 0 < len(acc["tris"]) < n_whole  # (shared between the six instances) the split groups' own triangles are left out


def _check_aux(a, acc, one_tree=True):
    nodes, shapes, roots, gate = a["nodes"], a["shapes"], acc["root_of"], acc["gate_next"]
    NONE = 0xFFFFFFFF
    has = np.flatnonzero(roots != NONE)
    an, at = acc["nodes"], acc["tris"]
    covered = np.zeros(len(shapes), dtype=int)
    tops = set()
    for R in has:
        R = int(R)
        end = int(nodes["skip"][R])
        sub = np.arange(R, end)
        assert np.all(nodes["xform"][R + 1:end] == 0), "only the subtree root may carry a transform"
        n_tris = int(nodes["n_shapes"][sub].sum())
        hdr = an[roots[R]]["b"]  # entry record of this instance: {centre, radius}{lmax, amax, top node, first shape}
        centre, radius = hdr["v"][0], hdr["w"][0:1].view(np.float32)[0]
        top, shape_base = int(hdr["v"][1][2:3].view(np.uint32)[0]), int(hdr["w"][1])
        tops.add(top)
        stack = [(top, 0, np.full(3, -np.inf), np.full(3, np.inf))]
        depth_max, seen = 0, 0
        while stack:
            idx, depth, lo, hi = stack.pop()
            depth_max = max(depth_max, depth)
            nd = an[idx]["b"]  # a 4-wide node: 2 to 4 children, the missing ones with an empty box
            present = [c for c in range(4) if int(nd["w"][c]) != NONE]
            assert present == list(range(len(present))) and len(present) >= 2, "children fill the node from the front"
            for c in range(4):
                mn, mx = nd["v"][2 * c], nd["v"][2 * c + 1]
                word = int(nd["w"][c])
                if word == NONE:
                    assert np.all(mn == np.inf) and np.all(mx == -np.inf)
                    continue
                ref, cnt = word >> 4, word & 15
                assert np.all(mn >= lo) and np.all(mx <= hi), "child box sticks out of its parent"
                if cnt == 0:
                    stack.append((ref, depth + 1, mn, mx))
                    continue
                assert 1 <= cnt <= 15
                for t in at[ref:ref + cnt]:
                    si, owner = shape_base + int(t["shape"]), R + int(t["owner"])  # leaves are instance-relative
                    covered[si] += 1
                    seen += 1
                    assert R <= owner < end and shapes["parent"][si] == owner
                    g = a["tri_geom"][shapes["prim"][si]]
                    assert np.array_equal(t["p1"], g["p1"]) and np.array_equal(t["e1"], g["e1"]) and \
                        np.array_equal(t["e2"], g["e2"])
                    for v in (g["p1"], g["p1"] + g["e1"], g["p1"] + g["e2"]):
                        assert np.all(v >= mn) and np.all(v <= mx), "leaf box does not hold its triangle"
                        assert np.linalg.norm(v - centre) <= radius
        assert seen == n_tris and depth_max <= 10  # three stack entries a level: RT_AUX_STACK 32
        # gate chains: walking gate_next from any group reaches the root through ancestors only, and every
        # ancestor that is skipped contains the box below it
        for g0 in sub[1::7]:
            g, below = int(g0), int(g0)
            required = set()
            while gate[g] != NONE:
                g = int(gate[g])
                required.add(g)
            p = int(nodes["parent"][below])
            while p != R:
                inside = np.all(nodes["bmin"][below] >= nodes["bmin"][p]) and np.all(nodes["bmax"][below] <= nodes["bmax"][p])
                assert inside or p in required
                below, p = p, int(nodes["parent"][p])
    in_sub = np.zeros(len(shapes), dtype=bool)
    for R in has:
        for i in range(int(R), int(nodes["skip"][R])):
            f, n = int(nodes["first_shape"][i]), int(nodes["n_shapes"][i])
            in_sub[f:f + n] = True
    assert np.array_equal(covered, in_sub.astype(int)), "every triangle of a covered subtree sits in exactly one leaf"
    if one_tree:
        assert len(tops) == 1, "the six instances of dragon.obj share one tree"
    return has


def test_flatten_matches_committed_hashes(obj_files):
    """tests/golden/flatten_hashes.json (tools/make_flatten_fixtures.py): the flattened RtScene of every config scene
    is byte-for-byte what it was when the GPU path was validated against the oracle and the reference renders."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_flatten_fixtures import NAMES, flatten_hashes
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "flatten_hashes.json")))
    assert sorted(want) == sorted(NAMES)
    for name in NAMES:
        got = flatten_hashes(name)
        for k, v in want[name].items():
            assert got[k] == v, f"{name}: flattened `{k}` changed"


def test_scene_without_triangle_groups_has_no_aux():
    scene, hw, fs = _flat("cover")
    acc = abi.accel_arrays(host.hip_lib(), fs)
    assert len(acc["nodes"]) == 0 and len(acc["tris"]) == 0


def test_render_without_gpu_fails_loudly():
    """There is no CPU fallback in the product: without a GPU the render call returns an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    scene = Scene.from_file(scene_path("basic_scene"))
    with pytest.raises(RuntimeError):
        scene.build(host.new_world(0)).render(scene.camera(8, 8))
