"""Host set-up with instanced meshes: a mesh handed to group_add_mesh several times is kept once,
divided once per threshold and flattened with shared triangle data -- and the result is what the generic path (every
triangle a shape of its own, partitioned by Group::divide as restated from groups.rs:198-299) produces."""
import numpy as np

from conftest import small_dragons
from raytracer_amd import abi, builder as B, host


def _mesh(rng, n):
    gx, gz = np.meshgrid(np.linspace(-2, 2, n), np.linspace(-2, 2, n))
    pos = np.stack([gx, 0.4 * np.sin(3 * gx) * np.cos(2 * gz) + rng.uniform(-0.05, 0.05, gx.shape), gz], axis=-1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    tris = np.concatenate([tris, tris[rng.choice(len(tris), 4)]]).astype(np.int32)  # a few triangles listed twice
    nrm = np.stack([-np.cos(3 * gx), np.ones_like(gx), np.sin(2 * gz)], -1).reshape(-1, 3)
    return pos.astype(np.float32), nrm.astype(np.float32), tris


def _relabel(a):
    seen, out = {}, np.empty(len(a), dtype=np.int64)
    for i, x in enumerate(a):
        out[i] = seen.setdefault(int(x), len(seen))
    return out


def _world(pos, nrm, tris, smooth, thresholds, generic, after=None):
    w = host.new_world(0)
    for k, th in enumerate(thresholds):
        g = w.group([("rotate-y", 0.3 * k), ("translate", 3.0 * k, 0, 0)])
        mat = B.Material(color=(1.0, 0.1 * k, 0.2))
        if generic:  # the reference's own way: one shape per triangle (obj.rs:24-36), an OBJ group under the root
            child = w.group(())
            for t in tris:
                p = pos[t]
                if smooth:
                    s = w.shape("smooth-triangle", (), mat, params=[*p[0], *p[1], *p[2], *nrm[t[0]], *nrm[t[1]], *nrm[t[2]]])
                else:
                    s = w.shape("triangle", (), mat, params=[*p[0], *p[1], *p[2]])
                w.group_add_shape(child, s)
            w.group_add_child(g, child)
        else:
            child = w.group_add_mesh(g, mat, pos, nrm if smooth else None, tris, tris if smooth else None)
        w.group_divide(g, th)
        if after:
            after(w, k, g, child)
        w.add_group(g)
    w.add_point_light((-5, 10, -5), (1, 1, 1))
    return abi.scene_arrays(w.flatten())


def test_mesh_path_flattens_like_one_shape_per_triangle():
    rng = np.random.RandomState(4)
    pos, nrm, tris = _mesh(rng, 12)
    for smooth in (False, True):
        for thresholds in ([5, 5, 5], [1, 40, 1], [10 ** 6, 3]):
            a = _world(pos, nrm, tris, smooth, thresholds, generic=False)
            b = _world(pos, nrm, tris, smooth, thresholds, generic=True)
            assert len(a["nodes"]) == len(b["nodes"]) and len(a["shapes"]) == len(b["shapes"])
            for f in ("bmin", "bmax", "xform", "flags", "first_shape", "n_shapes", "skip", "parent", "next_xf", "idrun", "n_children"):
                assert a["nodes"][f].tobytes() == b["nodes"][f].tobytes(), f
            for f in ("kind_flags", "xform", "material", "parent", "mn", "mx"):
                assert a["shapes"][f].tobytes() == b["shapes"][f].tobytes(), f
            # value classes: the same equalities (the ids themselves are first-seen numbers)
            assert np.array_equal(_relabel(a["shapes"]["vclass"]), _relabel(b["shapes"]["vclass"]))
            # triangle data: shared between instances on the mesh path, the same values shape by shape
            assert np.array_equal(a["tri_geom"][a["shapes"]["prim"]], b["tri_geom"][b["shapes"]["prim"]])
            assert np.array_equal(a["tri_normals"][a["shapes"]["prim"]], b["tri_normals"][b["shapes"]["prim"]])
            # a node's own triangles are a contiguous run from first_tri in both
            for arr in (a, b):
                n = arr["nodes"]
                tri_nodes = np.flatnonzero((n["flags"] & abi.RT_NODE_ALL_TRIS) != 0)
                for i in tri_nodes:
                    f0, k = int(n["first_shape"][i]), int(n["n_shapes"][i])
                    assert np.array_equal(arr["shapes"]["prim"][f0:f0 + k], int(n["first_tri"][i]) + np.arange(k))
            # one copy per distinct way of dividing the mesh at most (equal runs of different divisions are shared too)
            assert len(tris) <= len(a["tri_geom"]) <= len(tris) * len(set(thresholds))
            assert len(b["tri_geom"]) == len(tris) * len(thresholds)


def _same_flatten(a, b):
    assert len(a["nodes"]) == len(b["nodes"]) and len(a["shapes"]) == len(b["shapes"])
    for f in ("bmin", "bmax", "xform", "flags", "first_shape", "n_shapes", "skip", "parent", "next_xf", "idrun", "n_children"):
        assert a["nodes"][f].tobytes() == b["nodes"][f].tobytes(), f
    for f in ("kind_flags", "xform", "material", "parent", "mn", "mx"):
        assert a["shapes"][f].tobytes() == b["shapes"][f].tobytes(), f
    assert np.array_equal(_relabel(a["shapes"]["vclass"]), _relabel(b["shapes"]["vclass"]))
    tri = a["shapes"]["kind_flags"] & 0xF
    tri = (tri == B.SHAPE_KINDS["triangle"]) | (tri == B.SHAPE_KINDS["smooth-triangle"])
    assert tri.sum() >= len(a["shapes"]) - 3
    assert np.array_equal(a["tri_geom"][a["shapes"]["prim"][tri]], b["tri_geom"][b["shapes"]["prim"][tri]])


def test_copies_of_a_divide_template_that_are_changed_afterwards():
    """An untouched copy of a divide template is flattened from the template; one that gets a shape, a child, a
    transform or another divide afterwards must be walked group by group again -- and flatten like the generic path."""
    rng = np.random.RandomState(11)
    pos, nrm, tris = _mesh(rng, 10)

    def divide_again(w, k, g, child):
        if k != 1:
            w.group_divide(g, 3)

    def add_a_shape(w, k, g, child):
        if k == 0:
            w.group_add_shape(child, w.shape("sphere", [("scale", 0.2, 0.2, 0.2)], B.Material()))

    def add_a_child(w, k, g, child):
        if k == 2:
            extra = w.group([("translate", 0, 1, 0)])
            w.group_add_shape(extra, w.shape("cube", (), B.Material()))
            w.group_add_child(child, extra)

    def move_the_copy(w, k, g, child):
        if k >= 1:
            w.group_set_transform(child, [("scale", 1.0 + k, 1, 1)])

    for after in (divide_again, add_a_shape, add_a_child, move_the_copy):
        for smooth in (False, True):
            a = _world(pos, nrm, tris, smooth, [8, 8, 8], generic=False, after=after)
            b = _world(pos, nrm, tris, smooth, [8, 8, 8], generic=True, after=after)
            _same_flatten(a, b)


def _tops(w):
    acc = abi.accel_arrays(host.hip_lib(), w.flatten())
    heads = acc["root_of"][acc["root_of"] != 0xFFFFFFFF]
    return [int(acc["nodes"]["b"][h]["v"][1][2:3].copy().view("<u4")[0]) for h in heads]


def test_instances_share_a_tree_whoever_flattened_them():
    """rt_accel_build recognises instances by triangle index when the host shares one copy of the triangle data
    (this repo's flattener) and by the triangles' bytes when it repeats them (a flatten of the reference's World, which
    re-parses the OBJ file per instance): one auxiliary tree per distinct division either way."""
    rng = np.random.RandomState(5)
    pos, nrm, tris = _mesh(rng, 9)
    for generic in (False, True):
        for thresholds, n_trees in (([6, 6, 6], 1), ([6, 9, 6, 9], 2), ([4, 5, 6], 3)):
            w = host.new_world(0)
            for k, th in enumerate(thresholds):
                g = w.group([("translate", 3.0 * k, 0, 0)])
                mat = B.Material(color=(1.0, 0.1 * k, 0.2))
                if generic:
                    child = w.group(())
                    for t in tris:
                        w.group_add_shape(child, w.shape("triangle", (), mat, params=[*pos[t[0]], *pos[t[1]], *pos[t[2]]]))
                    w.group_add_child(g, child)
                else:
                    w.group_add_mesh(g, mat, pos, None, tris, None)
                w.group_divide(g, th)
                w.add_group(g)
            w.add_point_light((-5, 10, -5), (1, 1, 1))
            tops = _tops(w)
            assert len(tops) == len(thresholds) and len(set(tops)) == n_trees, (generic, thresholds, tops)
            assert tops[0] == min(tops) and all(tops[k] == tops[thresholds.index(th)] for k, th in enumerate(thresholds))


def test_value_classes_of_a_big_mesh_with_repeated_triangles():
    """Meshes of 65 536 triangles and more get their geometry classes on threads (TriValueTable::get_batch): the same
    equalities as the one-shape-per-triangle path finds, with every third triangle a repeat of an earlier one, some
    of them listed with -0 for 0, and a triangle holding a NaN (equal to nothing, not even itself)."""
    rng = np.random.RandomState(21)
    n = 66_000
    base = rng.randint(-50, 50, size=(n, 3, 3)).astype(np.float32)
    rep = rng.randint(0, n // 2, size=n // 3)
    base[n - len(rep):] = base[rep]
    base[base == 0][::2] *= -1.0
    zeros = np.flatnonzero((base == 0).any(axis=(1, 2)))[:500]
    base[zeros] = np.where(base[zeros] == 0, np.float32(-0.0), base[zeros])
    base[1234, 1, 2] = np.nan
    base[n - 7] = base[1234]
    pos = base.reshape(-1, 3)
    tris = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    mat = B.Material(color=(0.3, 0.4, 0.5))

    def classes(generic):
        w = host.new_world(0)
        g = w.group(())
        if generic:
            child = w.group(())
            for t in tris:
                w.group_add_shape(child, w.shape("triangle", (), mat, params=[*pos[t[0]], *pos[t[1]], *pos[t[2]]]))
            w.group_add_child(g, child)
        else:
            w.group_add_mesh(g, mat, pos, None, tris, None)
        w.add_group(g)
        w.add_point_light((-5, 10, -5), (1, 1, 1))
        return abi.scene_arrays(w.flatten())["shapes"]["vclass"].copy()

    a, b = classes(False), classes(True)
    never = 0xFFFFFFFF
    assert a[1234] == never and a[n - 7] == never and np.array_equal(a == never, b == never)
    assert np.array_equal(_relabel(a), _relabel(b))
    assert len(set(a.tolist())) < n - len(rep) + 2


def test_six_dragons_hold_one_copy_of_the_triangles():
    scene = small_dragons(24, 16)
    fs = scene.build(host.new_world(0)).flatten()
    arr = abi.scene_arrays(fs)
    n_mesh = 2 * 24 * 16
    assert fs.n_tris == n_mesh and len(arr["tri_geom"]) == n_mesh
    tri_shapes = arr["shapes"][(arr["shapes"]["kind_flags"] & abi.RT_SHAPE_KIND_MASK) >= 5]
    assert len(tri_shapes) == 6 * n_mesh
    assert np.array_equal(np.sort(tri_shapes["prim"].reshape(6, n_mesh), axis=1), np.tile(np.arange(n_mesh), (6, 1)))
    # six different materials: no value class is shared between instances, each instance has its mesh's classes
    vc = tri_shapes["vclass"].reshape(6, n_mesh)
    assert len(np.unique(vc)) == 6 * len(np.unique(vc[0]))
    # and the library recognises the six subtrees as one tree (shared auxiliary BVH)
    acc = abi.accel_arrays(host.hip_lib(), fs)
    roots = acc["root_of"][acc["root_of"] != 0xFFFFFFFF]
    tops = {int(acc["nodes"][r]["b"]["v"][1][2:3].view(np.uint32)[0]) for r in roots}  # entry record: b[1] = {lmax, amax, top, base}
    assert len(roots) == 6 and len(tops) == 1


def test_a_mesh_group_that_gets_another_shape_holds_shapes():
    """Adding a shape to the group group_add_mesh returned turns its triangles into ordinary shapes first (insertion
    order kept: the mesh came first), so nothing depends on the fast path."""
    rng = np.random.RandomState(2)
    pos, nrm, tris = _mesh(rng, 6)
    out = []
    for generic in (False, True):
        w = host.new_world(0)
        g = w.group([("scale", 2, 2, 2)])
        mat = B.Material(color=(0.3, 0.6, 0.9))
        if generic:
            child = w.group(())
            for t in tris:
                p = pos[t]
                w.group_add_shape(child, w.shape("triangle", (), mat, params=[*p[0], *p[1], *p[2]]))
            w.group_add_child(g, child)
        else:
            child = w.group_add_mesh(g, mat, pos, None, tris, None)
        w.group_add_shape(child, w.shape("sphere", [("translate", 0, 1, 0)]))
        w.group_divide(g, 4)
        w.add_group(g)
        out.append(abi.scene_arrays(w.flatten()))
    a, b = out
    assert a["nodes"].tobytes() == b["nodes"].tobytes()
    for f in ("kind_flags", "xform", "material", "parent", "prim"):
        assert a["shapes"][f].tobytes() == b["shapes"][f].tobytes(), f
    assert np.array_equal(_relabel(a["shapes"]["vclass"]), _relabel(b["shapes"]["vclass"]))
    assert a["tri_geom"].tobytes() == b["tri_geom"].tobytes()
