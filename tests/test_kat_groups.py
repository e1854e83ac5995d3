"""Reference unit tests for groups and bounding boxes restated against the oracle:
src/groups.rs:376-512 and src/bounds.rs:165-310."""
import math

import numpy as np
import pytest

import orc_probe as P
from oracle import oracle
from test_kat_shapes import approx, norm

f32 = np.float32
DET, GROUP = P.DET, P.GROUP
PI = float(f32(math.pi))


@pytest.fixture()
def w():
    return oracle.new_world()


def test_group_intersections(w):  # groups.rs:376-405
    g = w.group()
    assert len(P.group_intersect(w, g, (0, 0, 0), (0, 0, 1))) == 0

    g = w.group()
    s1 = w.shape("sphere")
    s2 = w.shape("sphere", [("translate", 0, 0, -3)])
    s3 = w.shape("sphere", [("translate", 5, 0, 0)])
    for s in (s1, s2, s3):
        w.group_add_shape(g, s)
    ts = P.group_intersect(w, g, (0, 0, -5), (0, 0, 1))
    assert list(ts) == [1, 3, 4, 6]  # s2, s2, s1, s1 in sorted order

    g = w.group([("scale", 2, 2, 2)])
    w.group_add_shape(g, w.shape("sphere", [("translate", 5, 0, 0)]))
    assert len(P.group_intersect(w, g, (10, 0, -10), (0, 0, 1))) == 2


def _nested(w, inner_scale):
    g1 = w.group([("rotate-y", float(f32(PI) / f32(2)))])
    g2 = w.group([("scale", *inner_scale)])
    w.group_add_shape(g2, w.shape("sphere", [("translate", 5, 0, 0)]))
    w.group_add_child(g1, g2)
    return g2


def test_group_normals(w):  # groups.rs:408-441
    g2 = _nested(w, (2, 2, 2))
    assert approx(P.world_to_object(w, (GROUP, g2, 0), (-2, 0, -10)), (0, 0, -1))
    g2 = _nested(w, (1, 2, 3))
    r3 = float(f32(np.sqrt(f32(3))) / f32(3))
    assert approx(P.normal_to_world(w, (GROUP, g2, 0), (r3, r3, r3)), (0.2857, 0.4286, -0.8571))
    g2 = _nested(w, (1, 2, 3))
    assert approx(P.normal_at(w, (GROUP, g2, 0), (1.7321, 1.1547, -5.5774)), (0.2857, 0.4286, -0.8571))


def test_group_bounding_box(w):  # groups.rs:444-464
    g = w.group()
    w.group_add_shape(g, w.shape("sphere", [("scale", 2, 2, 2), ("translate", 2, 5, -3)]))
    w.group_add_shape(g, w.shape("cylinder", [("scale", 0.5, 1, 0.5), ("translate", -4, -1, 4)], params=[-2, 2, 0]))
    mn, mx = P.group_bounds(w, g)
    assert np.array_equal(mn, (-4.5, -3, -5)) and np.array_equal(mx, (4, 7, 4.5))


def test_group_partitions(w):  # groups.rs:467-512
    s1 = w.shape("sphere", [("translate", -2, 0, 0)])
    s2 = w.shape("sphere", [("translate", 2, 0, 0)])
    s3 = w.shape("sphere")
    g = w.group()
    for s in (s1, s2, s3):
        w.group_add_shape(g, s)
    left, right = P.partition(w, g)
    assert P.group_shape_handle(w.w, g, 0) == s3
    assert P.group_shape_handle(w.w, left, 0) == s1
    assert P.group_shape_handle(w.w, right, 0) == s2

    w = oracle.new_world()
    s1 = w.shape("sphere", [("translate", -2, -2, 0)])
    s2 = w.shape("sphere", [("translate", -2, 2, 0)])
    s3 = w.shape("sphere", [("scale", 4, 4, 4)])
    g = w.group()
    for s in (s1, s2, s3):
        w.group_add_shape(g, s)
    w.group_divide(g, 1)
    assert P.group_shape_handle(w.w, g, 0) == s3
    sub = P.group_child(w.w, g, 0)
    assert sum(P.group_counts(w, sub)) == 2
    assert P.group_shape_handle(w.w, P.group_child(w.w, sub, 0), 0) == s1
    assert P.group_shape_handle(w.w, P.group_child(w.w, sub, 1), 0) == s2

    w = oracle.new_world()
    s1 = w.shape("sphere", [("translate", -2, 0, 0)])
    s2 = w.shape("sphere", [("translate", 2, 1, 0)])
    s3 = w.shape("sphere", [("translate", 2, -1, 0)])
    sub = w.group()
    for s in (s1, s2, s3):
        w.group_add_shape(sub, s)
    s4 = w.shape("sphere")
    g = w.group()
    w.group_add_shape(g, s4)
    w.group_add_child(g, sub)
    w.group_divide(g, 3)
    assert P.group_child(w.w, g, 0) == sub
    assert P.group_shape_handle(w.w, g, 0) == s4
    assert sum(P.group_counts(w, sub)) == 2
    c0, c1 = P.group_child(w.w, sub, 0), P.group_child(w.w, sub, 1)
    assert P.group_shape_handle(w.w, c0, 0) == s1
    assert P.group_shape_handle(w.w, c1, 0) == s2 and P.group_shape_handle(w.w, c1, 1) == s3


# ---- bounds.rs -----------------------------------------------------------------------------------
def test_bounding_box_transform():  # bounds.rs:214-219
    mn, mx = P.box_transform((-1, -1, -1), (1, 1, 1), [("rotate-y", float(f32(PI) / f32(4))),
                                                       ("rotate-x", float(f32(PI) / f32(4)))])
    assert approx(mn, (-1.4142, -1.7071, -1.7071)) and approx(mx, (1.4142, 1.7071, 1.7071))


UNIT = ((-1, -1, -1), (1, 1, 1))
BOX2 = ((5, -2, 0), (11, 4, 7))


@pytest.mark.parametrize("box,o,d,res", [
    (UNIT, (5, 0.5, 0), (-1, 0, 0), True), (UNIT, (-5, 0.5, 0), (1, 0, 0), True), (UNIT, (0.5, 5, 0), (0, -1, 0), True),
    (UNIT, (0.5, -5, 0), (0, 1, 0), True), (UNIT, (0.5, 0, 5), (0, 0, -1), True), (UNIT, (0.5, 0, -5), (0, 0, 1), True),
    (UNIT, (0, 0.5, 0), (0, 0, 1), True), (UNIT, (-2, 0, 0), (2, 4, 6), False), (UNIT, (0, -2, 0), (6, 2, 4), False),
    (UNIT, (0, 0, -2), (4, 6, 2), False), (UNIT, (2, 0, 2), (0, 0, -1), False), (UNIT, (0, 2, 2), (0, -1, 0), False),
    (UNIT, (2, 2, 0), (-1, 0, 0), False),
    (BOX2, (15, 1, 2), (-1, 0, 0), True), (BOX2, (-5, -1, 4), (1, 0, 0), True), (BOX2, (7, 6, 5), (0, -1, 0), True),
    (BOX2, (9, -5, 6), (0, 1, 0), True), (BOX2, (8, 2, 1.2), (0, 0, -1), True), (BOX2, (6, 0, -5), (0, 0, 1), True),
    (BOX2, (8, 1, 3.5), (0, 0, 1), True), (BOX2, (9, -1, -8), (2, 4, 6), False), (BOX2, (8, 3, -4), (6, 2, 4), False),
    (BOX2, (9, -1, -2), (4, 6, 2), False), (BOX2, (4, 0, 9), (0, 0, -1), False), (BOX2, (8, 6, -1), (0, -1, 0), False),
    (BOX2, (12, 5, 4), (-1, 0, 0), False)])
def test_bounding_box_intersect(box, o, d, res):  # bounds.rs:223-271 (26 ray cases)
    assert P.box_intersects(box[0], box[1], o, norm(d)) == res


@pytest.mark.parametrize("mn,mx,lmax,rmin", [
    ((-1, -4, -5), (9, 6, 5), (4, 6, 5), (4, -4, -5)), ((-1, -2, -3), (9, 5.5, 3), (4, 5.5, 3), (4, -2, -3)),
    ((-1, -2, -3), (5, 8, 3), (5, 3, 3), (-1, 3, -3)), ((-1, -2, -3), (5, 3, 7), (5, 3, 2), (-1, -2, 2))])
def test_bounding_box_splits(mn, mx, lmax, rmin):  # bounds.rs:274-310
    (lmn, lmx), (rmn, rmx) = P.box_split(mn, mx)
    assert np.array_equal(lmn, mn) and np.array_equal(lmx, lmax)
    assert np.array_equal(rmn, rmin) and np.array_equal(rmx, mx)
