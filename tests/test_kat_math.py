"""Reference unit tests for the host-side math restated against the oracle: src/matrices.rs:243-388
(4x4 cases; the 2x2/3x3 minors are covered through the 4x4 inverse), src/transformations.rs:142-263,
src/rays.rs:33-54, src/tuples.rs reflect."""
import math

import numpy as np

import orc_probe as P
from oracle import oracle
from test_kat_shapes import approx

f32 = np.float32
PI = float(f32(math.pi))
S2 = float(f32(np.sqrt(f32(2))) / f32(2))
M1 = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 8, 7, 6], [5, 4, 3, 2]]
M2 = [[-2, 1, 2, 3], [3, 2, 1, -1], [4, 3, 6, 5], [1, 2, 7, 8]]
I4 = np.eye(4, dtype=f32)


def pt(m, p):
    return P.mat_apply(m, (*p, 1))[:3]


def vec(m, v):
    return P.mat_apply(m, (*v, 0))[:3]


def test_matrix_basics():  # matrices.rs:243-308
    m3 = [[20, 22, 50, 48], [44, 54, 114, 108], [40, 58, 110, 102], [16, 26, 46, 42]]
    assert np.array_equal(P.matmul(M1, M2), np.array(m3, dtype=f32))
    a = [[1, 2, 3, 4], [2, 4, 4, 2], [8, 6, 4, 1], [0, 0, 0, 1]]
    assert np.array_equal(P.mat_apply(a, (1, 2, 3, 1)), (18, 24, 33, 1))
    assert np.array_equal(P.matmul(M1, I4), np.array(M1, dtype=f32))
    assert np.array_equal(P.matmul(I4, M1), np.array(M1, dtype=f32))
    assert np.array_equal(P.mat_apply(I4, (1, 2, 3, 1)), (1, 2, 3, 1))


def test_matrix_inverse():  # matrices.rs:346-387
    m1 = [[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]]
    want = [[0.21805, 0.45113, 0.24060, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
            [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]]
    inv = P.matrix_inverse(m1)
    assert inv[3][2] == f32(-160.0) / f32(532.0) and inv[2][3] == f32(105.0) / f32(532.0)  # cofactor / determinant
    assert approx(inv, want)
    m3 = P.matmul(M1, M2)
    assert approx(P.matmul(m3, P.matrix_inverse(M2)), M1)
    assert P.matrix_inverse([[1, 2, 3, 4], [2, 4, 6, 8], [0, 1, 0, 0], [0, 0, 1, 0]]) is None  # singular: inverse() is None


def test_translation_scaling():  # transformations.rs:142-166
    t = P.transform([("translate", 5, -3, 2)])
    assert np.array_equal(pt(t, (-3, 4, 5)), (2, 1, 7))
    assert np.array_equal(pt(P.matrix_inverse(t), (-3, 4, 5)), (-8, 7, 3))
    assert np.array_equal(vec(t, (-3, 4, 5)), (-3, 4, 5))
    s = P.transform([("scale", 2, 3, 4)])
    assert np.array_equal(pt(s, (-4, 6, 8)), (-8, 18, 32)) and np.array_equal(vec(s, (-4, 6, 8)), (-8, 18, 32))
    assert np.array_equal(vec(P.matrix_inverse(s), (-4, 6, 8)), (-2, 2, 2))
    assert np.array_equal(pt(P.transform([("scale", -1, 1, 1)]), (2, 3, 4)), (-2, 3, 4))


def test_rotations():  # transformations.rs:168-195
    q, h = float(f32(PI) / f32(4)), float(f32(PI) / f32(2))
    assert approx(pt(P.transform([("rotate-x", q)]), (0, 1, 0)), (0, S2, S2))
    assert approx(pt(P.transform([("rotate-x", h)]), (0, 1, 0)), (0, 0, 1))
    assert approx(pt(P.matrix_inverse(P.transform([("rotate-x", q)])), (0, 1, 0)), (0, S2, -S2))
    assert approx(pt(P.transform([("rotate-y", q)]), (0, 0, 1)), (S2, 0, S2))
    assert approx(pt(P.transform([("rotate-y", h)]), (0, 0, 1)), (1, 0, 0))
    assert approx(pt(P.transform([("rotate-z", q)]), (0, 1, 0)), (-S2, S2, 0))
    assert approx(pt(P.transform([("rotate-z", h)]), (0, 1, 0)), (-1, 0, 0))


def test_shearing():  # transformations.rs:197-215
    for k, want in ((0, (5, 3, 4)), (1, (6, 3, 4)), (2, (2, 5, 4)), (3, (2, 7, 4)), (4, (2, 3, 6)), (5, (2, 3, 7))):
        a = [0.0] * 6
        a[k] = 1.0
        assert np.array_equal(pt(P.transform([("shear", *a)]), (2, 3, 4)), want)


def test_chaining():  # transformations.rs:217-235: Transform applies its ops in call order
    h = float(f32(PI) / f32(2))
    p2 = pt(P.transform([("rotate-x", h)]), (1, 0, 1))
    p3 = pt(P.transform([("scale", 5, 5, 5)]), p2)
    p4 = pt(P.transform([("translate", 10, 5, 7)]), p3)
    assert approx(p2, (1, -1, 0)) and approx(p3, (5, -5, 0)) and approx(p4, (15, 0, 7))
    t = P.transform([("rotate-x", h), ("scale", 5, 5, 5), ("translate", 10, 5, 7)])
    assert approx(pt(t, (1, 0, 1)), (15, 0, 7))


def test_view_transformations():  # transformations.rs:237-262
    assert np.array_equal(P.view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)), I4)
    assert np.array_equal(P.view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)), P.transform([("scale", -1, 1, -1)]))
    assert np.array_equal(P.view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)), P.transform([("translate", 0, 0, -8)]))
    want = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
            [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]]
    assert approx(P.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0)), want)


def test_ray_transform():  # rays.rs:42-54 through Shape::intersect's object-space ray (shapes.rs:204-206)
    m = P.matrix_inverse(P.transform([("translate", -3, -4, -5)]))  # = translation(3,4,5)
    assert np.array_equal(pt(m, (1, 2, 3)), (4, 6, 8)) and np.array_equal(vec(m, (0, 1, 0)), (0, 1, 0))
    m = P.transform([("scale", 2, 3, 4)])
    assert np.array_equal(pt(m, (1, 2, 3)), (2, 6, 12)) and np.array_equal(vec(m, (0, 1, 0)), (0, 3, 0))


def test_ray_position():  # rays.rs:33-40: hit point of a plane facing the ray = position(t)
    w = oracle.new_world()
    w.add_point_light((0, 10, 0), (1, 1, 1))
    pl = w.shape("plane", [("rotate-z", float(f32(PI) / f32(2))), ("translate", 4.5, 0, 0)])
    c = P.comps_empty(w, (2, 3, 4), (1, 0, 0), 2.5, (P.DET, pl, 0))
    assert np.array_equal(c.point, (4.5, 3, 4))


def test_reflect():  # tuples.rs (reflect KATs)
    assert np.array_equal(P.reflect((1, -1, 0), (0, 1, 0)), (1, 1, 0))
    assert approx(P.reflect((0, -1, 0), (S2, S2, 0)), (1, 0, 0))
