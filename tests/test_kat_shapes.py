"""The reference's shape unit tests restated against the oracle: every
local_intersect / local_normal_at / constructor / bounds known answer of src/shapes/*.rs and
src/shapes.rs:330-339.  Literals are the reference's; `==` stays exact, EPSILON compares stay 1e-4."""
import math

import numpy as np
import pytest

import orc_probe as P
from oracle import oracle
from raytracer_amd import builder as B

EPS = 1e-4
DET = P.DET
f32 = np.float32
INF = math.inf


def norm(v):
    """Vector::normalize in f32 (src/tuples.rs:236-250): v / sqrt(x*x + y*y + z*z + w*w), w = 0."""
    x, y, z = (f32(c) for c in v)
    m = f32(np.sqrt(f32(f32(f32(x * x) + f32(y * y)) + f32(z * z)) + f32(0)))
    return (float(x / m), float(y / m), float(z / m))


def approx(a, b):
    return np.all(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) < EPS)


@pytest.fixture()
def w():
    return oracle.new_world()


# ---- sphere.rs:124-202 ---------------------------------------------------------------------
@pytest.mark.parametrize("o,ts", [((0, 0, -5), [4, 6]), ((0, 1, -5), [5, 5]), ((0, 2, -5), []), ((0, 0, 0), [-1, 1]),
                                  ((0, 0, 5), [-6, -4])])
def test_sphere_ray_intersect(w, o, ts):
    s = w.shape("sphere")
    got, _ = P.shape_intersect(w, (DET, s, 0), o, (0, 0, 1))
    assert list(got) == ts


def test_sphere_transformed_intersect(w):
    s = w.shape("sphere", [("scale", 2, 2, 2)])
    assert list(P.shape_intersect(w, (DET, s, 0), (0, 0, -5), (0, 0, 1))[0]) == [3, 7]
    s = w.shape("sphere", [("translate", 5, 0, 0)])
    assert len(P.shape_intersect(w, (DET, s, 0), (0, 0, -5), (0, 0, 1))[0]) == 0


def test_sphere_normal(w):
    s = w.shape("sphere")
    for p in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
        assert np.array_equal(P.local_normal_at(w, (DET, s, 0), p), p)
    r3 = float(f32(np.sqrt(f32(3))) / f32(3))
    n = P.local_normal_at(w, (DET, s, 0), (r3, r3, r3))
    assert approx(n, (r3, r3, r3)) and approx(n, norm(n))
    s = w.shape("sphere", [("translate", 0, 1, 0)])
    assert approx(P.normal_at(w, (DET, s, 0), (0, 1.70711, -0.70711)), (0, 0.70711, -0.70711))
    s = w.shape("sphere", [("rotate-z", float(f32(math.pi) / f32(5))), ("scale", 1, 0.5, 1)])
    s2 = float(f32(np.sqrt(f32(2))) / f32(2))
    assert approx(P.normal_at(w, (DET, s, 0), (0, s2, -s2)), (0, 0.97014, -0.24254))


# ---- plane.rs:115-132 ----------------------------------------------------------------------
def test_plane_basics(w):
    p = w.shape("plane")
    for pt in ((0, 0, 0), (10, 0, -10), (-5, 0, 150)):
        assert np.array_equal(P.local_normal_at(w, (DET, p, 0), pt), (0, 1, 0))
    assert len(P.shape_intersect(w, (DET, p, 0), (0, 10, 0), (0, 0, 1))[0]) == 0
    assert len(P.shape_intersect(w, (DET, p, 0), (0, 0, 0), (0, 0, 1))[0]) == 0
    assert list(P.shape_intersect(w, (DET, p, 0), (0, 1, 0), (0, -1, 0))[0]) == [1]
    assert list(P.shape_intersect(w, (DET, p, 0), (0, -1, 0), (0, 1, 0))[0]) == [1]


# ---- cube.rs:171-218 -----------------------------------------------------------------------
@pytest.mark.parametrize("o,d,t1,t2", [
    ((5, 0.5, 0), (-1, 0, 0), 4, 6), ((-5, 0.5, 0), (1, 0, 0), 4, 6), ((0.5, 5, 0), (0, -1, 0), 4, 6),
    ((0.5, -5, 0), (0, 1, 0), 4, 6), ((0.5, 0, 5), (0, 0, -1), 4, 6), ((0.5, 0, -5), (0, 0, 1), 4, 6),
    ((0, 0.5, 0), (0, 0, 1), -1, 1)])
def test_cube_hits(w, o, d, t1, t2):
    c = w.shape("cube")
    assert list(P.shape_intersect(w, (DET, c, 0), o, d)[0]) == [t1, t2]


@pytest.mark.parametrize("o,d", [
    ((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)),
    ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)), ((2, 2, 0), (-1, 0, 0))])
def test_cube_misses(w, o, d):
    c = w.shape("cube")
    assert len(P.shape_intersect(w, (DET, c, 0), o, d)[0]) == 0


@pytest.mark.parametrize("p,n", [
    ((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, -0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
    ((0.3, -1, 0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
    ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))])
def test_cube_normals(w, p, n):
    c = w.shape("cube")
    assert np.array_equal(P.local_normal_at(w, (DET, c, 0), p), n)


# ---- cylinder.rs:208-299 -------------------------------------------------------------------
def test_cylinder_basics(w):
    c = w.shape("cylinder", params=[-INF, INF, 0])
    for o, d in (((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))):
        assert len(P.shape_intersect(w, (DET, c, 0), o, norm(d))[0]) == 0
    for o, d, t1, t2 in (((1, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (0, 0, 1), 4, 6),
                         ((0.5, 0, -5), (0.1, 1, 1), 6.80798, 7.08872)):
        ts, _ = P.shape_intersect(w, (DET, c, 0), o, norm(d))
        assert abs(ts[0] - t1) < EPS and abs(ts[1] - t2) < EPS
    for p, n in (((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)), ((-1, 1, 0), (-1, 0, 0))):
        assert np.array_equal(P.local_normal_at(w, (DET, c, 0), p), n)


def test_cylinder_advanced(w):
    c = w.shape("cylinder", params=[1, 2, 0])
    for o, d, count in (((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
                        ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)):
        assert len(P.shape_intersect(w, (DET, c, 0), o, norm(d))[0]) == count
    c = w.shape("cylinder", params=[1, 2, 1])
    for o, d, count in (((0, 3, 0), (0, -1, 0), 2), ((0, 3, -2), (0, -1, 2), 2), ((0, 0, -2), (0, 1, 2), 2)):
        assert len(P.shape_intersect(w, (DET, c, 0), o, norm(d))[0]) == count
    for p, n in (((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)),
                 ((0, 2, 0), (0, 1, 0)), ((0.5, 2, 0), (0, 1, 0)), ((0, 2, 0.5), (0, 1, 0))):
        assert np.array_equal(P.local_normal_at(w, (DET, c, 0), p), n)


# ---- cone.rs:228-284 -----------------------------------------------------------------------
def test_cone_basics(w):
    c = w.shape("cone", params=[-INF, INF, 0])
    for o, d, t1, t2 in (((0, 0, -5), (0, 0, 1), 5, 5), ((1, 1, -5), (-0.5, -1, 1), 4.55006, 49.44994)):
        ts, _ = P.shape_intersect(w, (DET, c, 0), o, norm(d))
        assert abs(ts[0] - t1) < EPS and abs(ts[1] - t2) < EPS
    ts, _ = P.shape_intersect(w, (DET, c, 0), (0, 0, -1), norm((0, 1, 1)))
    assert abs(ts[0] - 0.35355) < EPS
    cc = w.shape("cone", params=[-0.5, 0.5, 1])
    for o, d, count in (((0, 0, -5), (0, 1, 0), 0), ((0, 0, -0.25), (0, 1, 1), 2), ((0, 0, -0.25), (0, 1, 0), 4)):
        assert len(P.shape_intersect(w, (DET, cc, 0), o, norm(d))[0]) == count
    r2 = float(f32(np.sqrt(f32(2))))
    for p, n in (((0, 0, 0), (0, 0, 0)), ((1, 1, 1), (1, -r2, 1)), ((-1, -1, 0), (-1, 1, 0))):
        assert np.array_equal(P.local_normal_at(w, (DET, c, 0), p), np.array(n, dtype=f32))


# ---- triangle.rs:144-166, smooth_triangle.rs:148-173 ------------------------------------------
def test_triangle_basics(w):
    t = w.shape("triangle", params=[0, 1, 0, -1, 0, 0, 1, 0, 0])
    fields = P.triangle_fields(w, (DET, t, 0))
    assert np.array_equal(fields[3], (-1, -1, 0)) and np.array_equal(fields[4], (1, -1, 0))
    assert np.array_equal(fields[5], (0, 0, -1))
    for o, d in (((0, -1, -2), (0, 1, 0)), ((1, 1, -2), (0, 0, 1)), ((-1, 1, -2), (0, 0, 1)), ((0, -1, -2), (0, 0, 1))):
        assert len(P.shape_intersect(w, (DET, t, 0), o, d)[0]) == 0
    assert list(P.shape_intersect(w, (DET, t, 0), (0, 0.5, -2), (0, 0, 1))[0]) == [2]


def test_smooth_triangle_basics(w):
    tri = w.shape("smooth-triangle", params=[0, 1, 0, -1, 0, 0, 1, 0, 0, 0, 1, 0, -1, 0, 0, 1, 0, 0])
    ts, uvs = P.shape_intersect(w, (DET, tri, 0), (-0.2, 0.3, -2), (0, 0, 1))
    assert uvs[0][0] == f32(0.45) and uvs[0][1] == f32(0.25)
    n = P.normal_at(w, (DET, tri, 0), (0, 0, 0), 0.45, 0.25)
    assert approx(n, (-0.5547, 0.83205, 0))


# ---- shapes.rs:330-339 -----------------------------------------------------------------------
def test_shape_bounding_box(w):
    s = w.shape("sphere", [("scale", 0.5, 2, 4), ("translate", 1, -3, 5)])
    mn, mx = P.shape_bounds(w, (DET, s, 0), parent_space=True)
    assert np.array_equal(mn, (0.5, -5, 1)) and np.array_equal(mx, (1.5, -1, 9))
