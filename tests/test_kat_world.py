"""The reference's own unit-test known answers for the integrator, restated against the oracle
(SURVEY.md §4): src/camera.rs:95-126, src/world.rs:167-403, src/intersections.rs:203-369.
Every literal below is copied from the reference test named in the comment; exact `==` in the
reference is exact here, `equal_approx` is |d| < 1e-4 (src/tuples.rs:7-13)."""
import math

import numpy as np
import pytest

import orc_probe as P
from oracle import oracle
from raytracer_amd import builder as B

EPS = 1e-4
S2 = float(np.float32(np.sqrt(np.float32(2.0))) / np.float32(2.0))  # f32::sqrt(2.) / 2.
SQ2 = float(np.float32(np.sqrt(np.float32(2.0))))
DET = P.DET


def approx(a, b):
    return np.all(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) < EPS)


def default_world(light=((-10, 10, -10), (1, 1, 1)), m1=None, m2=None):
    """World::default() (src/world.rs:20-41); returns (world, s1, s2) with detached handles == world shapes."""
    w = oracle.new_world()
    w.add_point_light(*light)
    s1 = w.shape("sphere", (), m1 or B.Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
    s2 = w.shape("sphere", [("scale", 0.5, 0.5, 0.5)], m2 or B.Material())
    w.add_shape(s1)
    w.add_shape(s2)
    return w, s1, s2


# ---- camera.rs -----------------------------------------------------------------------------
def test_camera_basics():  # camera.rs:95-100
    w = oracle.new_world()
    for hs, vs in ((200, 125), (125, 200)):
        _, _, ps = w.ray_for_pixel(B.Camera(hs, vs, np.float32(math.pi) / np.float32(2), ops=[]), 0, 0)
        assert np.float32(ps) == np.float32(0.01)


def test_camera_rays():  # camera.rs:103-126
    w = oracle.new_world()
    fov = float(np.float32(math.pi) / np.float32(2))
    o, d, _ = w.ray_for_pixel(B.Camera(201, 101, fov, ops=[]), 100, 50)
    assert np.array_equal(o, [0, 0, 0]) and approx(d, [0, 0, -1])
    o, d, _ = w.ray_for_pixel(B.Camera(201, 101, fov, ops=[]), 0, 0)
    assert np.array_equal(o, [0, 0, 0]) and approx(d, [0.66519, 0.33259, -0.66851])
    cam = B.Camera(201, 101, fov, ops=[("translate", 0, -2, 5), ("rotate-y", float(np.float32(math.pi) / np.float32(4)))])
    o, d, _ = w.ray_for_pixel(cam, 100, 50)
    assert approx(o, [0, 2, -5]) and approx(d, [S2, 0, -S2])


# ---- world.rs --------------------------------------------------------------------------------
def test_world_basics():  # world.rs:167-176
    w, _, _ = default_world()
    ts, _ = P.intersect(w, (0, 0, -5), (0, 0, 1))
    assert list(ts) == [4.0, 4.5, 5.5, 6.0]


def test_world_shading():  # world.rs:179-226
    w, s1, s2 = default_world()
    c = P.comps_empty(w, (0, 0, -5), (0, 0, 1), 4.0, (DET, s1, 0), remaining=1, what=1)
    assert approx(c, [0.38066, 0.47583, 0.2855])

    w, s1, s2 = default_world(light=((0, 0.25, 0), (1, 1, 1)))
    c = P.comps_empty(w, (0, 0, 0), (0, 0, 1), 0.5, (DET, s2, 0), remaining=1, what=1)
    assert approx(c, [0.90498, 0.90498, 0.90498])

    w, _, _ = default_world()
    rgb, _, _, _ = w.color_at([(0, 0, -5)], [(0, 1, 0)], 1)
    assert np.array_equal(rgb[0], [0, 0, 0])
    rgb, _, _, _ = w.color_at([(0, 0, -5)], [(0, 0, 1)], 1)
    assert approx(rgb[0], [0.38066, 0.47583, 0.2855])

    w, _, _ = default_world(m1=B.Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0),
                            m2=B.Material(ambient=1.0))
    rgb, _, _, _ = w.color_at([(0, 0, 0.75)], [(0, 0, -1)], 1)
    assert approx(rgb[0], [1, 1, 1])  # inner sphere's material colour

    w, _, _ = default_world(light=((0, 0, -10), (1, 1, 1)))
    sa = w.shape("sphere")
    sb = w.shape("sphere", [("translate", 0, 0, 10)])
    w.add_shape(sa)
    w.add_shape(sb)
    c = P.comps_empty(w, (0, 0, 5), (0, 0, 1), 4.0, (DET, sb, 0), remaining=1, what=1)
    assert np.array_equal(c, np.array([0.1, 0.1, 0.1], dtype=np.float32))


@pytest.mark.parametrize("point,res", [((-10, -10, 10), False), ((10, 10, 10), True), ((-20, -20, -20), False),
                                       ((-5, -5, -5), False)])
def test_world_shadows(point, res):  # world.rs:229-241
    w, _, _ = default_world()
    assert P.is_shadowed(w, (-10, -10, -10), point) == res


def test_world_reflection():  # world.rs:244-294
    w, s1, s2 = default_world(m2=B.Material(ambient=1.0))
    c = P.comps_empty(w, (0, 0, 0), (0, 0, 1), 1.0, (DET, s2, 0), remaining=1, what=2)
    assert np.array_equal(c, [0, 0, 0])

    for what, want in ((2, [0.19032, 0.2379, 0.14274]), (1, [0.87677, 0.92436, 0.82918])):
        w, _, _ = default_world()
        pl = w.shape("plane", [("translate", 0, -1, 0)], B.Material(reflective=0.5))
        w.add_shape(pl)
        c = P.comps_empty(w, (0, 0, -3), (0, -S2, S2), SQ2, (DET, pl, 0), remaining=1, what=what)
        assert approx(c, want)

    # mutually reflective parallel planes terminate (world.rs:282-293)
    w, _, _ = default_world(light=((0, 0, 0), (1, 1, 1)))
    for y in (-1, 1):
        w.add_shape(w.shape("plane", [("translate", 0, y, 0)], B.Material(reflective=1.0)))
    rgb, _, _, _ = w.color_at([(0, 0, 0)], [(0, 1, 0)], 1)
    assert np.all(np.isfinite(rgb))


def test_world_refraction():  # world.rs:297-403
    w, s1, s2 = default_world()
    xs = [(4.0, (DET, s2, 0)), (6.0, (DET, s2, 0))]
    assert np.array_equal(P.comps(w, (0, 0, -5), (0, 0, 1), xs, 0, remaining=1, what=3), [0, 0, 0])

    glass = B.Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, transparency=1.0, refractive_index=1.5)
    w, s1, s2 = default_world(m1=glass)
    assert np.array_equal(P.comps(w, (0, 0, -5), (0, 0, 1), xs, 0, remaining=0, what=3), [0, 0, 0])

    w, s1, s2 = default_world(m1=glass)  # total internal reflection
    xs = [(-S2, (DET, s2, 0)), (S2, (DET, s2, 0))]
    assert np.array_equal(P.comps(w, (0, 0, S2), (0, 1, 0), xs, 1, remaining=1, what=3), [0, 0, 0])

    w = oracle.new_world()
    w.add_point_light((-10, 10, -10), (1, 1, 1))
    pat = w.pattern(B.PAT_XYZ_RGB)
    s1 = w.shape("sphere", (), B.Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0, pattern=pat))
    s2 = w.shape("sphere", [("scale", 0.5, 0.5, 0.5)], B.Material(transparency=1.0, refractive_index=1.5))
    w.add_shape(s1)
    w.add_shape(s2)
    xs = [(-0.9899, (DET, s1, 0)), (-0.4899, (DET, s2, 0)), (0.4899, (DET, s2, 0)), (0.9899, (DET, s1, 0))]
    c = P.comps(w, (0, 0, 0.1), (0, 1, 0), xs, 2, remaining=1, what=3)
    assert approx(c, [0, 0.99888, 0.04725])

    for floor_m, want in ((B.Material(transparency=0.5, refractive_index=1.5), [0.93642, 0.68642, 0.68642]),
                          (B.Material(reflective=0.5, transparency=0.5, refractive_index=1.5),
                           [0.93391, 0.69643, 0.69243])):
        w, _, _ = default_world()
        floor = w.shape("plane", [("translate", 0, -1, 0)], floor_m)
        w.add_shape(floor)
        ball = w.shape("sphere", [("translate", 0, -3.5, -0.5)], B.Material(color=(1, 0, 0), ambient=0.5))
        w.add_shape(ball)
        c = P.comps(w, (0, 0, -3), (0, -S2, S2), [(SQ2, (DET, floor, 0))], 0, remaining=1, what=1)
        assert approx(c, want)


# ---- intersections.rs --------------------------------------------------------------------------
def test_intersections_hit_rule():  # intersections.rs:203-241 (hit = lowest positive t, order-independent)
    for ts, want in (([2.0, 1.0], 1.0), ([2.0, -1.0], 2.0), ([-1.0, -2.0], None), ([5.0, 7.0, -3.0, 2.0], 2.0)):
        w = oracle.new_world()
        w.add_point_light((0, 10, 0), (1, 1, 1))
        # planes facing the ray at the wanted distances reproduce an explicit t list
        for t in ts:
            w.add_shape(w.shape("plane", [("rotate-x", math.pi / 2), ("translate", 0, 0, t)]))
        _, t, ids, _ = w.color_at([(0, 0, 0)], [(0, 0, 1)], 0)
        if want is None:
            assert ids[0] == -1
        else:
            assert ids[0] == ts.index(want) and abs(t[0] - want) < 1e-5


def test_intersection_precomputation():  # intersections.rs:244-262
    w = oracle.new_world()
    s = w.shape("sphere")
    c = P.comps_empty(w, (0, 0, -5), (0, 0, 1), 4.0, (DET, s, 0))
    assert np.array_equal(c.point, [0, 0, -1]) and np.array_equal(c.eyev, [0, 0, -1])
    assert np.array_equal(c.normalv, [0, 0, -1]) and c.inside is False
    c = P.comps_empty(w, (0, 0, 0), (0, 0, 1), 1.0, (DET, s, 0))
    assert np.array_equal(c.point, [0, 0, 1]) and np.array_equal(c.eyev, [0, 0, -1])
    assert np.array_equal(c.normalv, [0, 0, -1]) and c.inside is True


def test_intersection_offset():  # intersections.rs:265-282
    w = oracle.new_world()
    s = w.shape("sphere", [("translate", 0, 0, 1)])
    c = P.comps_empty(w, (0, 0, -5), (0, 0, 1), 5.0, (DET, s, 0))
    assert c.over_point[2] < -EPS / 2 and c.point[2] > c.over_point[2]
    s = w.shape("sphere", [("translate", 0, 0, -1)], B.Material(transparency=1.0, refractive_index=1.5))
    c = P.comps(w, (0, 0, -5), (0, 0, 1), [(5.0, (DET, s, 0))], 0)
    assert c.under_point[2] > -EPS / 2 and c.point[2] < c.under_point[2]


def test_reflection_vector():  # intersections.rs:285-294
    w = oracle.new_world()
    p = w.shape("plane")
    c = P.comps_empty(w, (0, 1, -1), (0, -S2, S2), S2, (DET, p, 0))
    assert np.array_equal(c.reflectv, np.array([0, S2, S2], dtype=np.float32))


def test_refraction_at_intersections():  # intersections.rs:297-334
    w = oracle.new_world()
    a = w.shape("sphere", [("scale", 2, 2, 2)], B.Material(transparency=1.0, refractive_index=1.5))
    b = w.shape("sphere", [("translate", 0, 0, -0.25)], B.Material(transparency=1.0, refractive_index=2.0))
    c = w.shape("sphere", [("translate", 0, 0, 0.25)], B.Material(transparency=1.0, refractive_index=2.5))
    xs = [(2.0, (DET, a, 0)), (2.75, (DET, b, 0)), (3.25, (DET, c, 0)), (4.75, (DET, b, 0)), (5.25, (DET, c, 0)),
          (6.0, (DET, a, 0))]
    for i, n1, n2 in ((0, 1.0, 1.5), (1, 1.5, 2.0), (2, 2.0, 2.5), (3, 2.5, 2.5), (4, 2.5, 1.5), (5, 1.5, 1.0)):
        cc = P.comps(w, (0, 0, -4), (0, 0, 1), xs, i)
        assert cc.n1 == n1 and cc.n2 == n2


def test_fresnel_effect():  # intersections.rs:337-369
    w = oracle.new_world()
    s = w.shape("sphere", (), B.Material(transparency=1.0, refractive_index=1.5))
    c = P.comps(w, (0, 0, S2), (0, 1, 0), [(-S2, (DET, s, 0)), (S2, (DET, s, 0))], 1)
    assert c.schlick == 1.0
    c = P.comps(w, (0, 0, 0), (0, 1, 0), [(-1.0, (DET, s, 0)), (1.0, (DET, s, 0))], 1)
    assert abs(c.schlick - 0.04) < EPS
    c = P.comps(w, (0, 0.99, -2), (0, 0, 1), [(1.8589, (DET, s, 0))], 0)
    assert abs(c.schlick - 0.48873) < EPS
