"""Exactness fuzz on the GPU: random worlds of cubes and spheres (tests/test_gpu_parity.py _random_solids) and
small-dragon scenes with random cameras, random rays through rt_color_at, hit ids and t against the oracle.
usage: gpu_fuzz.py [frames | parts | few | copies] [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from raytracer_amd import host  # noqa: E402
from raytracer_amd.canvas import to_u8  # noqa: E402
from test_gpu_parity import _random_solids  # noqa: E402
from conftest import small_dragons  # noqa: E402

def mixed_world(w, rng, rich=False, loose=False):
    """Every primitive kind, patterns, nested transformed groups, a divided triangle mesh, CSG trees, two lights.
    rich: also texture maps (checker, align-check, image) through the three uv mappings, cube maps, xyz-rgb, and an
    area light in place of one point light."""
    from raytracer_amd import builder as B

    def uv(j):
        if j % 3 == 0:
            return w.uv_checker(float(rng.randint(2, 12)), float(rng.randint(2, 12)), tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3)))
        if j % 3 == 1:
            return w.uv_align_check(*[tuple(rng.uniform(0, 1, 3)) for _ in range(5)])
        return w.uv_image(rng.uniform(0, 1, size=(int(rng.randint(2, 40)), int(rng.randint(2, 40)), 3)).astype(np.float32))

    def rich_pattern(i):
        ops = [("scale", *rng.uniform(0.3, 1.5, 3)), ("rotate-y", rng.uniform(-2, 2))]
        if i % 3 == 0:
            return w.pattern(B.PAT_TEXTURE_MAP, None, ops, [B.UVMAP_SPHERICAL, B.UVMAP_PLANAR, B.UVMAP_CYLINDRICAL][i // 3 % 3], [uv(i // 9)])
        if i % 3 == 1:
            return w.pattern(B.PAT_CUBE_MAP, None, ops, 0, [uv(i + j) for j in range(6)])
        return w.pattern(B.PAT_XYZ_RGB, None, ops)

    def xf():
        return [("scale", *rng.uniform(0.3, 2.5, 3)), ("rotate-x", rng.uniform(-3, 3)), ("rotate-y", rng.uniform(-3, 3)),
                ("translate", *rng.uniform(-5, 5, 3))]

    def mat(i):
        pat = -1
        if rich and i % 2 == 1:
            pat = rich_pattern(i // 2)
        elif i % 3 == 0:
            kind = [B.PAT_STRIPED, B.PAT_GRADIENT, B.PAT_RING, B.PAT_CHECKER][i // 3 % 4]
            pat = w.pattern(kind, [tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3))],
                            [("scale", *rng.uniform(0.2, 1.5, 3)), ("rotate-z", rng.uniform(-2, 2))])
        return B.Material(color=tuple(rng.uniform(0, 1, 3)), reflective=0.4 * (i % 4 == 1), transparency=0.6 * (i % 5 == 2),
                          refractive_index=1.0 + 0.5 * (i % 5 == 2), pattern=pat)

    def solid(i):
        kind = ["sphere", "cube", "cylinder", "cone"][i % 4]
        params = None
        if kind in ("cylinder", "cone"):
            lo = rng.uniform(-2, 0)
            params = [lo, lo + rng.uniform(0.5, 3), float(i % 8 < 4)]
        return w.shape(kind, xf(), mat(i), shadow=bool(i % 6), params=params)

    w.add_shape(w.shape("plane", [("translate", 0, -6, 0)], mat(0)))
    for i in range(int(rng.randint(3, 10))):
        w.add_shape(solid(i))
    if loose:  # no groups at all: the direct kernels (rt_hip.hip k_render_direct / k_color_at_direct)
        if rng.randint(2):
            ops = [B.CSG_UNION, B.CSG_INTERSECT, B.CSG_DIFFERENCE]
            w.add_csg(w.csg(int(rng.choice(ops)), B.CHILD_SHAPE, solid(40), B.CHILD_SHAPE, solid(41)))
        if rng.randint(2):
            w.add_area_light((8, 9, -4), (2, 0, 0), 2, (0, 0, 2), 3, (0.8, 0.7, 0.7))
        else:
            w.add_point_light((8, 9, -4), (0.8, 0.7, 0.7))
        return
    outer = w.group(xf())
    inner = w.group(xf())
    for i in range(int(rng.randint(2, 9))):
        w.group_add_shape(inner if i % 2 else outer, solid(20 + i))
    w.group_add_child(outer, inner)
    w.add_group(outer)
    # a random triangle soup, divided like an OBJ group
    nv, nt = int(rng.randint(20, 200)), int(rng.randint(30, 400))
    pos = rng.uniform(-2, 2, size=(nv, 3)).astype(np.float32)
    vidx = np.stack([rng.choice(nv, 3, replace=False) for _ in range(nt)]).astype(np.int32)
    mesh = w.group(xf())
    smooth = bool(rng.randint(2))
    nrm = rng.normal(size=(nv, 3)).astype(np.float32) if smooth else None
    w.group_add_mesh(mesh, mat(7), pos, nrm, vidx, vidx if smooth else None)
    w.group_divide(mesh, int(rng.choice([1, 4, 20])))
    w.add_group(mesh)
    # CSG: (a op b) op c
    ops = [B.CSG_UNION, B.CSG_INTERSECT, B.CSG_DIFFERENCE]
    c1 = w.csg(int(rng.choice(ops)), B.CHILD_SHAPE, solid(40), B.CHILD_SHAPE, solid(41))
    c2 = w.csg(int(rng.choice(ops)), B.CHILD_CSG, c1, B.CHILD_SHAPE, solid(42))
    w.add_csg(c2)
    w.add_point_light((-10, 12, -10), (0.7, 0.7, 0.7))
    if rich and rng.randint(2):
        w.add_area_light((8, 9, -4), (2, 0, 0), 2, (0, 0, 2), 3, (0.4, 0.3, 0.3))
    else:
        w.add_point_light((8, 9, -4), (0.4, 0.3, 0.3))


def instanced_world(w, rng):
    """The same opaque triangle mesh (a bumpy height field plus loose soup triangles, some listed twice) parsed into two
    or three differently transformed, divided groups, as scene.rs does for every `obj:` instance: the auxiliary trees
    are shared between them (rt_accel.h).  A floor, a reflective sphere, one light."""
    from raytracer_amd import builder as B
    n = int(rng.randint(8, 48))
    gx, gz = np.meshgrid(np.linspace(-2, 2, n), np.linspace(-2, 2, n))
    pos = np.stack([gx, 0.4 * np.sin(3 * gx) * np.cos(2 * gz) + rng.uniform(-0.05, 0.05, gx.shape), gz], axis=-1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    soup = np.stack([rng.choice(n * n, 3, replace=False) for _ in range(int(rng.randint(1, 60)))]).reshape(-1, 3)
    tris = np.concatenate([tris, soup, tris[rng.choice(len(tris), 5)]]).astype(np.int32)
    smooth = bool(rng.randint(2))
    nrm = np.stack([-1.2 * np.cos(3 * gx) * np.cos(2 * gz), np.ones_like(gx), 0.8 * np.sin(3 * gx) * np.sin(2 * gz)], -1).reshape(-1, 3)
    threshold = int(rng.choice([1, 8, 50]))
    for k in range(int(rng.randint(2, 4))):
        g = w.group([("scale", *rng.uniform(0.5, 2.0, 3)), ("rotate-y", rng.uniform(-3, 3)), ("rotate-x", rng.uniform(-0.5, 0.5)),
                     ("translate", *rng.uniform(-4, 4, 3))])
        mat = B.Material(color=tuple(rng.uniform(0, 1, 3)), reflective=0.3 * (k == 1))
        w.group_add_mesh(g, mat, pos.astype(np.float32), nrm.astype(np.float32) if smooth else None, tris, tris if smooth else None)
        w.group_divide(g, threshold)
        w.add_group(g)
    w.add_shape(w.shape("plane", [("translate", 0, -3, 0)], B.Material(color=(0.8, 0.8, 0.8), reflective=0.2)))
    w.add_shape(w.shape("sphere", [("translate", 0, 3, 0)], B.Material(color=(0.2, 0.2, 0.3), reflective=0.6)))
    w.add_point_light((-6, 10, -8), (1, 1, 1))


def run_seed(seed, n=40000, log=print):
    """One world and one batch of rays; returns the number of failed checks."""
    bad = 0
    rng = np.random.RandomState(1000 + seed)
    ng, nl = int(rng.randint(4, 28)), int(rng.randint(0, 24))
    if seed >= 3000:
        gw, ow = host.new_world(0), oracle.new_world()
        instanced_world(gw, np.random.RandomState(seed))
        instanced_world(ow, np.random.RandomState(seed))
        span, what = 9.0, "instanced"
    elif seed % 4 == 3 and seed < 2000:
        scene = small_dragons(int(rng.randint(8, 40)), int(rng.randint(6, 30)))
        gw, ow = scene.build(host.new_world(0)), scene.build(oracle.new_world())
        span, what = 12.0, "small dragons"
    elif seed % 4 == 1 or seed >= 2000:
        gw, ow = host.new_world(0), oracle.new_world()
        rich, loose = seed >= 1000, seed >= 2000
        mixed_world(gw, np.random.RandomState(seed), rich, loose)
        mixed_world(ow, np.random.RandomState(seed), rich, loose)
        span, what = 12.0, "loose rich" if loose else "mixed rich" if rich else "mixed"
    else:
        gw, ow = host.new_world(0), oracle.new_world()
        _random_solids(gw, np.random.RandomState(seed), ng, nl)
        _random_solids(ow, np.random.RandomState(seed), ng, nl)
        span, what = 14.0, f"solids {ng}+{nl}"
    o = rng.uniform(-span, span, size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(-span / 2, span / 2, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10
    d[:k] *= rng.uniform(1e-3, 1e3, size=(k, 1)).astype(np.float32)
    d[k:2 * k, int(rng.randint(3))] = 0.0
    d[2 * k:3 * k, int(rng.randint(3))] = 3e-5
    for remaining in (0, 3):
        # (a ray whose n1/n2 list passes RT_XS_CAP continues in the context's side lists: nothing is skipped)
        grgb, gt, gid, _ = gw.color_at(o, d, remaining)
        rrgb, rt, rid, _ = ow.color_at(o, d, remaining)
        hit = rid >= 0
        ids_ok = np.array_equal(gid, rid)
        t_ok = np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32)) if ids_ok else False
        both_nan = np.isnan(grgb).any(axis=1) & np.isnan(rrgb).any(axis=1)
        dd = np.abs(to_u8(np.nan_to_num(grgb)).astype(int) - to_u8(np.nan_to_num(rrgb)).astype(int)).max(axis=1)
        frac = float(((dd <= 1) | both_nan).mean())
        ok = ids_ok and t_ok and frac >= 0.999
        bad += not ok
        log(f"seed {seed} {what} remaining {remaining}: hits {hit.mean():.2f} ids {'ok' if ids_ok else (gid != rid).sum()} "
            f"t {'ok' if t_ok else 'DIFF'} colour within 1 LSB {frac:.5f} {'' if ok else '<-- FAIL'}")
    return bad


class _FrameScene:
    """The world and camera of run_frame(seed), in the form tests/test_gpu_parity.py Raw takes."""

    def __init__(self, seed, cam):
        self.seed, self.cam = seed, cam

    def build(self, w):
        mixed_world(w, np.random.RandomState(self.seed), True, bool(self.seed % 2))
        return w

    def camera(self, hsize, vsize):
        return self.cam


def run_frame(seed, log=print):
    """One mixed world (rich; group-less for odd seeds) rendered whole through Camera::render with a random camera,
    odd frame sizes, both jitter modes: the bars of tests/test_gpu_parity.py check().  Even seeds have two lights: there
    the reference re-traces secondary rays per light, so the rays TRACED differ by design and the
    device's count of the rays the reference casts (RtStats.rays_reference) is held against the oracle's total."""
    from raytracer_amd import abi, builder as B
    from test_gpu_parity import Raw, check
    rng = np.random.RandomState(5000 + seed)
    gw, ow = host.new_world(0), oracle.new_world()
    mixed_world(gw, np.random.RandomState(seed), True, bool(seed % 2))
    mixed_world(ow, np.random.RandomState(seed), True, bool(seed % 2))
    eye = rng.uniform(-11, 11, 3)
    eye[1] = abs(eye[1]) + 1.0
    cam = B.Camera(int(rng.randint(17, 230)), int(rng.randint(9, 140)), float(rng.uniform(0.5, 1.6)), tuple(eye),
                   tuple(rng.uniform(-2, 2, 3)), (0, 1, 0))
    ro = B.RenderOptions(jitter_mode=B.JITTER_COUNTER if seed % 3 else B.JITTER_CONST, seed=seed)
    g = gw.render(cam, ro, aux=True)
    r = ow.render(cam, ro, aux=True)
    n_lights = int(gw.flatten().n_lights)
    try:
        check(g, r, f"frame seed {seed}", one_light=n_lights == 1)
        if n_lights != 1:
            _, _, st = Raw(_FrameScene(seed, cam), (cam.hsize, cam.vsize)).render(flags=abi.RT_RENDER_COUNT_WORK)
            want = sum(r.stats[k] for k in ("primary", "shadow", "reflect", "refract"))
            assert st.rays_reference == want, f"rays_reference {st.rays_reference} != oracle total {want}"
    except AssertionError as e:
        log(f"frame seed {seed} {cam.hsize}x{cam.vsize}: FAIL {e}")
        return 1
    log(f"frame seed {seed} {cam.hsize}x{cam.vsize}: ok, hits {r.hit_mask.mean():.2f}")
    return 0


def few_world(w, rng):
    """One to four loose shapes of any kind (any of them shadow-less), nothing else, under an area light of up to 4x4
    cells or a point light: the worlds whose shadow rays keep their origin in every shape's space and meet the
    shadow-casting shapes first (rt_device.h hoisted_is_shadowed; area lights only, rt_hip.hip launch_part)."""
    from raytracer_amd import builder as B
    n = int(rng.randint(1, 5))
    for i in range(n):
        kind = ["sphere", "cube", "cylinder", "cone", "plane"][int(rng.randint(5))]
        params = None
        if kind in ("cylinder", "cone"):
            lo = rng.uniform(-2, 0)
            params = [lo, lo + rng.uniform(0.5, 3), float(rng.randint(2))]
        ops = [("scale", *rng.uniform(0.3, 3.0, 3)), ("rotate-x", rng.uniform(-3, 3)), ("rotate-z", rng.uniform(-3, 3)),
               ("translate", *rng.uniform(-4, 4, 3))]
        if kind == "plane":
            ops = [("rotate-x", rng.uniform(-0.3, 0.3)), ("translate", 0, rng.uniform(-6, -2), 0)]
        if rng.randint(4) == 0:
            ops = []  # identity transform: the origin is used as it is
        m = B.Material(color=tuple(rng.uniform(0, 1, 3)), reflective=0.5 * (rng.randint(3) == 0),
                       transparency=0.7 * (rng.randint(4) == 0), refractive_index=float(rng.choice([1.0, 1.5])))
        w.add_shape(w.shape(kind, ops, m, shadow=bool(rng.randint(3)), params=params))
    corner = tuple(rng.uniform(-8, 8, 3) + np.array([0, 10, 0]))
    w.add_area_light(corner, tuple(rng.uniform(-3, 3, 3)), int(rng.randint(1, 5)), tuple(rng.uniform(-3, 3, 3)),
                     int(rng.randint(1, 5)), tuple(rng.uniform(0.3, 1, 3)))
    if rng.randint(3) == 0:
        w.add_point_light(tuple(rng.uniform(-10, 10, 3)), (0.3, 0.3, 0.3))


def run_few(seed, log=print):
    """A few_world frame against the oracle (bars of tests/test_gpu_parity.py check)."""
    from raytracer_amd import builder as B
    from test_gpu_parity import check
    rng = np.random.RandomState(11000 + seed)
    gw, ow = host.new_world(0), oracle.new_world()
    few_world(gw, np.random.RandomState(seed))
    few_world(ow, np.random.RandomState(seed))
    eye = rng.uniform(-9, 9, 3)
    cam = B.Camera(int(rng.randint(17, 200)), int(rng.randint(9, 120)), float(rng.uniform(0.5, 1.6)), tuple(eye),
                   tuple(rng.uniform(-2, 2, 3)), (0, 1, 0))
    ro = B.RenderOptions(jitter_mode=B.JITTER_COUNTER if seed % 3 else B.JITTER_CONST, seed=seed)
    g = gw.render(cam, ro, aux=True)
    r = ow.render(cam, ro, aux=True)
    try:
        check(g, r, f"few seed {seed}", one_light=int(gw.flatten().n_lights) == 1)
    except AssertionError as e:
        log(f"few seed {seed} {cam.hsize}x{cam.vsize}: FAIL {e}")
        return 1
    log(f"few seed {seed} {cam.hsize}x{cam.vsize}: ok, hits {r.hit_mask.mean():.2f}, shadow rays {r.stats['shadow']}")
    return 0


def copies_world(w, rng, nested=False):
    """Triangles listed more than once, next to triangles of the same corners in another order (another Shape, often
    the very same t) and the odd sphere: one to three groups, each under a transform of its own, and loose shapes.
    nested: trees of groups, most of them without a transform, that draw from ONE pool of triangles -- copies in
    different groups under one transformed ancestor, some in groups whose box the ray misses -- some of them divided
    (groups.rs:198-299).  (rt_device.h end_of_copy_run)"""
    import itertools
    perms = list(itertools.permutations(range(3)))

    def new_pool():
        exact = rng.randint(2) == 0  # small integers: every corner order gives the same t far more often
        pool = []
        for _ in range(int(rng.randint(1, 4))):
            c = rng.randint(-3, 4, size=(3, 3)).astype(float) if exact else rng.uniform(-3, 3, size=(3, 3))
            c[:, 2] = c[:, 2] * 0.25
            pool.append(c)
        return pool

    solids = []
    for _ in range(int(rng.randint(0, 3))):
        kind = ["sphere", "cube", "cylinder", "cone"][int(rng.randint(4))]
        solids.append((kind, [("scale", *rng.uniform(0.3, 1.2, 3)), ("translate", *(rng.randint(-4, 5, 3) * 0.5))],
                       [-1.0, 1.0, float(rng.randint(2))] if kind in ("cylinder", "cone") else None))

    def fill(add, pool=None, lo=2, hi=12):
        pool = pool or new_pool()
        smooth = rng.randint(4) == 0
        for _ in range(int(rng.randint(lo, hi))):
            if rng.randint(8) == 0:
                add(w.shape("sphere", [("scale", *rng.uniform(0.3, 1.5, 3)), ("translate", *rng.uniform(-2, 2, 3))], None))
                continue
            if solids and rng.randint(4) == 0:  # solids listed more than once, in two materials
                from raytracer_amd import builder as B
                kind, ops, params = solids[int(rng.randint(len(solids)))]
                add(w.shape(kind, ops, B.Material(color=(1, 0.2, 0.2)) if rng.randint(3) else B.Material(color=(0.2, 1, 0.2)), params=params))
                continue
            c = pool[int(rng.randint(len(pool)))][list(perms[int(rng.randint(6)) if rng.randint(2) else 0])]
            if smooth:
                add(w.shape("smooth-triangle", (), None, params=list(c.ravel()) + [0, 0, -1, 0.1, 0, -1, 0, 0.1, -1]))
            else:
                add(w.shape("triangle", (), None, params=list(c.ravel())))

    def some_ops():
        return [("rotate-y", rng.uniform(-0.5, 0.5)), ("translate", *rng.uniform(-3, 3, 3))] if rng.randint(2) else \
               [("translate", *(rng.randint(-6, 7, 3) * 0.5))]

    if nested:
        pool = new_pool() + new_pool()

        def in_plane():  # equal shapes under different transforms, hit at the same t more often than not
            return [("translate", float(rng.randint(-4, 5)) * 0.5, float(rng.randint(-4, 5)) * 0.5, 0.0)]

        def tree(depth):
            g = w.group((in_plane() if rng.randint(2) else some_ops()) if rng.randint(3) == 0 else ())
            fill(lambda sh: w.group_add_shape(g, sh), pool, 0, 7)
            for _ in range(int(rng.randint(0, 4)) if depth < 3 else 0):
                w.group_add_child(g, tree(depth + 1))
            return g

        for _ in range(int(rng.randint(1, 3))):
            g = tree(0)
            if rng.randint(3) == 0:
                w.group_divide(g, int(rng.randint(1, 5)))
            w.add_group(g)
        if rng.randint(2):
            fill(w.add_shape, pool)
        for _ in range(int(rng.randint(0, 3)) if rng.randint(2) else 0):  # csgs whose leaves come from the same pool
            from raytracer_amd import builder as B
            leaves = []
            fill(leaves.append, pool, 3, 4)
            ops = [B.CSG_UNION, B.CSG_INTERSECT, B.CSG_DIFFERENCE]
            c = w.csg(int(rng.choice(ops)), B.CHILD_SHAPE, leaves[0], B.CHILD_SHAPE, leaves[1])
            if rng.randint(2):
                c = w.csg(int(rng.choice(ops)), B.CHILD_CSG, c, B.CHILD_SHAPE, leaves[2]) if rng.randint(2) else \
                    w.csg(int(rng.choice(ops)), B.CHILD_SHAPE, leaves[2], B.CHILD_CSG, c)
            w.add_csg(c)
        w.add_point_light(tuple(rng.uniform(-10, 10, 3)), (1, 1, 1))
        return
    for _ in range(int(rng.randint(1, 4))):
        g = w.group(some_ops())
        fill(lambda sh: w.group_add_shape(g, sh))
        w.add_group(g)
    if rng.randint(2):
        fill(w.add_shape)
    w.add_point_light(tuple(rng.uniform(-10, 10, 3)), (1, 1, 1))


def run_copies(seed, log=print):
    """A copies_world frame and 4 000 random rays against the oracle (odd seeds: the nested worlds)."""
    from raytracer_amd import builder as B
    from test_gpu_parity import check
    rng = np.random.RandomState(13000 + seed)
    gw, ow = host.new_world(0), oracle.new_world()
    copies_world(gw, np.random.RandomState(seed), bool(seed % 2))
    copies_world(ow, np.random.RandomState(seed), bool(seed % 2))
    eye = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.choice([-9.0, 9.0])])
    if rng.randint(2):
        eye = np.round(eye)
    cam = B.Camera(int(rng.randint(17, 200)), int(rng.randint(9, 120)), float(rng.uniform(0.5, 1.3)), tuple(eye),
                   (0, 0, 0), (0, 1, 0))
    g = gw.render(cam, aux=True)
    r = ow.render(cam, aux=True)
    o = np.tile(eye.astype(np.float32), (4000, 1))
    target = rng.randint(-6, 7, size=(4000, 3)) * 0.5 if rng.randint(2) else rng.uniform(-3, 3, size=(4000, 3))
    d = (target - eye).astype(np.float32)
    if rng.randint(2):  # axis-parallel rays through half-integer points
        o = np.column_stack([target[:, 0], target[:, 1], np.full(4000, eye[2])]).astype(np.float32)
        d = np.tile(np.array([[0, 0, -np.sign(eye[2])]], dtype=np.float32), (4000, 1))
    _, gt, gid, guv = gw.color_at(o, d, 1)
    _, rt, rid, ruv = ow.color_at(o, d, 1)
    try:
        check(g, r, f"copies seed {seed}")
        assert np.array_equal(gid, rid), f"ray ids differ at {np.nonzero(gid != rid)[0][:5]}"
        hit = rid >= 0
        assert np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32)), "ray t differs"
    except AssertionError as e:
        log(f"copies seed {seed} {cam.hsize}x{cam.vsize}: FAIL {e}")
        return 1
    moved = int((r.object_id >= 0).sum())
    log(f"copies seed {seed} {cam.hsize}x{cam.vsize}: ok, pixels hit {moved}, rays hit {int((rid >= 0).sum())}")
    return 0


class _FuzzScene:
    """What tests/test_gpu_parity.py Raw needs of a scene: build(world) and camera(w, h)."""

    def __init__(self, seed):
        self.seed = seed
        rng = np.random.RandomState(7000 + seed)
        self.eye = rng.uniform(-11, 11, 3)
        self.eye[1] = abs(self.eye[1]) + 1.0
        self.to, self.fov = rng.uniform(-2, 2, 3), float(rng.uniform(0.5, 1.6))

    def build(self, w):
        mixed_world(w, np.random.RandomState(self.seed), True, bool(self.seed % 2))
        return w

    def camera(self, hsize, vsize):
        from raytracer_amd import builder as B
        return B.Camera(hsize, vsize, self.fov, tuple(self.eye), tuple(self.to), (0, 1, 0))


def run_parts(seed, log=print):
    """A random frame through rt_render_device as 1 part and as n interleaved parts of random stripe height: same bytes."""
    from raytracer_amd import tiles
    from test_gpu_parity import Raw
    rng = np.random.RandomState(9000 + seed)
    size = (int(rng.randint(9, 400)), int(rng.randint(5, 300)))
    raw = Raw(_FuzzScene(seed), size)
    full, ids, st = raw.render()
    n, tile_h = int(rng.randint(2, 9)), int(rng.choice([1, 2, 3, 8, 16]))
    canvas, idc, primary = np.full_like(full, np.nan), np.full_like(ids, -9), 0
    for p in range(n):
        rows = tiles.part_rows(size[1], tile_h, n, p)
        rgb, oid, s = raw.render(tile_h=tile_h, n_parts=n, part=p)
        if len(rows):
            canvas[rows], idc[rows] = rgb[: len(rows)], oid[: len(rows)]
        primary += s.primary
    ok = np.array_equal(canvas.view(np.uint32), full.view(np.uint32)) and np.array_equal(idc, ids) and primary == st.primary
    log(f"parts seed {seed} {size[0]}x{size[1]} n {n} tile_h {tile_h}: {'ok' if ok else 'FAIL'}")
    return 0 if ok else 1


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "parts":
        first, count = int(sys.argv[2]), int(sys.argv[3])
        bad = sum(run_parts(seed, log=lambda m: print(m, flush=True)) for seed in range(first, first + count))
        print("failures:", bad)
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "copies":
        first, count = int(sys.argv[2]), int(sys.argv[3])
        bad = sum(run_copies(seed, log=lambda m: print(m, flush=True)) for seed in range(first, first + count))
        print("failures:", bad)
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "few":
        first, count = int(sys.argv[2]), int(sys.argv[3])
        bad = sum(run_few(seed, log=lambda m: print(m, flush=True)) for seed in range(first, first + count))
        print("failures:", bad)
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "frames":
        first, count = int(sys.argv[2]), int(sys.argv[3])
        bad = sum(run_frame(seed, log=lambda m: print(m, flush=True)) for seed in range(first, first + count))
        print("failures:", bad)
        sys.exit(1 if bad else 0)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    bad = 0
    for seed in range(first, first + count):
        bad += run_seed(seed, log=lambda m: print(m, flush=True))
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
