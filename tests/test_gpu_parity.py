"""Parity of the HIP render path against the CPU oracle, through the C ABI (librt_hip.so via
librt_host.so / ctypes).  Bars (BASELINE.json north_star): primary hit mask and object-id map
bit-exact; 8-bit channels within +-1 LSB on >= 99.9 % of pixels (tolerance written in `check`).
At BASELINE's full sizes (4K) the oracle is too slow, so those cases use size-independent
properties: parts reassemble to the same bytes, pruning / auxiliary BVHs change nothing."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

from conftest import golden_path, scene_path, small_dragons
from raytracer_amd import abi, builder as B, host
from raytracer_amd.canvas import to_u8
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu

LSB_FRACTION = 0.999  # >= 99.9 % of pixels within +-1 LSB per channel


def _oracle():
    from oracle import oracle
    return oracle


def check(g, r, name, one_light=True):
    """one_light False: the world has several lights, where the reference re-traces every secondary ray once per light
    (world.rs:66-86) and the device traces it once and weights it: the per-kind counts of rays TRACED then
    differ by design, and the caller compares RtStats.rays_reference with the oracle's total instead."""
    assert g.stats["overflow"] == 0, name
    assert np.array_equal(g.hit_mask, r.hit_mask), f"{name}: primary hit mask differs"
    assert np.array_equal(g.object_id, r.object_id), f"{name}: primary object ids differ"
    hit = r.hit_mask.astype(bool)
    assert np.array_equal(g.hit_t[hit].view(np.uint32), r.hit_t[hit].view(np.uint32)), f"{name}: primary hit t differs"
    d = np.abs(to_u8(g.rgb).astype(int) - to_u8(r.rgb).astype(int)).max(axis=2)
    frac = float((d <= 1).mean())
    assert frac >= LSB_FRACTION, f"{name}: only {frac:.5f} of pixels within 1 LSB"
    for k in ("primary", "shadow", "reflect", "refract") if one_light else ("primary",):
        assert g.stats[k] == r.stats[k], f"{name}: {k} ray count {g.stats[k]} != oracle {r.stats[k]}"


SCENES = [("basic_scene", None), ("cover", (640, 640)), ("csg", None), ("checkered_cube", None),
          ("checkered_sphere", None), ("checkered_plane", None), ("checkered_cylinder", None), ("space_ship", (640, 360)),
          ("space_teapot", (640, 360)), ("teapot_low", (480, 270))]


@pytest.mark.parametrize("name,size", SCENES)
def test_scene_matches_oracle(name, size, obj_files):
    scene = Scene.from_file(scene_path(name), obj_files=obj_files)
    cam = scene.camera(*size) if size else scene.camera()
    g = scene.build(host.new_world(0)).render(cam, aux=True)
    r = scene.build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, name)


@pytest.mark.parametrize("name", ["basic_scene", "cover", "csg", "checkered_cube", "checkered_sphere",
                                  "checkered_plane", "checkered_cylinder", "space_ship", "space_teapot"])
def test_scene_matches_reference_render(name, obj_files):
    """Directly against the reference's own renders (tests/golden = samples/rendered), native size."""
    scene = Scene.from_file(scene_path(name), obj_files=obj_files)
    g = scene.build(host.new_world(0)).render(scene.camera())
    gold = np.asarray(Image.open(golden_path(name)).convert("RGB"), dtype=np.uint8)
    d = np.abs(to_u8(g.rgb).astype(int) - gold.astype(int)).max(axis=2)
    assert float((d <= 1).mean()) >= LSB_FRACTION, f"{name}: {(d > 1).sum()} pixels off by more than 1 LSB"


def _write_ppm(path, rng, w, h, binary):
    """A seeded stand-in for the textures the reference does not ship (README.md:120-138): smooth gradients plus noise,
    so that a texel picked one off shows up as a colour error."""
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255) // max(w - 1, 1), (y * 255) // max(h - 1, 1), (x + y) % 256], axis=2)
    img = ((img + rng.randint(0, 64, size=img.shape)) % 256).astype(np.uint8)
    with open(path, "wb") as f:
        if binary:
            f.write(b"P6\n%d %d\n255\n" % (w, h) + img.tobytes())
        else:
            f.write(b"P3\n%d %d\n255\n" % (w, h) + b"\n".join(b" ".join(b"%d" % v for v in row.ravel()) for row in img) + b"\n")


@pytest.mark.parametrize("name,files,tex", [
    ("earth", ["earthmap1k.ppm"], (96, 48)),
    ("skybox", ["negx.ppm", "posx.ppm", "negy.ppm", "posy.ppm", "negz.ppm", "posz.ppm"], (40, 40))])
def test_image_textures_match_oracle(name, files, tex, tmp_path):
    """earth.yaml / skybox.yaml (UvImage through the spherical and the cube map, uv_pattern.rs:117-132) with seeded
    synthetic PPMs in place of the downloads the reference names: the texel tables in HBM give the oracle's colours."""
    rng = np.random.RandomState(3)
    paths = []
    for k, f in enumerate(files):
        paths.append(str(tmp_path / f))
        _write_ppm(paths[-1], rng, tex[0], tex[1], binary=k % 2 == 0)
    scene = Scene.from_file(scene_path(name), ppm_files=paths)
    cam = scene.camera(480, 270)
    g = scene.build(host.new_world(0)).render(cam, aux=True)
    r = scene.build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, name)
    assert len(np.unique(to_u8(g.rgb).reshape(-1, 3), axis=0)) > 500  # the texture is what is seen


@pytest.mark.parametrize("mode,seed", [(B.JITTER_CONST, 0), (B.JITTER_COUNTER, 1), (B.JITTER_COUNTER, 12345)])
def test_soft_shadows_area_light(mode, seed):
    scene = Scene.from_file(scene_path("soft_shadows"))
    cam = scene.camera(320, 180)
    ro = B.RenderOptions(jitter_mode=mode, seed=seed)
    g = scene.build(host.new_world(0)).render(cam, ro, aux=True)
    r = scene.build(_oracle().new_world()).render(cam, ro, aux=True)
    check(g, r, f"soft_shadows mode={mode}")
    assert g.stats["shadow"] > 100 * 0.5 * cam.hsize * cam.vsize  # 10x10 samples per shaded hit


def test_small_dragons_matches_oracle():
    scene = small_dragons()
    cam = scene.camera(600, 240)
    g = scene.build(host.new_world(0)).render(cam, aux=True)
    r = scene.build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, "small dragons")
    assert g.stats["refract"] > 0 and g.stats["reflect"] > 0  # the transparent shadow-less bbox cubes are in play


def test_oversized_subtrees_covered_piecewise_match_oracle(monkeypatch):
    """A subtree beyond the size of one auxiliary tree is covered by trees over its child groups, and its own
    triangles keep the reference's linear tests (rt_accel.h; RT_AUX_MAX_TRIS lowers the 3.1 M limit for the test):
    the same bytes as with whole-subtree trees, and the oracle's ids."""
    scene = small_dragons()
    cam = scene.camera(600, 240)
    whole = scene.build(host.new_world(0)).render(cam, aux=True)
    monkeypatch.setenv("RT_AUX_MAX_TRIS", "700")
    g = scene.build(host.new_world(0)).render(cam, aux=True)
    monkeypatch.delenv("RT_AUX_MAX_TRIS")
    assert np.array_equal(g.rgb, whole.rgb) and np.array_equal(g.object_id, whole.object_id)
    r = scene.build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, "small dragons, split subtrees")


def test_color_at_random_rays(obj_files):
    """rt_color_at vs World::color_at / Intersections::hit of the oracle on arbitrary rays, including
    axis-parallel directions (|d| < 1e-4 slab rule, Q7) and non-unit directions."""
    scene = small_dragons(32, 24)
    rng = np.random.RandomState(7)
    n = 20000
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.2
    tgt = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:2000] *= rng.uniform(0.05, 20.0, size=(2000, 1)).astype(np.float32)
    d[2000:2300, 0] = 0.0
    d[2300:2600, 1] = 5e-5
    d[2600:2900, 2] = 0.0
    gw, ow = scene.build(host.new_world(0)), scene.build(_oracle().new_world())
    for remaining in (0, 4):
        grgb, gt, gid, guv = gw.color_at(o, d, remaining)
        rrgb, rt, rid, ruv = ow.color_at(o, d, remaining)
        assert np.array_equal(gid, rid)
        hit = rid >= 0
        assert hit.mean() > 0.2
        assert np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32))
        smooth = hit & (ruv[:, 0] != np.finfo(np.float32).max)  # (u, v) exist only for smooth triangles
        assert smooth.sum() > 100                                    # (intersections.rs:34-41: f32::MAX otherwise)
        assert np.array_equal(guv[smooth].view(np.uint32), ruv[smooth].view(np.uint32))
        dd = np.abs(to_u8(grgb).astype(int) - to_u8(rrgb).astype(int)).max(axis=1)
        assert (dd <= 1).mean() >= LSB_FRACTION


def test_kat_world_default_on_gpu():
    """world.rs:197-205 on the device: the default world, one ray that misses and one that hits."""
    w = host.new_world(0)
    w.add_point_light((-10, 10, -10), (1, 1, 1))
    w.add_shape(w.shape("sphere", (), B.Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)))
    w.add_shape(w.shape("sphere", [("scale", 0.5, 0.5, 0.5)]))
    rgb, t, ids, _ = w.color_at([(0, 0, -5), (0, 0, -5)], [(0, 1, 0), (0, 0, 1)], 1)
    assert np.array_equal(rgb[0], [0, 0, 0]) and ids[0] == -1 and np.isnan(t[0])
    assert np.all(np.abs(rgb[1] - np.array([0.38066, 0.47583, 0.2855])) < 1e-4) and ids[1] == 0 and t[1] == 4.0


# ---- the raw C ABI ------------------------------------------------------------------------------------
class Raw:
    def __init__(self, scene, size):
        import torch
        self.torch = torch
        self.lib = host.hip_lib()
        self.hw = scene.build(host.new_world(0))
        self.ctx, self.ds = self.hw.device_scene()
        self.cam = scene.camera(*size)
        self.rcam = self.hw.rt_camera(self.cam)
        self.dev = torch.device("cuda", 0)

    def render(self, flags=0, tile_h=1, n_parts=1, part=0, mode=abi.RT_JITTER_COUNTER, want_stats=True):
        torch = self.torch
        opts = abi.RtRenderOptions(mode, 0.5, 1, tile_h, n_parts, part, flags)
        rows = int(self.lib.rt_part_rows(C.byref(self.rcam), C.byref(opts)))
        out = torch.zeros((max(rows, 1), self.cam.hsize, 3), dtype=torch.float32, device=self.dev)
        oid = torch.full((max(rows, 1), self.cam.hsize), -7, dtype=torch.int32, device=self.dev)
        aux = abi.RtAux(None, oid.data_ptr(), None)
        st = abi.RtStats()
        torch.cuda.synchronize()  # torch filled the buffers on the null stream; the render runs on the context's stream
        rc = self.lib.rt_render_device(self.ctx, self.ds, C.byref(self.rcam), C.byref(opts), out.data_ptr(), C.byref(aux),
                                       None, C.byref(st) if want_stats else None)
        assert rc == abi.RT_OK, self.lib.rt_last_error(self.ctx).decode()
        torch.cuda.synchronize()
        return out[:rows].cpu().numpy(), oid[:rows].cpu().numpy(), st


@pytest.mark.parametrize("name,size", [("cover", (3840, 2160)), ("dragons", (3840, 2160))])
def test_parts_reassemble_at_4k(name, size):
    """BASELINE configs at full size: the frame rendered as 1, 2, 4 and 8 interleaved parts is the same
    bytes (what the 1/2/4/8-GPU runs gather), and rendering twice gives the same bytes."""
    from raytracer_amd import synth, tiles
    meshes = {"dragon.obj": synth.synth_dragon(220, 220)} if name == "dragons" else None
    scene = Scene.from_file(scene_path(name), meshes=meshes)
    raw = Raw(scene, size)
    full, ids, st = raw.render()
    again, _, _ = raw.render()
    assert np.array_equal(full.view(np.uint32), again.view(np.uint32))
    assert st.overflow == 0 and st.primary == size[0] * size[1]
    for n in (2, 4, 8):
        canvas = np.empty_like(full)
        idc = np.empty_like(ids)
        rays = 0
        for p in range(n):
            rgb, oid, s = raw.render(tile_h=2, n_parts=n, part=p)
            rows = tiles.part_rows(size[1], 2, n, p)
            canvas[rows], idc[rows] = rgb, oid
            rays += s.primary + s.shadow + s.reflect + s.refract
        assert np.array_equal(canvas.view(np.uint32), full.view(np.uint32)), f"{name}: {n} parts differ from 1"
        assert np.array_equal(idc, ids)
        assert rays == st.primary + st.shadow + st.reflect + st.refract


@pytest.mark.parametrize("name", ["dragons", "cover", "csg", "checkered_cube", "basic_scene", "space_teapot"])
def test_single_feature_kernels_change_nothing(name, obj_files, monkeypatch):
    """The launch picks the job-loop kernel compiled with the fewest features that cover the scene (rt_hip.hip
    jobs_kernel: groups / n1-n2 lists / csg / patterns, one each; the group-less ones at four waves per SIMD).  With
    RT_NO_LEAN in the environment every scene goes through the complete kernel: same bytes, same ids, same rays."""
    from raytracer_amd import synth
    meshes = {"dragon.obj": synth.synth_dragon(220, 220)} if name == "dragons" else None
    scene = Scene.from_file(scene_path(name), obj_files=obj_files, meshes=meshes)
    raw = Raw(scene, (2563, 1437))
    lean, lids, ls = raw.render()
    monkeypatch.setenv("RT_NO_LEAN", "1")
    full, fids, fs = raw.render()
    assert np.array_equal(lean.view(np.uint32), full.view(np.uint32)) and np.array_equal(lids, fids)
    assert (ls.primary, ls.shadow, ls.reflect, ls.refract, ls.overflow) == (fs.primary, fs.shadow, fs.reflect, fs.refract, 0)


def test_group_less_mixture_kernel_changes_nothing(tmp_path, monkeypatch):
    """A world without groups that needs several features at once (cover.yaml's glass with a checker pattern on its
    white material: the n1/n2 lists and patterns) renders through the instantiation with everything but the group walk,
    at four waves per SIMD: the same bytes as through the complete kernel."""
    text = open(scene_path("cover")).read()
    plain = "- define: white-material\n  material:\n    color: [ 1, 1, 1 ]"
    assert plain in text
    text = text.replace(plain, "- define: white-material\n  material:\n    pattern:\n      kind: checker\n      colors:\n"
                               "        - [1, 1, 1]\n        - [0.6, 0.6, 0.6]\n    color: [ 1, 1, 1 ]", 1)
    path = tmp_path / "mix.yaml"
    path.write_text(text)
    raw = Raw(Scene.from_file(str(path)), (1925, 1083))
    lean, lids, ls = raw.render()
    monkeypatch.setenv("RT_NO_LEAN", "1")
    full, fids, fs = raw.render()
    assert np.array_equal(lean.view(np.uint32), full.view(np.uint32)) and np.array_equal(lids, fids)
    assert (ls.primary, ls.shadow, ls.reflect, ls.refract, ls.overflow) == (fs.primary, fs.shadow, fs.reflect, fs.refract, 0)
    assert ls.refract > 0 and len(np.unique(lean.reshape(-1, 3), axis=0)) > 1000


def test_a_traversal_stack_that_runs_full_changes_nothing(monkeypatch):
    """The kernel for worlds of groups only keeps 20 of the 32 rows of the auxiliary traversal's stack (four blocks a CU);
    a ray whose traversal needs more is given up there and takes the reference's walk through the subtree (rt_device.h
    aux_coop, key 1).  RT_RENDER_SHORT_STACK leaves 5 rows, which many rays of this frame outgrow: same bytes and ids,
    through the lean kernel and the complete one, and the reference walk shows in the work counters."""
    from raytracer_amd import synth
    scene = Scene.from_file(scene_path("dragons"), meshes={"dragon.obj": synth.synth_dragon(120, 120)})
    raw = Raw(scene, (480, 270))
    full, ids, st = raw.render()
    short, sids, ss = raw.render(flags=abi.RT_RENDER_SHORT_STACK)
    assert np.array_equal(full.view(np.uint32), short.view(np.uint32)) and np.array_equal(ids, sids)
    assert (ss.primary, ss.shadow, ss.reflect, ss.refract, ss.overflow) == (st.primary, st.shadow, st.reflect, st.refract, 0)
    _, _, cw = raw.render(flags=abi.RT_RENDER_COUNT_WORK)
    _, _, cs = raw.render(flags=abi.RT_RENDER_COUNT_WORK | abi.RT_RENDER_SHORT_STACK)
    # nearly every traversal gives up after a few steps and the groups' own triangles are tested one by one instead
    assert 4 * cs.aux_tests < cw.aux_tests and 2 * cs.tri_tests > 3 * cw.tri_tests and cs.overflow == 0
    monkeypatch.setenv("RT_NO_LEAN", "1")
    cshort, cids, _ = raw.render(flags=abi.RT_RENDER_SHORT_STACK)
    assert np.array_equal(full.view(np.uint32), cshort.view(np.uint32)) and np.array_equal(ids, cids)


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (65, 9), (130, 67)])
def test_ragged_frames_match_oracle(size):
    """Frames that are not whole 8x8 tiles (the work queue hands out items outside the image) and ragged parts:
    same bars as the full scenes, and three interleaved parts of an odd height reassemble to the full frame."""
    from raytracer_amd import tiles
    scene = small_dragons(32, 24)
    cam = scene.camera(*size)
    g = scene.build(host.new_world(0)).render(cam, aux=True)
    r = scene.build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, f"small dragons {size}")
    raw = Raw(scene, size)
    full, ids, st = raw.render()
    assert np.array_equal(full.view(np.uint32), g.rgb.view(np.uint32))
    canvas, idc = np.empty_like(full), np.empty_like(ids)
    for p in range(3):
        rows = tiles.part_rows(size[1], 2, 3, p)
        rgb, oid, _ = raw.render(tile_h=2, n_parts=3, part=p)
        assert len(rgb) == len(rows)
        canvas[rows], idc[rows] = rgb, oid
    assert np.array_equal(canvas.view(np.uint32), full.view(np.uint32)) and np.array_equal(idc, ids)


@pytest.mark.parametrize("name,size", [("csg", (1925, 1083)), ("checkered_sphere", (3837, 2161)), ("basic_scene", (2563, 1437))])
def test_chunked_queue_draws_change_nothing(name, size):
    """Scenes of a few shapes draw several tiles from the work queue per atomic once a frame is large (launch_part's
    queue_chunk: 3 tiles at 1925x1083, 4 at 4K).  The same ragged frame rendered as 16 interleaved parts, each too
    small for chunking, is the same bytes, and every pixel is traced exactly once."""
    from raytracer_amd import tiles
    raw = Raw(Scene.from_file(scene_path(name)), size)
    full, ids, st = raw.render()
    assert st.overflow == 0 and st.primary == size[0] * size[1]
    canvas, idc = np.full_like(full, np.nan), np.full_like(ids, -9)
    primary = 0
    for p in range(16):
        rgb, oid, s = raw.render(tile_h=8, n_parts=16, part=p)
        rows = tiles.part_rows(size[1], 8, 16, p)
        canvas[rows], idc[rows] = rgb, oid
        primary += s.primary
    assert primary == st.primary
    assert np.array_equal(canvas.view(np.uint32), full.view(np.uint32)) and np.array_equal(idc, ids)


def _random_solids(world, rng, n_group, n_loose):
    """Cubes and spheres with anisotropic scales (0.02 .. 4), rotations, shears and translations: some in two groups
    (one of them transformed), the rest loose.  Enough shapes per list for the cull records to be built."""
    def solid(i):
        kind = "cube" if i % 2 else "sphere"
        ops = [("scale", *rng.uniform(0.02, 4.0, 3) ** rng.choice([1.0, 0.3])),
               ("rotate-x", rng.uniform(-3, 3)), ("rotate-z", rng.uniform(-3, 3)),
               ("shear", *(rng.uniform(-0.5, 0.5, 6) * (i % 3 == 0))),
               ("translate", *rng.uniform(-6, 6, 3))]
        mat = B.Material(color=tuple(rng.uniform(0, 1, 3)), reflective=float(i % 4 == 0) * 0.3,
                         transparency=float(i % 5 == 0) * 0.5, refractive_index=1.0 + 0.5 * (i % 5 == 0))
        return world.shape(kind, ops, mat, shadow=bool(i % 7))
    g1, g2 = world.group(()), world.group([("rotate-y", 0.7), ("scale", 1.5, 0.75, 1.2), ("translate", 1, -2, 0.5)])
    for i in range(n_group):
        world.group_add_shape(g1 if i % 2 else g2, solid(i))
    world.add_group(g1)
    world.add_group(g2)
    for i in range(n_loose):
        world.add_shape(solid(100 + i))
    world.add_point_light((-10, 12, -10), (1, 1, 1))


def test_shape_cull_spheres_change_nothing():
    """rt_accel.h shape_cull_record: rays the cull spheres reject are rays the reference's own tests reject.  Random
    cubes and spheres (thin, sheared, rotated; grouped and loose) against random rays, including rays aimed to graze
    the shapes, axis-parallel and non-unit directions: hit ids and t bit-identical to the oracle, which has no culling."""
    rng = np.random.RandomState(5)
    gw, ow = host.new_world(0), _oracle().new_world()
    _random_solids(gw, np.random.RandomState(9), 16, 14)
    _random_solids(ow, np.random.RandomState(9), 16, 14)
    acc = abi.scene_arrays(gw.flatten())
    assert len(acc["nodes"]) == 2 and acc["n_loose_shapes"] == 14
    n = 120000
    o = rng.uniform(-14, 14, size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(-7, 7, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10
    d[:k] *= rng.uniform(1e-3, 1e3, size=(k, 1)).astype(np.float32)
    d[k:2 * k, 0] = 0.0
    d[2 * k:3 * k, 1] = 3e-5
    d[3 * k:4 * k, 2] = 0.0
    o[4 * k:5 * k] *= 30.0
    d[4 * k:5 * k] = tgt[4 * k:5 * k] - o[4 * k:5 * k]
    d[4 * k:5 * k] /= np.linalg.norm(d[4 * k:5 * k], axis=1, keepdims=True)
    for remaining in (0, 3):
        grgb, gt, gid, _ = gw.color_at(o, d, remaining)
        rrgb, rt, rid, _ = ow.color_at(o, d, remaining)
        assert np.array_equal(gid, rid), f"{(gid != rid).sum()} hit ids differ"
        hit = rid >= 0
        assert hit.mean() > 0.3
        assert np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32))
        both_nan = np.isnan(grgb).any(axis=1) & np.isnan(rrgb).any(axis=1)
        dd = np.abs(to_u8(np.nan_to_num(grgb)).astype(int) - to_u8(np.nan_to_num(rrgb)).astype(int)).max(axis=1)
        assert ((dd <= 1) | both_nan).mean() >= LSB_FRACTION


@pytest.mark.parametrize("name", ["cover", "checkered_cube"])
def test_shape_culling_changes_nothing_at_4k(name):
    """cover.yaml at 4K (BASELINE config 2; 18 loose solids => cull records): bit-identical with RT_RENDER_NO_PRUNE."""
    raw = Raw(Scene.from_file(scene_path(name)), (3840, 2160))
    base, ids, _ = raw.render()
    rgb, oid, _ = raw.render(flags=abi.RT_RENDER_NO_PRUNE)
    assert np.array_equal(rgb.view(np.uint32), base.view(np.uint32)) and np.array_equal(oid, ids)


def test_accel_and_pruning_change_nothing_at_4k():
    """dragons at 4K (BASELINE config 5, synthetic mesh): the auxiliary BVHs and the distance pruning are
    pure culling — bit-identical image and ids with either or both switched off."""
    from raytracer_amd import synth
    scene = Scene.from_file(scene_path("dragons"), meshes={"dragon.obj": synth.synth_dragon(330, 330)})
    raw = Raw(scene, (3840, 2160))
    base, ids, st = raw.render(flags=abi.RT_RENDER_COUNT_WORK)
    assert st.tri_tests > 0 and st.node_visits > 0
    for flags in (abi.RT_RENDER_NO_ACCEL, abi.RT_RENDER_NO_PRUNE, abi.RT_RENDER_NO_ACCEL | abi.RT_RENDER_NO_PRUNE):
        rgb, oid, s = raw.render(flags=flags)
        assert np.array_equal(rgb.view(np.uint32), base.view(np.uint32)), f"flags={flags}"
        assert np.array_equal(oid, ids)


def test_bench_workload_full_mesh_at_4k():
    """The workload bench.py times, itself: dragons.yaml at 3840x2160 over the full synthetic mesh (871 200 triangles
    x 6; deeper auxiliary trees and longer stacks than the smaller stand-ins above).  The frame is bit-identical, ids
    included, with the auxiliary BVHs and the distance pruning off (the reference's linear walk, groups.rs:118-120),
    nothing overflows, and 20 000 random rays through the same world give the oracle's hits bit for bit."""
    from raytracer_amd import synth
    scene = Scene.from_file(scene_path("dragons"), meshes={"dragon.obj": synth.synth_dragon()})
    raw = Raw(scene, (3840, 2160))
    base, ids, st = raw.render()
    assert st.overflow == 0 and st.primary == 3840 * 2160
    assert st.primary + st.shadow + st.reflect + st.refract > 12_000_000
    rgb, oid, s2 = raw.render(flags=abi.RT_RENDER_NO_ACCEL | abi.RT_RENDER_NO_PRUNE)
    assert s2.overflow == 0
    assert np.array_equal(rgb.view(np.uint32), base.view(np.uint32)) and np.array_equal(oid, ids)
    assert (st.shadow, st.reflect, st.refract) == (s2.shadow, s2.reflect, s2.refract)
    del rgb, oid
    rng = np.random.RandomState(11)
    n = 20000
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.2
    tgt = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    tgt[:, 1] = np.abs(tgt[:, 1])
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:1000] *= rng.uniform(0.05, 20.0, size=(1000, 1)).astype(np.float32)
    d[1000:1200, 0] = 0.0
    d[1200:1400, 1] = 5e-5
    grgb, gt, gid, guv = raw.hw.color_at(o, d, 4)
    ow = scene.build(_oracle().new_world())
    rrgb, rt, rid, ruv = ow.color_at(o, d, 4)
    assert np.array_equal(gid, rid), f"{(gid != rid).sum()} hit ids differ"
    hit = rid >= 0
    assert hit.mean() > 0.2
    assert np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32))
    smooth = hit & (ruv[:, 0] != np.finfo(np.float32).max)
    assert smooth.sum() > 200 and np.array_equal(guv[smooth].view(np.uint32), ruv[smooth].view(np.uint32))
    dd = np.abs(to_u8(grgb).astype(int) - to_u8(rrgb).astype(int)).max(axis=1)
    assert (dd <= 1).mean() >= LSB_FRACTION


@pytest.mark.parametrize("mode", [abi.RT_JITTER_CONST, abi.RT_JITTER_COUNTER])
def test_soft_shadows_parts_through_direct_kernel(mode):
    """soft_shadows.yaml (BASELINE config 4: area light, 100 samples, "tiled over 8 GPUs") at 1920x1080 through
    k_render_direct, whose row mapping only this test reaches with n_parts > 1: the frame as 8 interleaved parts of
    8-row and of 2-row stripes is the same bytes as one part, and a 1/6-size frame of it equals the oracle."""
    from raytracer_amd import tiles
    scene = Scene.from_file(scene_path("soft_shadows"))
    size = (1920, 1080)
    raw = Raw(scene, size)
    full, ids, st = raw.render(mode=mode)
    assert st.overflow == 0 and st.primary == size[0] * size[1] and st.shadow > 50 * st.primary
    for tile_h in (8, 2):
        canvas, idc, rays = np.full_like(full, np.nan), np.full_like(ids, -9), 0
        for p in range(8):
            rgb, oid, s = raw.render(tile_h=tile_h, n_parts=8, part=p, mode=mode)
            rows = tiles.part_rows(size[1], tile_h, 8, p)
            assert len(rgb) == len(rows)
            canvas[rows], idc[rows] = rgb, oid
            rays += s.primary + s.shadow + s.reflect + s.refract
        assert np.array_equal(canvas.view(np.uint32), full.view(np.uint32)), f"tile_h {tile_h}: 8 parts differ from 1"
        assert np.array_equal(idc, ids)
        assert rays == st.primary + st.shadow + st.reflect + st.refract
    # ragged: 3 parts of 5-row stripes over a height that is no multiple of anything
    small = Raw(scene, (333, 187))
    sfull, sids, _ = small.render(mode=mode)
    canvas, idc = np.full_like(sfull, np.nan), np.full_like(sids, -9)
    for p in range(3):
        rgb, oid, _ = small.render(tile_h=5, n_parts=3, part=p, mode=mode)
        rows = tiles.part_rows(187, 5, 3, p)
        canvas[rows], idc[rows] = rgb, oid
    assert np.array_equal(canvas.view(np.uint32), sfull.view(np.uint32)) and np.array_equal(idc, sids)


def test_soft_shadows_at_4k_in_parts_and_through_both_direct_kernels(monkeypatch):
    """BASELINE config 4 at its full size: soft_shadows.yaml at 3840x2160, 100 shadow samples a hit, "tiled over 8
    GPUs".  The eight interleaved parts reassemble to the bytes of the whole frame, and the whole frame is the same
    bytes through the direct kernel's two instantiations: shadow rays that keep their origin in every shape's space and
    test the shadow-casting shapes first (rt_device.h hoisted_is_shadowed), and the plain World::is_shadowed loop
    (RT_NO_HOIST in the environment at launch time)."""
    from raytracer_amd import tiles
    scene = Scene.from_file(scene_path("soft_shadows"))
    size = (3840, 2160)
    raw = Raw(scene, size)
    full, ids, st = raw.render()
    assert st.overflow == 0 and st.primary == size[0] * size[1] and st.shadow > 50 * st.primary
    canvas, idc, rays = np.full_like(full, np.nan), np.full_like(ids, -9), 0
    for p in range(8):
        rgb, oid, s = raw.render(tile_h=8, n_parts=8, part=p)
        rows = tiles.part_rows(size[1], 8, 8, p)
        canvas[rows], idc[rows] = rgb, oid
        rays += s.primary + s.shadow + s.reflect + s.refract
    assert np.array_equal(canvas.view(np.uint32), full.view(np.uint32)) and np.array_equal(idc, ids)
    assert rays == st.primary + st.shadow + st.reflect + st.refract
    del canvas, idc
    monkeypatch.setenv("RT_NO_HOIST", "1")
    plain, pids, ps = raw.render()
    assert np.array_equal(plain.view(np.uint32), full.view(np.uint32)) and np.array_equal(pids, ids)
    assert (ps.primary, ps.shadow, ps.reflect, ps.refract) == (st.primary, st.shadow, st.reflect, st.refract)


@pytest.mark.parametrize("seed", [1, 3, 6])
def test_fuzz_parts(seed, root):
    """tools/gpu_fuzz.py run_parts: a random world (odd seeds: group-less with an area light, the direct kernel) as one
    part and as n interleaved parts of a random stripe height gives the same bytes."""
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import gpu_fuzz
    lines = []
    assert gpu_fuzz.run_parts(seed, log=lines.append) == 0, "\n".join(lines)


@pytest.mark.parametrize("seed", [0, 1, 2, 5, 8])
def test_fuzz_few_shapes_under_area_lights(seed, root):
    """tools/gpu_fuzz.py run_few: one to four loose shapes of any kind, some shadow-less, under a random area light --
    the worlds that go through the direct kernel's HOIST instantiation -- against the oracle."""
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import gpu_fuzz
    lines = []
    assert gpu_fuzz.run_few(seed, log=lines.append) == 0, "\n".join(lines)


def test_two_streams_in_flight():
    """Two frames rendered back to back on two different streams, neither synchronised in between: each launch has
    its own work queue and counter block (the ring in the context), so both canvases equal the synchronous render."""
    import torch
    scene = small_dragons(48, 36)
    raw = Raw(scene, (1280, 720))
    base, _, st = raw.render()
    opts = abi.RtRenderOptions(abi.RT_JITTER_COUNTER, 0.5, 1, 1, 1, 0, 0)
    outs = [torch.zeros((720, 1280, 3), dtype=torch.float32, device=raw.dev) for _ in range(12)]
    streams = [torch.cuda.Stream(raw.dev) for _ in range(3)]
    torch.cuda.synchronize()
    for k, out in enumerate(outs):  # 12 launches over 3 streams: the ring of 8 blocks wraps
        rc = raw.lib.rt_render_device(raw.ctx, raw.ds, C.byref(raw.rcam), C.byref(opts), out.data_ptr(), None,
                                      C.c_void_p(streams[k % 3].cuda_stream), None)
        assert rc == abi.RT_OK
    assert raw.lib.rt_check(raw.ctx, None) == abi.RT_OK
    for out in outs:
        assert np.array_equal(out.cpu().numpy().view(np.uint32), base.view(np.uint32))


class _GlassStack:
    """n parallel glass panes (planes facing the camera, refractive index 1.5), the camera between two of them near the
    far end: the n1/n2 walk of a hit (intersections.rs:141-160) lists every pane behind the eye as well."""

    def __init__(self, n, area_light=False):
        self.n = n
        self.area_light = area_light

    def build(self, w):
        glass = B.Material(color=(0.9, 0.9, 1.0), diffuse=0.1, transparency=0.9, reflective=0.1, refractive_index=1.5)
        for k in range(self.n):
            w.add_shape(w.shape("plane", [("rotate-x", np.pi / 2), ("translate", 0, 0, 0.01 * k)], glass))
        if self.area_light:  # group-less with an area light: the direct kernel (rt_hip.hip launch_part)
            w.add_area_light((-1, 5, -10), (2, 0, 0), 2, (0, 2, 0), 2, (1, 1, 1))
        else:
            w.add_point_light((0, 5, -10), (1, 1, 1))
        return w

    def camera(self, hsize, vsize):
        return B.Camera(hsize, vsize, 0.8, (0, 0, 0.01 * (self.n - 20) + 0.005), (0, 0, 100.0), (0, 1, 0))


def test_overflow_is_never_silent(monkeypatch):
    """The reference's intersection list is unbounded (intersections.rs:66-73); the device keeps RT_XS_CAP entries per
    ray and, for a ray that passes them, continues in side lists in device memory that the context allocates when the
    first such ray shows up.  A synchronous call does that by itself and runs again: it returns the right image.  An
    asynchronous call (stats == NULL: bench loop, multi-GPU callers) cannot: its overflow stays in the context, where the
    next synchronising call or rt_check finds it, once -- never an image that only looks valid."""
    import torch
    scene = _GlassStack(140)
    want = scene.build(_oracle().new_world()).render(scene.camera(64, 48), B.RenderOptions(jitter_mode=B.JITTER_CONST), aux=True)
    opts = abi.RtRenderOptions(abi.RT_JITTER_CONST, 0.5, 0, 1, 1, 0, 0)

    def same(rgb):
        d = np.abs(to_u8(rgb).astype(int) - to_u8(want.rgb).astype(int)).max(axis=2)
        return float((d <= 1).mean()) >= LSB_FRACTION

    # asynchronous first: the overflow is reported by rt_check, once; the frame rendered again is right
    raw = Raw(scene, (64, 48))
    out = torch.zeros((48, 64, 3), dtype=torch.float32, device=raw.dev)
    torch.cuda.synchronize()
    rc = raw.lib.rt_render_device(raw.ctx, raw.ds, C.byref(raw.rcam), C.byref(opts), out.data_ptr(), None, None, None)
    assert rc == abi.RT_OK  # nothing known yet
    assert raw.lib.rt_check(raw.ctx, None) == abi.RT_E_OVERFLOW and b"overflow" in raw.lib.rt_last_error(raw.ctx)
    assert raw.lib.rt_check(raw.ctx, None) == abi.RT_OK  # reported once
    for _ in range(2):  # two launches in flight on the side lists: ordered one after the other
        rc = raw.lib.rt_render_device(raw.ctx, raw.ds, C.byref(raw.rcam), C.byref(opts), out.data_ptr(), None, None, None)
        assert rc == abi.RT_OK
    assert raw.lib.rt_check(raw.ctx, None) == abi.RT_OK
    assert same(out.cpu().numpy())

    # synchronous first, on a context of its own: the call allocates the side lists and renders again by itself
    raw2 = Raw(scene, (64, 48))
    rgb, ids, st = raw2.render(mode=abi.RT_JITTER_CONST)
    assert st.overflow == 0 and same(rgb) and np.array_equal(ids, want.object_id)
    host_out = np.zeros((48, 64, 3), dtype=np.float32)
    raw3 = Raw(scene, (64, 48))
    rc = raw3.lib.rt_render(raw3.ctx, raw3.ds, C.byref(raw3.rcam), C.byref(opts), host_out.ctypes.data, None, None)
    assert rc == abi.RT_OK and same(host_out)

    # the same through the direct kernel (group-less world, area light), which has no side lists of its own: the call
    # that overflows there runs again in the persistent kernel
    area = _GlassStack(140, area_light=True)
    want_a = area.build(_oracle().new_world()).render(area.camera(64, 48), B.RenderOptions(jitter_mode=B.JITTER_CONST), aux=True)
    raw_a = Raw(area, (64, 48))
    rgb, ids, st = raw_a.render(mode=abi.RT_JITTER_CONST)
    d = np.abs(to_u8(rgb).astype(int) - to_u8(want_a.rgb).astype(int)).max(axis=2)
    assert st.overflow == 0 and float((d <= 1).mean()) >= LSB_FRACTION and np.array_equal(ids, want_a.object_id)

    # an earlier asynchronous overflow is reported by the next synchronous call even though that call itself succeeds
    raw4 = Raw(scene, (64, 48))
    rc = raw4.lib.rt_render_device(raw4.ctx, raw4.ds, C.byref(raw4.rcam), C.byref(opts), out.data_ptr(), None, None, None)
    assert rc == abi.RT_OK
    rc = raw4.lib.rt_render(raw4.ctx, raw4.ds, C.byref(raw4.rcam), C.byref(opts), host_out.ctypes.data, None, None)
    assert rc == abi.RT_E_OVERFLOW and b"earlier" in raw4.lib.rt_last_error(raw4.ctx)
    assert same(host_out)  # the synchronous call's own frame is the right one
    assert raw4.lib.rt_check(raw4.ctx, None) == abi.RT_OK

    # the entry point that may not allocate reports the overflow instead (rt_color_at_static); once rt_color_at has
    # given the context its side lists, it works
    raw5 = Raw(scene, (64, 48))
    o = np.tile(np.array([[0.0, 0.0, 0.01 * 120 + 0.005]], dtype=np.float32), (64, 1))
    d = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (64, 1))

    def static(n_expected_rc):
        rgb = np.full((64, 3), np.nan, dtype=np.float32)
        ids = np.full(64, -77, dtype=np.int32)
        copts = abi.RtRenderOptions(abi.RT_JITTER_CONST, 0.5, 0, 1, 1, 0, 0)
        rc = raw5.lib.rt_color_at_static(raw5.ctx, raw5.ds, C.byref(copts), o.ctypes.data, d.ctypes.data, 64, 4,
                                         rgb.ctypes.data, None, ids.ctypes.data, None, None)
        assert rc == n_expected_rc, raw5.lib.rt_last_error(raw5.ctx)
        return rgb, ids

    static(abi.RT_E_OVERFLOW)
    assert b"may not allocate" in raw5.lib.rt_last_error(raw5.ctx)
    rgb_a, t_a, id_a, _ = raw5.hw.color_at(o, d, 4, B.RenderOptions(jitter_mode=B.JITTER_CONST))
    rgb_s, id_s = static(abi.RT_OK)
    assert np.array_equal(rgb_a.view(np.uint32), rgb_s.view(np.uint32)) and np.array_equal(id_a, id_s)
    ow = scene.build(_oracle().new_world())
    rgb_r, t_r, id_r, _ = ow.color_at(o, d, 4, B.RenderOptions(jitter_mode=B.JITTER_CONST))
    assert np.array_equal(id_a, id_r) and np.array_equal(t_a.view(np.uint32), t_r.view(np.uint32))
    assert np.abs(to_u8(rgb_a).astype(int) - to_u8(rgb_r).astype(int)).max() <= 1

    # the documented hard limit: with side lists capped below what the scene needs the failure is loud
    monkeypatch.setenv("RT_SIDE_MAX", "100")
    raw6 = Raw(scene, (64, 48))
    rc = raw6.lib.rt_render(raw6.ctx, raw6.ds, C.byref(raw6.rcam), C.byref(opts), host_out.ctypes.data, None, None)
    assert rc == abi.RT_E_OVERFLOW and b"RT_SIDE_MAX" in raw6.lib.rt_last_error(raw6.ctx)
    monkeypatch.delenv("RT_SIDE_MAX")
    ok = Raw(_GlassStack(30), (64, 48))  # another scene, another context: never needed any of it
    _, _, st2 = ok.render(mode=abi.RT_JITTER_CONST)
    assert st2.overflow == 0


def test_transparent_triangle_soup_matches_oracle():
    """400 glass triangles in a divided group between the camera and a wall: most rays cross dozens of them, refracted
    rays list every crossing for the n1/n2 walk (intersections.rs:141-160), many more than RT_XS_CAP for the rays that
    run along the pile.  The frame equals the oracle's: ids and t bit for bit, colours within the suite's bar."""
    rng = np.random.RandomState(404)

    def build(w):
        glass = B.Material(color=(0.8, 0.9, 1.0), diffuse=0.2, ambient=0.1, transparency=0.85, reflective=0.1,
                           refractive_index=1.3)
        g = w.group([("rotate-y", 0.3)])
        for k in range(400):
            c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), 0.03 * k])
            p = [tuple(c + rng.uniform(-1.2, 1.2, 3) * (1, 1, 0.02)) for _ in range(3)]
            w.group_add_shape(g, w.shape("triangle", (), glass, params=[x for q in p for x in q]))
        w.group_divide(g, 8)
        w.add_group(g)
        w.add_shape(w.shape("plane", [("rotate-x", np.pi / 2), ("translate", 0, 0, 20)], B.Material(color=(1, 0.8, 0.6))))
        w.add_point_light((-5, 8, -10), (1, 1, 1))
        return w

    cam = B.Camera(96, 64, 0.9, (0.2, 0.1, -3.0), (0, 0, 6.0), (0, 1, 0))
    ro = B.RenderOptions(jitter_mode=B.JITTER_CONST)
    rng = np.random.RandomState(404)
    g = build(host.new_world(0)).render(cam, ro, aux=True)
    rng = np.random.RandomState(404)
    r = build(_oracle().new_world()).render(cam, ro, aux=True)
    assert r.stats["refract"] > 1000
    check(g, r, "triangle soup")
    # and along the pile, where one ray meets nearly all 400: rt_color_at against the oracle
    o = np.zeros((256, 3), dtype=np.float32)
    o[:, 0] = np.linspace(-1.0, 1.0, 256)
    o[:, 2] = -2.0
    d = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (256, 1))
    d = d @ np.array([[np.cos(0.3), 0, -np.sin(0.3)], [0, 1, 0], [np.sin(0.3), 0, np.cos(0.3)]], dtype=np.float32).T
    rng = np.random.RandomState(404)
    gw = build(host.new_world(0))
    rng = np.random.RandomState(404)
    ow = build(_oracle().new_world())
    grgb, gt, gid, _ = gw.color_at(o, d, 4)
    rrgb, rt_, rid, _ = ow.color_at(o, d, 4)
    assert np.array_equal(gid, rid) and np.array_equal(gt[rid >= 0].view(np.uint32), rt_[rid >= 0].view(np.uint32))
    assert np.abs(to_u8(np.nan_to_num(grgb)).astype(int) - to_u8(np.nan_to_num(rrgb)).astype(int)).max() <= 1


def test_abi_error_paths():
    lib = host.hip_lib()
    scene = Scene.from_file(scene_path("basic_scene"))
    raw = Raw(scene, (16, 16))
    # part >= n_parts
    opts = abi.RtRenderOptions(0, 0.5, 0, 1, 2, 2, 0)
    out = raw.torch.zeros((16, 16, 3), dtype=raw.torch.float32, device=raw.dev)
    assert lib.rt_render_device(raw.ctx, raw.ds, C.byref(raw.rcam), C.byref(opts), out.data_ptr(), None, None, None) == abi.RT_E_ABI
    # NULL arguments
    assert lib.rt_render_device(raw.ctx, raw.ds, None, None, out.data_ptr(), None, None, None) == abi.RT_E_ABI
    # empty camera
    cam0 = abi.RtCamera(0, 16, 0.01, 1, 1)
    assert lib.rt_render_device(raw.ctx, raw.ds, C.byref(cam0), None, out.data_ptr(), None, None, None) == abi.RT_E_ABI
    # a frame wider than a lane's pixel token (row << 16 | column) is refused before anything is launched or written
    wide = abi.RtCamera(70000, 16, 0.01, 1, 1)
    assert lib.rt_render_device(raw.ctx, raw.ds, C.byref(wide), None, out.data_ptr(), None, None, None) == abi.RT_E_ABI
    assert b"too large" in lib.rt_last_error(raw.ctx)
    # a scene with an out-of-range index is rejected before anything reaches the GPU
    fs = raw.hw.flatten()
    arr = abi.scene_arrays(fs)
    shapes = arr["shapes"].copy()
    shapes["material"][0] = 10 ** 6
    bad = abi.RtScene.from_buffer_copy(fs)
    bad.shapes = shapes.ctypes.data
    ds = C.c_void_p()
    assert lib.rt_scene_create(raw.ctx, C.byref(bad), C.byref(ds)) == abi.RT_E_SCENE
    assert b"material" in lib.rt_last_error(raw.ctx)
    bad2 = abi.RtScene.from_buffer_copy(fs)
    bad2.abi_version = 99
    assert lib.rt_scene_create(raw.ctx, C.byref(bad2), C.byref(ds)) == abi.RT_E_ABI
    # a csg tree that could give one ray more intersections than even the slow-path kernels' csg list holds
    # (RT_BIG_CSG_CAP = 256: seventy cylinders, four each) is refused when the scene is created -- never discovered ray
    # by ray (trees between the two capacities render: test_worlds_beyond_the_ordinary_kernels_capacities)
    w = host.new_world(0)
    cyl = [w.shape("cylinder", [("translate", 0.03 * k, 0, 0)], None, params=[-1.0, 1.0, 1.0]) for k in range(70)]
    tree = w.csg(B.CSG_UNION, B.CHILD_SHAPE, cyl[0], B.CHILD_SHAPE, cyl[1])
    for k in range(2, 70):
        tree = w.csg(B.CSG_UNION, B.CHILD_CSG, tree, B.CHILD_SHAPE, cyl[k])
    w.add_csg(tree)
    w.add_point_light((0, 5, -5), (1, 1, 1))
    with pytest.raises(RuntimeError, match="RT_BIG_CSG_CAP"):
        w.device_scene()
    w = host.new_world(0)
    w.add_shape(w.shape("sphere", ()))
    w.set_recursion_limit(17)
    with pytest.raises(RuntimeError, match="recursion_limit"):
        w.device_scene()
    # a part with no rows renders nothing and succeeds (ragged: 3 rows over 8 parts)
    scene3 = Raw(scene, (8, 3))
    rgb, _, st = scene3.render(tile_h=1, n_parts=8, part=5)
    assert rgb.shape[0] == 0 and st.primary == 0


def _deep_groups(w):
    """a sphere and a cube under ten nested transformed groups (RT_MAX_XF_DEPTH is 6), patterned so that
    world_to_object's chain shows in the colours"""
    pat = w.pattern(B.PAT_CHECKER, [(0.9, 0.2, 0.2), (0.2, 0.2, 0.9)], [("scale", 0.4, 0.4, 0.4)])
    inner = w.group([("rotate-y", 0.3)])
    w.group_add_shape(inner, w.shape("sphere", [("translate", 0.5, 0, 0)], B.Material(pattern=pat, reflective=0.2)))
    w.group_add_shape(inner, w.shape("cube", [("scale", 0.4, 0.4, 0.4), ("translate", -1.2, 0.2, 0)], B.Material(color=(0.3, 0.8, 0.3))))
    g = inner
    for k in range(9):
        outer = w.group([("rotate-z", 0.1 * (k + 1)), ("scale", 1.05, 0.95, 1.0), ("translate", 0.05 * k, -0.03 * k, 0.02)])
        w.group_add_child(outer, g)
        g = outer
    w.add_group(g)
    w.add_shape(w.shape("plane", [("translate", 0, -2, 0)], B.Material(color=(0.8, 0.8, 0.7))))
    w.add_point_light((-5, 8, -8), (1, 1, 1))


def _csg_comb(w, levels):
    """T_k = (a_k u b_k) op T_{k-1}: every level keeps a finished list waiting while the rest is evaluated, so the
    evaluation stack grows with the levels (RT_CSG_DEPTH is 8) and so does the list (RT_CSG_CAP is 32 entries)"""
    def ball(k, j):
        return w.shape("sphere", [("scale", 0.5, 0.5, 0.5), ("translate", 0.35 * k - 2.0, 0.3 * j, 0.1 * k)],
                       B.Material(color=(0.2 + 0.06 * k, 0.9 - 0.06 * k, 0.3 + 0.3 * j)))
    tree = w.csg(B.CSG_UNION, B.CHILD_SHAPE, ball(0, 0), B.CHILD_SHAPE, ball(0, 1))
    for k in range(1, levels):
        pair = w.csg(B.CSG_UNION, B.CHILD_SHAPE, ball(k, 0), B.CHILD_SHAPE, ball(k, 1))
        tree = w.csg([B.CSG_UNION, B.CSG_DIFFERENCE, B.CSG_UNION][k % 3], B.CHILD_CSG, pair, B.CHILD_CSG, tree)
    w.add_csg(tree)
    w.add_point_light((-5, 8, -8), (1, 1, 1))


def _nine_cylinders(w):
    """nine closed cylinders in one csg: up to 36 intersections of one ray with its leaves"""
    cyl = [w.shape("cylinder", [("rotate-z", 1.2), ("translate", 0.3 * k - 1.2, 0.1 * k, 0)], B.Material(color=(0.1 * k, 0.5, 1 - 0.1 * k)),
                   params=[-1.0, 1.0, 1.0]) for k in range(9)]
    tree = w.csg(B.CSG_UNION, B.CHILD_SHAPE, cyl[0], B.CHILD_SHAPE, cyl[1])
    for k in range(2, 9):
        tree = w.csg(B.CSG_UNION if k % 2 else B.CSG_DIFFERENCE, B.CHILD_CSG, tree, B.CHILD_SHAPE, cyl[k])
    w.add_csg(tree)
    w.add_point_light((0, 5, -5), (1, 1, 1))


def _mirror_hall(w):
    """two facing mirrors and a glass ball, recursion_limit 12 (the ordinary kernels' ray stack holds 8 pending rays)"""
    mirror = B.Material(color=(0.05, 0.05, 0.08), reflective=0.9, diffuse=0.2)
    w.add_shape(w.shape("plane", [("rotate-x", np.pi / 2), ("translate", 0, 0, 4)], mirror))
    w.add_shape(w.shape("plane", [("rotate-x", np.pi / 2), ("translate", 0, 0, -6)], mirror))
    w.add_shape(w.shape("plane", [("translate", 0, -1, 0)], B.Material(color=(0.7, 0.6, 0.5))))
    w.add_shape(w.shape("sphere", [("translate", 0.4, 0, 0)], B.Material(color=(0.8, 0.1, 0.1), transparency=0.6, reflective=0.3,
                                                                          refractive_index=1.5, diffuse=0.3)))
    w.set_recursion_limit(12)
    w.add_point_light((-3, 6, -4), (1, 1, 1))


@pytest.mark.parametrize("build", [_deep_groups, _csg_comb, _nine_cylinders, _mirror_hall])
def test_worlds_beyond_the_ordinary_kernels_capacities(build):
    """Where the reference is unbounded the ordinary kernels have fixed capacities (rt_device.h Lim): six nested group
    transforms, a csg evaluation stack of 8 and list of 32, 8 pending rays.  A world that passes one renders through the
    slow-path instantiation (32 / 32 / 256 / 17) instead of being refused, and equals the oracle like any other."""
    def make(w):
        if build is _csg_comb:
            build(w, 12)
        else:
            build(w)
        return w
    cam = B.Camera(200, 120, 0.9, (0.5, 1.5, -6), (0, 0, 0), (0, 1, 0))
    gw = make(host.new_world(0))
    g = gw.render(cam, aux=True)
    r = make(_oracle().new_world()).render(cam, aux=True)
    check(g, r, build.__name__)
    assert r.hit_mask.mean() > 0.02
    # ... and through rt_color_at, which shares the context (and so the kernels) with the frame above
    rng = np.random.RandomState(3)
    o = rng.uniform(-3, 3, size=(5000, 3)).astype(np.float32) + np.array([0, 1, -6], dtype=np.float32)
    d = -o + rng.uniform(-1.5, 1.5, size=(5000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    remaining = 12 if build is _mirror_hall else 4
    grgb, gt, gid, _ = gw.color_at(o, d, remaining)
    rrgb, rt, rid, _ = make(_oracle().new_world()).color_at(o, d, remaining)
    assert np.array_equal(gid, rid)
    hit = rid >= 0
    assert np.array_equal(gt[hit].view(np.uint32), rt[hit].view(np.uint32))
    dd = np.abs(to_u8(grgb).astype(int) - to_u8(rrgb).astype(int)).max(axis=1)
    assert (dd <= 1).mean() >= LSB_FRACTION


def test_color_at_deeper_than_the_ordinary_ray_stack():
    """rt_color_at with remaining = 10 on a world that fits the ordinary kernels: the call moves the context to the
    slow-path kernels (17 pending rays) and equals the oracle."""
    def world(w):
        _mirror_hall(w)
        w.set_recursion_limit(4)
        return w
    rng = np.random.RandomState(4)
    o = rng.uniform(-2, 2, size=(3000, 3)).astype(np.float32)
    d = rng.normal(size=(3000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gw, ow = world(host.new_world(0)), world(_oracle().new_world())
    for remaining in (7, 10, 16):
        grgb, gt, gid, _ = gw.color_at(o, d, remaining)
        rrgb, rt, rid, _ = ow.color_at(o, d, remaining)
        assert np.array_equal(gid, rid) and np.array_equal(gt[rid >= 0].view(np.uint32), rt[rid >= 0].view(np.uint32))
        dd = np.abs(to_u8(grgb).astype(int) - to_u8(rrgb).astype(int)).max(axis=1)
        assert (dd <= 1).mean() >= LSB_FRACTION, remaining
    with pytest.raises(RuntimeError, match="ray stack"):
        gw.color_at(o[:4], d[:4], 17)


def test_world_without_lights_is_black():
    """world.rs:66-86: no light, no colour (Q4) — and no secondary rays are spawned."""
    w = host.new_world(0)
    w.add_shape(w.shape("sphere", (), B.Material(reflective=0.5)))
    rgb, t, ids, _ = w.color_at([(0, 0, -5)], [(0, 0, 1)], 4)
    assert np.array_equal(rgb[0], [0, 0, 0]) and ids[0] == 0 and t[0] == 4.0


@pytest.mark.parametrize("seed", [29, 49, 65, 73, 77, 153, 2, 7])
def test_fuzz_regressions(seed, root):
    """Worlds of tools/gpu_fuzz.py that once differed from the oracle, and two that never did.  29, 73, 77, 153: a cone
    inside a group was culled by distance, but the reference's cone is not inside its own box (a hit without the
    min < y < max test for rays parallel to the side, cone.rs:131-141; caps of radius sqrt|y|, cone.rs:90-94).
    49, 65: a mesh listing a triangle twice; of entries equal in (t, Shape, u, v) the reference keeps the later one."""
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import gpu_fuzz
    lines = []
    assert gpu_fuzz.run_seed(seed, log=lines.append) == 0, "\n".join(lines)
    assert len(lines) == 2 and not any("skipped" in ln for ln in lines), "\n".join(lines)


@pytest.mark.parametrize("seed", [0, 1, 4, 7])
def test_fuzz_frames(seed, root):
    """Whole frames of tools/gpu_fuzz.py's random worlds (every primitive and pattern kind, texture and cube maps,
    groups, a divided triangle soup, CSG; group-less with an area light for odd seeds), random cameras, odd sizes."""
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import gpu_fuzz
    lines = []
    assert gpu_fuzz.run_frame(seed, log=lines.append) == 0 and ": ok" in lines[-1], "\n".join(lines)


def test_gather_pipeline_over_rccl(root):
    """bench.py's N > 1 exchange step on one GPU: tiles.CanvasPipeline over RCCL (world size 1, in a process of its
    own), renders launched on torch's stream.  Every assembled canvas is the frame rendered synchronously."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_gather_rehearsal.py"), "5"], capture_output=True,
                       text=True, timeout=240)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("frame ")]
    assert r.returncode == 0 and len(lines) == 5 and all("mismatching pixels 0," in ln for ln in lines), r.stdout + r.stderr[-2000:]


def test_triangles_holding_a_nan_are_nobodys_duplicates():
    """A shape that holds a NaN equals no shape, itself included (derived PartialEq over f32), so two smooth triangles
    with the same corners and the same NaN normals are two different shapes: a ray that hits both at the same t reports
    the FIRST (stable order of equal t), not "the last of its value-identical duplicates" that the device
    reports for genuine duplicates (Q2).  Genuine duplicates next to them, for contrast."""
    nan = float("nan")
    tri = [-1, 0, 0, 1, 0, 0, 0, 1.5, 0]

    def build(w):
        g = w.group([("translate", 0, 0, 0.5)])
        for _ in range(2):  # same corners, NaN normals: never equal
            w.group_add_shape(g, w.shape("smooth-triangle", (), None, params=tri + [0, 0, -1, nan, 0, -1, 0, 0, -1]))
        g2 = w.group([("translate", 3, 0, 0.5)])
        for _ in range(2):  # genuine duplicates
            w.group_add_shape(g2, w.shape("smooth-triangle", (), None, params=tri + [0, 0, -1, 0, 0, -1, 0, 0, -1]))
        w.add_group(g)
        w.add_group(g2)
        w.add_point_light((0, 5, -10), (1, 1, 1))
        return w

    o = np.array([[0.0, 0.4, -5.0], [3.0, 0.4, -5.0], [0.3, 0.2, -5.0], [3.3, 0.2, -5.0]], dtype=np.float32)
    d = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (4, 1))
    grgb, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 2)
    rrgb, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 2)
    assert np.array_equal(gid, rid) and np.array_equal(gt.view(np.uint32), rt_.view(np.uint32))
    assert list(rid) == [0, 3, 0, 3]  # NaN pair: the first; genuine duplicates: the later one (Q2)


_COPY_A = [0, 0, 0, 4, 0, 0, 0, 4, 0]
_COPY_X = [4, 0, 0, 0, 4, 0, 0, 0, 0]  # the corners of _COPY_A in another order: another Shape, the very same t


@pytest.mark.parametrize("order,expected", [("axa", 0), ("aa", 1), ("aaa", 2), ("aaxa", 1), ("axaa", 0), ("xaa", 0),
                                            ("aaxaa", 1), ("asa", 2), ("aSa", 0)])
@pytest.mark.parametrize("where", ["group", "moved group", "loose"])
def test_a_coincident_triangle_between_two_copies_of_a_triangle(order, expected, where):
    """Of two equal entries the reference's Intersections drops the earlier only when they are NEIGHBOURS in the list
    sorted by t.  A third triangle of the same three corners in another order (x), listed between two
    copies (a) and hit at the very same t (corners chosen so that every order gives t = 5 exactly), keeps them apart:
    the hit is the last of the FIRST run of equal neighbours at that t (rt_device.h end_of_copy_run; found by
    tools/gpu_fuzz.py frames, seed 20146).  s: a sphere hit elsewhere (no tie); S: a sphere hit at t = 5 as well.
    Through rays and through pixels, in a group, in a transformed group, and loose in the world."""
    def shapes(w):
        for c in order:
            if c == "s":
                yield w.shape("sphere", [("translate", 1, 1, -3)], None)
            elif c == "S":  # radius 1 around (1, 1, -1): first met at z = 0 by the ray below, t = 5
                yield w.shape("sphere", [("translate", 1, 1, -1)], None)
            else:
                yield w.shape("triangle", (), None, params=_COPY_A if c == "a" else _COPY_X)

    def build(w):
        if where == "loose":
            for sh in shapes(w):
                w.add_shape(sh)
        else:
            g = w.group([("translate", 0.5, 0.25, 0)] if where == "moved group" else ())
            for sh in shapes(w):
                w.group_add_shape(g, sh)
            w.add_group(g)
        w.add_point_light((0, 5, 10), (1, 1, 1))
        return w

    shift = (0.5, 0.25) if where == "moved group" else (0.0, 0.0)
    o = np.array([[1.0 + shift[0], 1.0 + shift[1], 5.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)
    _, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 0)
    _, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 0)
    assert gt[0] == rt_[0] == 5.0
    assert rid[0] == expected and gid[0] == rid[0]
    cam = B.Camera(48, 32, 0.9, (1.2, 1.1, 6.0), (1.2, 1.1, 0.0), (0, 1, 0))
    g = build(host.new_world(0)).render(cam, aux=True)
    r = build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, f"copies {order} {where}")
    assert len(np.unique(r.object_id)) >= 2


@pytest.mark.parametrize("tree,expected", [
    (("a", [("a",)]), 1),                     # a copy in a child group: neighbours all the same
    (("a", [("x",), ("a",)]), 0),             # ... with another group's x at the same t between them
    (("a", [("s",), ("a",)]), 2),             # ... with another group's sphere between them, met elsewhere
    (("aa", [("xa",)]), 1),
    (("", [("a", [("a",)]), ("a",), ("xa",)]), 2),
    (("a", [("a", [], "moved")]), 1),         # an equal shape under ANOTHER transform, moved within its own plane: same t
    (("a", [("x", [], "moved"), ("a",)]), 0),
    (("a", [("a", [], "moved"), ("x",), ("a", [], "moved"), ("a",)]), 1),
])
def test_copies_of_a_triangle_in_different_groups(tree, expected):
    """Copies in different groups under one transformed ancestor are neighbours in the reference's list like any
    others (groups depth-first, shapes before children): end_of_copy_run walks that order with the reference's own
    box tests.  Ids are ranks: a group's shapes, then its children's."""
    def make(w, spec):
        g = w.group([("translate", 0, 0, 0)] if len(spec) < 3 else [("translate", 0.25, 0.5, 0)])
        for c in spec[0]:
            if c == "s":
                w.group_add_shape(g, w.shape("sphere", [("translate", 1, 1, -3)], None))
            else:
                w.group_add_shape(g, w.shape("triangle", (), None, params=_COPY_A if c == "a" else _COPY_X))
        for child in (spec[1] if len(spec) > 1 else []):
            w.group_add_child(g, make(w, child))
        return g

    def build(w):
        w.add_group(make(w, tree))
        w.add_point_light((0, 5, 10), (1, 1, 1))
        return w

    o = np.array([[1.5, 1.5, 5.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)
    _, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 0)
    _, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 0)
    assert gt[0] == rt_[0] == 5.0
    assert rid[0] == expected and gid[0] == rid[0]
    cam = B.Camera(48, 32, 0.9, (1.2, 1.1, 6.0), (1.2, 1.1, 0.0), (0, 1, 0))
    check(build(host.new_world(0)).render(cam, aux=True), build(_oracle().new_world()).render(cam, aux=True), f"copies {tree}")


@pytest.mark.parametrize("leaves,loose,expected", [("as", "a", 2), ("ax", "a", 2), ("aa", "a", 2), ("sa", "xa", 1),
                                                   ("xa", "a", 0), ("as", "xa", 0)])
def test_copies_of_a_triangle_in_a_csg(leaves, loose, expected):
    """The entries of a csg are what its evaluation leaves of its leaves' intersections (csg.rs:26-72), first in the
    world's list (world.rs:45-49): a leaf that is a copy of a later triangle is its neighbour like any other entry.
    A union of two leaves, then loose triangles; s: a sphere met elsewhere.  (ax + a: the union drops the right leaf's x,
    met inside the left leaf at the same t, so the two a are neighbours.)"""
    def shape(w, c):
        if c == "s":
            return w.shape("sphere", [("translate", 1, 1, -3)], None)
        return w.shape("triangle", (), None, params=_COPY_A if c == "a" else _COPY_X)

    def build(w):
        w.add_csg(w.csg(B.CSG_UNION, B.CHILD_SHAPE, shape(w, leaves[0]), B.CHILD_SHAPE, shape(w, leaves[1])))
        for c in loose:
            w.add_shape(shape(w, c))
        w.add_point_light((0, 5, 10), (1, 1, 1))
        return w

    o = np.array([[1.5, 1.5, 5.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)
    _, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 0)
    _, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 0)
    assert gt[0] == rt_[0] == 5.0
    assert rid[0] == expected and gid[0] == rid[0]
    cam = B.Camera(48, 32, 0.9, (1.2, 1.1, 6.0), (1.2, 1.1, 0.0), (0, 1, 0))
    check(build(host.new_world(0)).render(cam, aux=True), build(_oracle().new_world()).render(cam, aux=True), f"csg copies {leaves} {loose}")


@pytest.mark.parametrize("kind", ["sphere", "cube", "cylinder", "cone", "plane"])
@pytest.mark.parametrize("order,expected", [("aaa", 2), ("aba", 0), ("aab", 1), ("baa", 0), ("abaa", 0), ("aabaa", 1)])
@pytest.mark.parametrize("grouped", [False, True])
def test_equal_solids_listed_more_than_once(kind, order, expected, grouped):
    """The run of equal neighbours is a rule for every kind of shape: a sphere listed three times reports the third, and
    a sphere of another material in the same place (b: another Shape, the same t) between two of them keeps them apart."""
    params = [-1.0, 1.0, 1.0] if kind in ("cylinder", "cone") else None

    def build(w):
        shapes = [w.shape(kind, [("translate", 0, 0.25, 0)], B.Material(color=(1, 0.2, 0.2) if c == "a" else (0.2, 1, 0.2)),
                          params=params) for c in order]
        if grouped:
            g = w.group([("translate", 0.5, 0, 0)])
            for sh in shapes:
                w.group_add_shape(g, sh)
            w.add_group(g)
        else:
            for sh in shapes:
                w.add_shape(sh)
        w.add_point_light((0, 5, -10), (1, 1, 1))
        return w

    x = 0.6 if grouped else 0.1
    o = np.array([[x, 0.5, -5.0], [x + 0.2, 0.3, -5.0]], dtype=np.float32)
    d = np.array([[0, -0.25 if kind == "plane" else 0.0, 1.0]] * 2, dtype=np.float32)
    _, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 0)
    _, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 0)
    assert np.array_equal(gt.view(np.uint32), rt_.view(np.uint32)) and (rid >= 0).any()
    assert all(r in (-1, expected) for r in rid) and np.array_equal(gid, rid)
    cam = B.Camera(40, 30, 0.9, (0.3, 1.0, -6.0), (0.3, 0.2, 0.0), (0, 1, 0))
    check(build(host.new_world(0)).render(cam, aux=True), build(_oracle().new_world()).render(cam, aux=True), f"{kind} {order}")


@pytest.mark.parametrize("grouped", [False, True])
def test_equal_spheres_hit_at_infinity(grouped):
    """A direction so short that dot(d, d) underflows gives t = +inf, a hit the reference shades (0 < inf), and equal
    entries at inf are neighbours like any others: three spheres in one place report the third."""
    def build(w):
        shapes = [w.shape("sphere", (), None) for _ in range(3)]
        if grouped:
            g = w.group([("translate", 0, 0, 0)])
            for sh in shapes:
                w.group_add_shape(g, sh)
            w.add_group(g)
        else:
            for sh in shapes:
                w.add_shape(sh)
        w.add_point_light((0, 5, -10), (1, 1, 1))
        return w

    o = np.array([[0, 0, -5.0]] * 4, dtype=np.float32)
    d = np.array([[0, 0, 1e-23], [0, 0, 1e-22], [0, 0, 3e-23], [0, 0, 1.0]], dtype=np.float32)
    _, gt, gid, _ = build(host.new_world(0)).color_at(o, d, 0)
    _, rt_, rid, _ = build(_oracle().new_world()).color_at(o, d, 0)
    assert np.array_equal(gid, rid) and np.array_equal(gt.view(np.uint32), rt_.view(np.uint32))
    assert np.isinf(rt_[0]) and rid[0] == 2 and rid[3] == 2


def test_the_bound_on_the_search_between_copies():
    """The one bound of the pass over the ids (RT_COPY_GAP_MAX = 65 536 intersections per hit, rt_device.h): a triangle
    listed again 70 000 places on in ONE undivided group.  The reference reports the later copy; the device stops its
    search where the budget ends and reports the winner -- the id only: t is the same.  With
    60 000 between them the search gets there."""
    def world(w, between):
        far = np.zeros((between, 3, 3), dtype=np.float32)
        far[:, :, 0] = 100.0 + np.arange(between, dtype=np.float32)[:, None] * 0.01
        far[:, 1, 1] = 1.0
        far[:, 2, 2] = 1.0
        tri = np.array([[0, 0, 0], [4, 0, 0], [0, 4, 0]], dtype=np.float32)
        corners = np.concatenate([tri[None], far, tri[None]])
        positions = corners.reshape(-1, 3)
        vidx = np.arange(len(positions), dtype=np.int32).reshape(-1, 3)
        g = w.group(())
        w.group_add_mesh(g, None, positions, None, vidx, None)
        w.add_group(g)
        w.add_point_light((0, 5, 10), (1, 1, 1))
        return w

    o = np.array([[1.0, 1.0, 5.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)
    for between, expected in ((60000, 60001), (70000, 0)):
        _, gt, gid, _ = world(host.new_world(0), between).color_at(o, d, 0)
        _, rt_, rid, _ = world(_oracle().new_world(), between).color_at(o, d, 0)
        assert gt[0] == rt_[0] == 5.0 and rid[0] == between + 1
        assert gid[0] == expected


@pytest.mark.parametrize("order,expected", [("aaa", 2), ("aba", 0), ("aab", 1)])
@pytest.mark.parametrize("cells", [2, 4])
def test_equal_solids_under_an_area_light(order, expected, cells):
    """The same rule through the direct kernels (group-less worlds under a multi-sample area light: k_render_direct,
    hoisted for worlds of a few shapes): they write the winner's id and the pass over the ids follows."""
    def build(w):
        for c in order:
            w.add_shape(w.shape("sphere", [("translate", 0, 1, 0)], B.Material(color=(1, 0.2, 0.2) if c == "a" else (0.2, 1, 0.2))))
        w.add_shape(w.shape("plane", (), B.Material(color=(0.9, 0.9, 0.9))))
        w.add_area_light((-3, 6, -4), (2, 0, 0), cells, (0, 0, 2), cells, (1, 1, 1))
        return w

    cam = B.Camera(80, 60, 0.9, (0.5, 3.0, -6.0), (0.5, 0.5, 0.0), (0, 1, 0))
    ro = B.RenderOptions(jitter_mode=B.JITTER_COUNTER, seed=3)
    g = build(host.new_world(0)).render(cam, ro, aux=True)
    r = build(_oracle().new_world()).render(cam, ro, aux=True)
    check(g, r, f"area light {order}")
    ids = set(np.unique(r.object_id)) - {-1}
    assert ids == {expected, len(order)}  # the spheres' entry and the floor


@pytest.mark.parametrize("n,threshold", [(48, 6), (96, 20)])
def test_rays_through_the_edges_and_corners_of_a_mesh(n, threshold):
    """A flat n x n grid of quads at dyadic coordinates, divided, under an auxiliary tree: rays straight down the grid
    lines meet two triangles (an edge) or up to six (a corner) at the very same t, in different leaves of the tree,
    and the hit is the one listed first -- the (t, rank) order of the traversal's hit keys."""
    xs = (np.arange(n + 1, dtype=np.float32) - n / 2) / 8.0
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    positions = np.stack([gx.ravel(), gy.ravel(), np.zeros((n + 1) ** 2, dtype=np.float32)], axis=1)
    normals = np.tile(np.array([[0, 0, -1]], dtype=np.float32), (len(positions), 1))
    iu, iv = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a, b_, c, d_ = iu * (n + 1) + iv, (iu + 1) * (n + 1) + iv, (iu + 1) * (n + 1) + iv + 1, iu * (n + 1) + iv + 1
    vidx = np.concatenate([np.stack([a, b_, c], axis=-1).reshape(-1, 3), np.stack([a, c, d_], axis=-1).reshape(-1, 3)]).astype(np.int32)
    rng = np.random.RandomState(n)
    vidx = vidx[rng.permutation(len(vidx))]  # neighbours far apart in the list: the ranks decide, not the leaves

    def build(w):
        g = w.group([("translate", 0.25, -0.5, 0)])
        w.group_add_mesh(g, B.Material(color=(0.8, 0.4, 0.2)), positions, normals, vidx, vidx)
        w.group_divide(g, threshold)
        w.add_group(g)
        w.add_point_light((0, 5, -10), (1, 1, 1))
        return w

    half = (np.arange(2 * n + 1, dtype=np.float32) - n) / 16.0  # grid lines and the middles between them
    px, py = np.meshgrid(half, half, indexing="ij")
    o = np.stack([px.ravel() + 0.25, py.ravel() - 0.5, np.full(px.size, -5.0, dtype=np.float32)], axis=1).astype(np.float32)
    d = np.tile(np.array([[0, 0, 1.0]], dtype=np.float32), (len(o), 1))
    gw, ow = build(host.new_world(0)), build(_oracle().new_world())
    assert gw.device_scene() and host.hip_lib()
    _, gt, gid, guv = gw.color_at(o, d, 0)
    _, rt_, rid, ruv = ow.color_at(o, d, 0)
    assert (rid >= 0).mean() > 0.9 and np.array_equal(gid, rid)
    hit = rid >= 0
    assert np.array_equal(gt[hit].view(np.uint32), rt_[hit].view(np.uint32))
    assert np.array_equal(guv[hit].view(np.uint32), ruv[hit].view(np.uint32))


@pytest.mark.parametrize("op", ["union", "intersect", "difference"])
@pytest.mark.parametrize("kind", ["sphere", "cube", "cylinder"])
@pytest.mark.parametrize("same", [True, False])
def test_csg_of_coincident_solids(op, kind, same):
    """A csg of two solids in the very same place (equal shapes, or equal but for the material): every crossing is a
    tie of a left and a right entry, and the filter sees them in list order after equal neighbours were dropped
    (csg.rs:26-72); nested once more with a sphere, and a loose copy of the solid behind."""
    ops = {"union": B.CSG_UNION, "intersect": B.CSG_INTERSECT, "difference": B.CSG_DIFFERENCE}
    params = [-1.0, 1.0, 1.0] if kind == "cylinder" else None

    def build(w):
        a = B.Material(color=(0.9, 0.3, 0.3))
        b = a if same else B.Material(color=(0.3, 0.9, 0.3))
        left = w.shape(kind, [("rotate-y", 0.4)], a, params=params)
        right = w.shape(kind, [("rotate-y", 0.4)], b, params=params)
        c = w.csg(ops[op], B.CHILD_SHAPE, left, B.CHILD_SHAPE, right)
        ball = w.shape("sphere", [("scale", 0.7, 0.7, 0.7), ("translate", 0.6, 0.5, -0.5)], B.Material(color=(0.3, 0.3, 0.9)))
        w.add_csg(w.csg(B.CSG_DIFFERENCE, B.CHILD_CSG, c, B.CHILD_SHAPE, ball))
        w.add_shape(w.shape(kind, [("rotate-y", 0.4)], a, params=params))
        w.add_shape(w.shape("plane", [("translate", 0, -1.5, 0)], B.Material(color=(0.9, 0.9, 0.5))))
        w.add_point_light((-4, 6, -8), (1, 1, 1))
        return w

    cam = B.Camera(96, 72, 0.8, (0.5, 1.2, -5.0), (0, 0, 0), (0, 1, 0))
    g = build(host.new_world(0)).render(cam, aux=True)
    r = build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, f"csg {op} {kind} {same}")


@pytest.mark.parametrize("order", ["ab", "ba", "aba", "abab", "aabb"])
@pytest.mark.parametrize("kind", ["sphere", "cube"])
def test_coincident_glass_of_different_refractive_index(order, kind):
    """Two kinds of glass in the very same place: every surface is a tie of entries of different shapes, taken in
    list order (stable sort) by `hit` and by the containers of n1 / n2 (intersections.rs:141-170)."""
    mats = {"a": B.Material(color=(0.1, 0.1, 0.1), diffuse=0.1, transparency=0.9, reflective=0.5, refractive_index=1.5),
            "b": B.Material(color=(0.1, 0.2, 0.1), diffuse=0.1, transparency=0.8, reflective=0.3, refractive_index=2.0)}

    def build(w):
        for c in order:
            w.add_shape(w.shape(kind, [("rotate-y", 0.4)], mats[c]))
        w.add_shape(w.shape("plane", [("translate", 0, -1.5, 0)], B.Material(color=(0.9, 0.9, 0.5))))
        w.add_point_light((-4, 6, -8), (1, 1, 1))
        return w

    cam = B.Camera(96, 72, 0.8, (0.5, 1.2, -5.0), (0, 0, 0), (0, 1, 0))
    g = build(host.new_world(0)).render(cam, aux=True)
    r = build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, f"coincident glass {kind} {order}")
    assert g.stats["refract"] > 500


@pytest.mark.parametrize("flags", [(False, True), (True, False), (False, False, True), (False, True, False)])
@pytest.mark.parametrize("grouped", [False, True, "mesh", "in mesh"])
def test_coincident_solids_that_differ_in_shadow(flags, grouped):
    """World::is_shadowed looks at the closest hit only, and of hits at one t at the first listed (world.rs:101-111,
    SURVEY.md Q5): a shadow-less sphere listed before a shadow-casting one in the same place casts no shadow, the other
    order does.  (Two passes on the device: the nearest shadow-less hit, then any occluder before it in (t, rank).)"""
    def build(w):
        shapes = [w.shape("sphere", [("translate", 0, 1, 0)], B.Material(color=(0.3 + 0.2 * k, 0.5, 0.5)), shadow=f)
                  for k, f in enumerate(flags)]
        if grouped == "in mesh":  # coincident triangles inside the group whose auxiliary tree the shadow rays traverse
            shapes = [w.shape("triangle", (), B.Material(color=(0.3 + 0.2 * k, 0.5, 0.5)), shadow=f,
                              params=[-4, 3, -3, 3, 3, -3, 0, 3, 4]) for k, f in enumerate(flags)]
        if grouped is True:
            g = w.group([("translate", 0.5, 0, 0)])
            for sh in shapes:
                w.group_add_shape(g, sh)
            w.add_group(g)
        elif grouped != "in mesh":
            for sh in shapes:
                w.add_shape(sh)
        if grouped in ("mesh", "in mesh"):  # a mesh with an auxiliary tree: the shadow rays take the two-pass search
            from raytracer_amd import synth
            m = synth.synth_dragon(50, 50)[0]
            gm = w.group([("scale", 0.4, 0.4, 0.4), ("translate", 2.2, 0.8, 0.5)])
            w.group_add_mesh(gm, B.Material(color=(0.8, 0.3, 0.2)), m["positions"], m["normals"], m["vidx"], m["nidx"])
            if grouped == "in mesh":
                for sh in shapes:
                    w.group_add_shape(gm, sh)
            w.group_divide(gm, 50)
            w.add_group(gm)
        w.add_shape(w.shape("plane", (), B.Material(color=(0.9, 0.9, 0.9))))
        w.add_point_light((-3, 6, -4), (1, 1, 1))
        return w

    cam = B.Camera(80, 60, 0.9, (0.5, 3.0, -6.0), (0.5, 0.5, 0.0), (0, 1, 0))
    g = build(host.new_world(0)).render(cam, aux=True)
    r = build(_oracle().new_world()).render(cam, aux=True)
    check(g, r, f"shadow flags {flags}")
    assert np.array_equal(to_u8(g.rgb), to_u8(r.rgb))


def _dup_glass_world(w, outer, inner, grouped, kind):
    """`outer` glass solids in one place, `inner` denser ones half the size inside them, one more glass one after
    those; a floor and a light."""
    glass = B.Material(color=(0.1, 0.1, 0.1), diffuse=0.1, transparency=0.9, reflective=0.9, refractive_index=1.5)
    dense = B.Material(color=(0.2, 0.1, 0.1), diffuse=0.2, transparency=0.8, reflective=0.3, refractive_index=2.0)
    params = [-1.0, 1.0, 1.0] if kind in ("cylinder", "cone") else None
    shapes = [w.shape(kind, (), glass, params=params) for _ in range(outer)]
    shapes += [w.shape(kind, [("scale", 0.5, 0.5, 0.5)], dense, params=params) for _ in range(inner)]
    shapes += [w.shape(kind, (), glass, params=params)]
    if grouped:
        g = w.group([("rotate-y", 0.3)])
        for sh in shapes:
            w.group_add_shape(g, sh)
        w.add_group(g)
    else:
        for sh in shapes:
            w.add_shape(sh)
    w.add_shape(w.shape("plane", [("translate", 0, -1.5, 0)], B.Material(color=(0.9, 0.9, 0.5))))
    w.add_point_light((-4, 6, -8), (1, 1, 1))
    return w


@pytest.mark.parametrize("kind,outer,inner,grouped", [("sphere", 1, 2, False), ("sphere", 2, 1, True), ("sphere", 3, 2, False),
                                                      ("cube", 2, 2, True), ("cube", 3, 0, False), ("cylinder", 2, 1, False),
                                                      ("cylinder", 1, 2, True)])
def test_transparent_solids_listed_more_than_once(kind, outer, inner, grouped):
    """n1 / n2 of Computations::prepare (intersections.rs:141-170) walk the list AFTER equal neighbours were dropped,
    with `containers` compared by Shape ==: glass listed two and three times around denser glass listed twice gives the
    oracle's colours, ids and ray counts (tools/gpu_dup_glass.py runs all 54 combinations)."""
    cam = B.Camera(96, 72, 0.8, (0.5, 1.2, -5.0), (0, 0, 0), (0, 1, 0))
    g = _dup_glass_world(host.new_world(0), outer, inner, grouped, kind).render(cam, aux=True)
    r = _dup_glass_world(_oracle().new_world(), outer, inner, grouped, kind).render(cam, aux=True)
    check(g, r, f"glass {kind} {outer} {inner} {grouped}")
    assert g.stats["refract"] > 500


def test_copies_of_a_triangle_in_parts():
    """The pass over the ids (k_copies_pixels) finds a pixel's ray from its row in the PART: interleaved parts of a
    frame whose every pixel needs the pass give the ids of the whole frame, which are the oracle's."""
    from raytracer_amd import tiles

    class Copies:
        def build(self, w):
            g = w.group([("rotate-z", 0.3)])
            for c in "aaxaxa":
                w.group_add_shape(g, w.shape("triangle", (), None, params=_COPY_A if c == "a" else _COPY_X))
            w.add_group(g)
            w.add_point_light((0, 5, 10), (1, 1, 1))
            return w

        def camera(self, hsize, vsize):
            return B.Camera(hsize, vsize, 0.9, (1.2, 1.1, 6.0), (1.2, 1.1, 0.0), (0, 1, 0))

    raw = Raw(Copies(), (70, 45))
    _, ids, _ = raw.render()
    r = Copies().build(_oracle().new_world()).render(Copies().camera(70, 45), aux=True)
    assert np.array_equal(ids, r.object_id) and (ids >= 0).mean() > 0.2
    for n, tile_h in ((3, 2), (4, 1), (2, 8)):
        idc = np.empty_like(ids)
        for p in range(n):
            _, oid, _ = raw.render(tile_h=tile_h, n_parts=n, part=p)
            idc[tiles.part_rows(45, tile_h, n, p)] = oid
        assert np.array_equal(idc, ids), (n, tile_h)
