"""include/rt_loaders.h (librt_host.so): the native OBJ / PPM loaders and the PPM writer.

Pinned by the reference's own tests for these formats -- obj.rs:48-205 (triangles, polygon fan, groups, smooth
triangles) and canvas.rs:158-170 (from_ppm) -- and held against the harness's Python readers (raytracer_amd/obj.py,
scene.read_ppm, canvas.write_ppm), which the oracle-side tests use, on the reference's sample OBJ files and on synthetic
files with every construct the parsers accept.  The third-party crates behind the reference's functions (`wavefront`
0.2, `image` 0.24) are not in the reference tree: whitespace of to_ppm's output and 16-bit PPM samples are unpinned and
say so where they are handled.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from raytracer_amd import abi, builder as B, canvas, host, loaders, obj as pyobj, scene as pyscene


def _tris(groups, g=0):
    G = groups[g]
    return G["positions"][G["vidx"]]


def test_library_exports_every_loader_symbol():
    L = host.host_lib()
    for name in loaders.EXPORTS:
        assert hasattr(L, name), name


def test_kat_obj_triangles_and_polygon():  # obj.rs:48-123
    g = loaders.parse_obj_bytes(b"\nv -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\nf 1 2 3\nf 1 3 4\n")
    assert len(g) == 1
    assert np.array_equal(_tris(g), [[[-1, 1, 0], [-1, 0, 0], [1, 0, 0]], [[-1, 1, 0], [1, 0, 0], [1, 1, 0]]])
    assert (g[0]["nidx"] == -1).all() and g[0]["normals"] is None
    g = loaders.parse_obj_bytes(b"\nv -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\nf 1 2 3 4 5\n")
    assert np.array_equal(_tris(g), [[[-1, 1, 0], [-1, 0, 0], [1, 0, 0]], [[-1, 1, 0], [1, 0, 0], [1, 1, 0]],
                                     [[-1, 1, 0], [1, 1, 0], [0, 2, 0]]])


def test_kat_obj_groups_and_smooth_triangles():  # obj.rs:125-205
    g = loaders.parse_obj_bytes(b"\nv -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\ng FirstGroup\nf 1 2 3\n\ng SecondGroup\nf 1 3 4\n")
    assert [x["name"] for x in g] == ["FirstGroup", "SecondGroup"]
    assert np.array_equal(_tris(g, 0), [[[-1, 1, 0], [-1, 0, 0], [1, 0, 0]]])
    assert np.array_equal(_tris(g, 1), [[[-1, 1, 0], [1, 0, 0], [1, 1, 0]]])
    g = loaders.parse_obj_bytes(b"\nv 0 1 0\nv -1 0 0\nv 1 0 0\n\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\n\n"
                                b"f 1//3 2//1 3//2\nf 1//3 2//1 3//2\n")
    G = g[0]
    assert np.array_equal(_tris(g), [[[0, 1, 0], [-1, 0, 0], [1, 0, 0]]] * 2)
    assert np.array_equal(G["normals"][G["nidx"]], [[[0, 1, 0], [-1, 0, 0], [1, 0, 0]]] * 2)


def _same_groups(a, b):
    assert [x["name"] for x in a] == [x["name"] for x in b]
    for x, y in zip(a, b):
        assert np.array_equal(x["positions"].view(np.uint32), y["positions"].view(np.uint32))
        assert (x["normals"] is None) == (y["normals"] is None)
        if x["normals"] is not None:
            assert np.array_equal(x["normals"].view(np.uint32), y["normals"].view(np.uint32))
        assert np.array_equal(x["vidx"], y["vidx"]) and np.array_equal(x["nidx"], y["nidx"])


@pytest.mark.parametrize("name", ["space_ship.obj", "teapot_low.obj"])
@pytest.mark.parametrize("threads", [1, 3, 16])
def test_reference_sample_files_parse_like_the_python_reader(name, threads):
    path = os.path.join(ROOT, "scenes", "obj", name)
    data = open(path, "rb").read()
    _same_groups(loaders.parse_obj_bytes(data, threads), pyobj.parse_obj(path))


def _messy_obj(rng, n_v=700, n_f=900):
    """Comments, CRLF, tabs, negative indices, v/vt/vn and v//vn and v/vt corners, polygons, groups that are re-opened,
    a `g` line without faces, statements the reader skips, exponents and signs in numbers."""
    lines = ["# header", "mtllib x.mtl", "o thing"]
    nv = nn = 0
    groups = ["", "hull", "wing", "hull", "tail fin", "wing"]
    gi = 0
    for k in range(n_f):
        for _ in range(int(rng.randint(0, 3))):
            x = rng.uniform(-50, 50, 3)
            lines.append(f"v {x[0]:.7g} {x[1]:+.5e}\t{x[2]:.9f}" + ("\r" if k % 7 == 0 else ""))
            nv += 1
            if rng.rand() < 0.5:
                y = rng.uniform(-1, 1, 3)
                lines.append(f"vn {y[0]:.6f} {y[1]:.6f} {y[2]:.6f}")
                nn += 1
        if k % 50 == 10:
            gi += 1
            lines.append(f"g {groups[gi % len(groups)]}" if groups[gi % len(groups)] else "g")
        if k % 97 == 5:
            lines.append("g empty_group")
            lines.append(f"g {groups[gi % len(groups)]}" if groups[gi % len(groups)] else "g")
        if k % 13 == 0:
            lines.append("s off" if k % 2 else "usemtl m")
            lines.append("vt 0.5 0.5")
        if nv < 3:
            continue
        nc = 3 + int(rng.randint(0, 4) == 0) * int(rng.randint(1, 4))
        corners = []
        style = int(rng.randint(0, 4)) if nn else int(rng.randint(0, 2))
        for _ in range(nc):
            vi = int(rng.randint(1, nv + 1))
            if rng.rand() < 0.3:
                vi = vi - nv - 1  # relative
            if style == 0:
                corners.append(f"{vi}")
            elif style == 1:
                corners.append(f"{vi}/1")
            else:
                ni = int(rng.randint(1, nn + 1))
                if rng.rand() < 0.3:
                    ni = ni - nn - 1
                corners.append(f"{vi}//{ni}" if style == 2 else f"{vi}/1/{ni}")
        lines.append("f " + " ".join(corners) + ("   " if k % 5 == 0 else ""))
    return ("\n".join(lines) + ("\n" if rng.rand() < 0.5 else "")).encode()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_messy_files_parse_like_the_python_reader_whatever_the_chunking(seed):
    data = _messy_obj(np.random.RandomState(seed))
    want = pyobj.parse_obj_text(data.decode())
    assert len(want) >= 4
    for threads in (1, 2, 5, 16, 64):  # chunk borders fall between any two lines
        _same_groups(loaders.parse_obj_bytes(data, threads), want)


def test_malformed_obj_files_are_refused():
    for bad, what in [(b"v 1 2\nf 1 1 1\n", "vertex"), (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "not defined"),
                      (b"v 0 0 0\nv 1 0 0\nf 1 2\n", "fewer than three"), (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x\n", "index"),
                      (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1//1 2//1 3//1\n", "normal")]:
        with pytest.raises(RuntimeError, match=what):
            loaders.parse_obj_bytes(bad)
    assert loaders.parse_obj_bytes(b"") == [] and loaders.parse_obj_bytes(b"# nothing\nv 1 2 3\n") == []


def test_group_add_obj_builds_the_world_the_python_path_builds():
    """rtl_group_add_obj = parse + one rth_group_add_mesh per OBJ group: the flattened scene is the same, array for
    array, as with the Python reader feeding group_add_mesh (what the oracle-side tests do)."""
    path = os.path.join(ROOT, "scenes", "obj", "teapot_low.obj")
    data = open(path, "rb").read()
    mat = B.Material(color=(0.9, 0.2, 0.1), reflective=0.3)

    def world(native):
        w = host.new_world(0)
        g = w.group([("scale", 0.1, 0.1, 0.1)])
        if native:
            assert loaders.group_add_obj(w, g, mat, data, threads=4) == len(pyobj.parse_obj(path))
        else:
            for og in pyobj.parse_obj(path):
                w.group_add_mesh(g, mat, og["positions"], og["normals"], og["vidx"], og["nidx"])
        w.group_divide(g, 20)
        w.add_group(g)
        w.add_point_light((0, 10, -10), (1, 1, 1))
        return abi.scene_arrays(w.flatten()), w

    a, wa = world(True)
    b, wb = world(False)
    assert a.keys() == b.keys()
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), k
    assert len(a["tri_geom"]) > 200


def test_kat_canvas_from_ppm():  # canvas.rs:158-170
    img = loaders.ppm_decode(b"P3\n2 2\n100\n100 100 100  50 50 50\n75 50 25  0 0 0\n")
    assert img.shape == (2, 2, 3)
    assert np.array_equal(img[1, 0], np.array([0.75, 0.5, 0.25], dtype=np.float32))  # pixel_at(0, 1)
    assert np.array_equal(img[0, 0], [1, 1, 1]) and np.array_equal(img[1, 1], [0, 0, 0])


@pytest.mark.parametrize("maxval", [1, 100, 255])
def test_ppm_decode_equals_the_python_reader(tmp_path, maxval):
    rng = np.random.RandomState(maxval)
    h, w = 37, 53
    px = rng.randint(0, maxval + 1, size=(h, w, 3))
    p3 = (f"P3\n# made by a test\n{w} {h} # size\n{maxval}\n" +
          "\n".join(" ".join(str(int(v)) for v in row.reshape(-1)) + (" # row" if i % 5 == 0 else "") for i, row in enumerate(px)) + "\n")
    p6 = f"P6\n{w}\t{h}\n#c\n{maxval}\n".encode() + px.astype(np.uint8).tobytes()
    for name, data in (("a.ppm", p3.encode()), ("b.ppm", p6)):
        path = tmp_path / name
        path.write_bytes(data)
        got = loaders.ppm_decode(data)
        want = pyscene.read_ppm(str(path))
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    with pytest.raises(RuntimeError, match="fewer samples"):
        loaders.ppm_decode(p6[:-1])
    with pytest.raises(RuntimeError, match="fewer samples"):
        loaders.ppm_decode(p3.encode()[: len(p3) // 2])


def test_ppm_decode_refuses_what_is_not_pinned():
    with pytest.raises(RuntimeError, match="maxval > 255"):
        loaders.ppm_decode(b"P3\n1 1\n65535\n1 2 3\n")
    with pytest.raises(RuntimeError, match="P3 and P6"):
        loaders.ppm_decode(b"P5\n1 1\n255\n\x00")
    with pytest.raises(RuntimeError, match="above maxval"):
        loaders.ppm_decode(b"P3\n1 1\n10\n1 2 30\n")


def test_truncated_p6_files_are_refused_before_anything_is_read():
    """A P6 file that ends at the maxval token, one byte later, or one byte short of its samples (the
    first used to step past the end of the caller's buffer); and a header whose sizes overflow."""
    with pytest.raises(RuntimeError, match="truncated"):
        loaders.ppm_decode(b"P6 1 1 255")
    with pytest.raises(RuntimeError, match="truncated"):
        loaders.ppm_decode(b"P6 1 1 255#")
    with pytest.raises(RuntimeError, match="fewer samples"):
        loaders.ppm_decode(b"P6 1 1 255\n")
    with pytest.raises(RuntimeError, match="fewer samples"):
        loaders.ppm_decode(b"P6 1 1 255\n\x01\x02")
    assert np.array_equal(loaders.ppm_decode(b"P6 1 1 255\n\x00\x33\xff")[0, 0],
                          np.array([0, 0x33, 255], dtype=np.float32) / np.float32(255))
    for hdr in (b"P6 4294967295 4294967295 255\n", b"P3 4294967295 4294967295 255\n", b"P6 4294967295 3 255\n"):
        with pytest.raises(RuntimeError, match="too large|fewer samples"):
            loaders.ppm_decode(hdr + b"\x00" * 16)


def test_obj_groups_get_tables_of_their_own_vertices():
    """A file with many `g` sections: every group is handed only the vertices it uses (rtl_group_add_obj), and the
    flattened world is the one the whole-file tables give (the Python reader passes those)."""
    rng = np.random.RandomState(11)
    n_groups, per = 40, 12
    lines, nv = [], 0
    for g in range(n_groups):
        lines.append(f"g part{g}")
        for _ in range(per):
            x, y, z = rng.uniform(-1, 1, 3) + g
            lines.append(f"v {x:.6f} {y:.6f} {z:.6f}")
            lines.append(f"vn {rng.uniform(-1, 1):.6f} {rng.uniform(-1, 1):.6f} 1.0")
        for k in range(per - 2):
            a, b, c = nv + 1, nv + k + 2, nv + k + 3
            lines.append(f"f {a}//{a} {b}//{b} {c}//{c}")
        nv += per
    # a last group that reaches back into the first groups' vertices, by relative indices too
    lines += ["g reach", f"f 1//1 {per + 2}//{per + 2} -1//-1"]
    data = ("\n".join(lines) + "\n").encode()
    mat = B.Material(color=(0.2, 0.4, 0.6))

    def world(native):
        w = host.new_world(0)
        g = w.group([("translate", 0.5, 0, 0)])
        if native:
            assert loaders.group_add_obj(w, g, mat, data, threads=3) == n_groups + 1
        else:
            for og in loaders.parse_obj_bytes(data):
                w.group_add_mesh(g, mat, og["positions"], og["normals"], og["vidx"], og["nidx"])
        w.add_group(g)
        w.add_point_light((0, 10, -10), (1, 1, 1))
        return abi.scene_arrays(w.flatten())

    a, b = world(True), world(False)
    assert a.keys() == b.keys()
    for k in a:
        assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), k
    assert len(a["tri_geom"]) == n_groups * (per - 2) + 1


def test_uv_image_from_ppm_is_the_texture_the_python_path_uploads(tmp_path):
    rng = np.random.RandomState(4)
    px = rng.randint(0, 256, size=(16, 24, 3)).astype(np.uint8)
    data = b"P6\n24 16\n255\n" + px.tobytes()
    (tmp_path / "t.ppm").write_bytes(data)

    def world(native):
        w = host.new_world(0)
        uv = loaders.uv_image_ppm(w, data) if native else w.uv_image(pyscene.read_ppm(str(tmp_path / "t.ppm")))
        pat = w.pattern(B.PAT_TEXTURE_MAP, None, (), B.UVMAP_SPHERICAL, [uv])
        w.add_shape(w.shape("sphere", (), B.Material(pattern=pat)))
        return abi.scene_arrays(w.flatten())

    a, b = world(True), world(False)
    for k in ("texels", "images", "uvpatterns", "patterns"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert a["texels"].size == 16 * 24 * 3


@pytest.mark.parametrize("threads", [1, 5, 0])
def test_ppm_encode_is_the_python_writer_and_reads_back(tmp_path, threads):
    rng = np.random.RandomState(9)
    frame = rng.uniform(-0.2, 1.2, size=(61, 47, 3)).astype(np.float32)
    frame[0, 0] = [np.nan, np.inf, -np.inf]
    u8 = canvas.to_u8(frame)
    text = loaders.ppm_encode(u8, threads)
    canvas.write_ppm(str(tmp_path / "py.ppm"), frame)
    assert text == (tmp_path / "py.ppm").read_bytes()
    assert text.startswith(b"P3\n47 61\n255\n") and text.decode("utf-8")  # canvas.rs:144-156: the output is UTF-8
    back = loaders.ppm_decode(text)
    assert np.array_equal(back, u8.astype(np.float32) / np.float32(255))
    loaders.ppm_write(str(tmp_path / "native.ppm"), u8, threads)
    assert (tmp_path / "native.ppm").read_bytes() == text
    with pytest.raises(RuntimeError, match="cannot open"):
        loaders.ppm_write(str(tmp_path / "no_such_dir" / "x.ppm"), u8)


def test_large_ascii_ppm_is_decoded_in_pieces():
    """A P3 body of several MB is cut at line ends and parsed on threads: every sample lands where it belongs, comments
    and ragged lines included."""
    rng = np.random.RandomState(12)
    h, w = 512, 1024
    px = rng.randint(0, 256, size=h * w * 3)
    lines, i = [], 0
    while i < len(px):
        n = int(rng.randint(1, 40))
        lines.append(" ".join(str(int(v)) for v in px[i:i + n]) + (" # c 1 2 3" if len(lines) % 11 == 0 else ""))
        i += n
    data = (f"P3\n{w} {h}\n255\n" + "\n".join(lines) + "\n").encode()
    assert len(data) > 4 << 20
    img = loaders.ppm_decode(data)
    assert np.array_equal(img.reshape(-1), px.astype(np.float32) / np.float32(255))
    with pytest.raises(RuntimeError, match="fewer samples"):
        loaders.ppm_decode(data[: len(data) - 200000])
    with pytest.raises(RuntimeError, match="malformed sample|above maxval"):
        loaders.ppm_decode(data.replace(b" 17 ", b" 1x7 ", 1) if b" 17 " in data else data[:-10] + b" zz\n")
