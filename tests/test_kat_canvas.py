"""Known answers of the reference for what follows Camera::render (scene.rs:215-224): the Bayer matrices of
effects/dithering.rs (test valid_bayer_matrix, :58-91), Canvas::apply_dithering (canvas.rs:37-55) and the 8-bit rule of
Canvas::to_ppm (canvas.rs:73-75 = tuples.rs:367-377), restated in the oracle (orc_bayer_matrix, orc_canvas_to_u8), which
is what the device's u8 output path is held against on the GPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle


def bayer(n):
    out = np.zeros(n * n, dtype=np.float32)
    assert oracle.lib().orc_bayer_matrix(n, out.ctypes.data_as(C.c_void_p)) == 0
    return out.reshape(n, n)


def to_u8(rgb, dither=0, rows=None):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    r = None if rows is None else np.ascontiguousarray(rows, dtype=np.uint32)
    rc = oracle.lib().orc_canvas_to_u8(rgb.ctypes.data_as(C.c_void_p), w, h, dither,
                                       None if r is None else r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def test_valid_bayer_matrix():  # effects/dithering.rs:58-91, literally
    assert np.array_equal(bayer(2), np.float32(1 / 4) * np.array([[0, 2], [3, 1]], dtype=np.float32))
    assert np.array_equal(bayer(4), np.float32(1 / 16) * np.array(
        [[0, 8, 2, 10], [12, 4, 14, 6], [3, 11, 1, 9], [15, 7, 13, 5]], dtype=np.float32))
    assert np.array_equal(bayer(8), np.float32(1 / 64) * np.array(
        [[0, 32, 8, 40, 2, 34, 10, 42], [48, 16, 56, 24, 50, 18, 58, 26], [12, 44, 4, 36, 14, 46, 6, 38],
         [60, 28, 52, 20, 62, 30, 54, 22], [3, 35, 11, 43, 1, 33, 9, 41], [51, 19, 59, 27, 49, 17, 57, 25],
         [15, 47, 7, 39, 13, 45, 5, 37], [63, 31, 55, 23, 61, 29, 53, 21]], dtype=np.float32))
    b16 = bayer(16) * 256
    assert sorted(b16.ravel().tolist()) == list(range(256))  # every level once
    for n in (0, 1, 3, 77):  # check_pow2 (:44-55): N must be a power of two >= 2
        out = np.zeros(max(n * n, 1), dtype=np.float32)
        assert oracle.lib().orc_bayer_matrix(n, out.ctypes.data_as(C.c_void_p)) != 0


def test_quantisation_rule():
    """(0_f32.max(c).min(1.) * 255.).round() as u8: clamps, rounds half away from zero, NaN -> 0 (f32::max returns the
    other operand); the canvas of canvas.rs:144-148 (1.5, 0.5 and -0.5 channels)."""
    c = np.zeros((3, 5, 3), dtype=np.float32)
    c[0, 0] = (1.5, 0, 0)
    c[1, 2] = (0, 0.5, 0)
    c[2, 4] = (-0.5, 0, 1)
    out = to_u8(c)
    assert tuple(out[0, 0]) == (255, 0, 0) and tuple(out[1, 2]) == (0, 128, 0) and tuple(out[2, 4]) == (0, 0, 255)
    vals = np.array([np.nan, np.inf, -np.inf, 0.1 / 255, 0.5 / 255, 1.5 / 255, 254.5 / 255, 0.999999, 1e-30, -0.0],
                    dtype=np.float32)
    got = to_u8(np.repeat(vals[None, :, None], 3, axis=2))[0, :, 0]
    want = [0, 255, 0, 0, int(np.round(np.float32(0.5 / 255) * np.float32(255) + 1e-9)), 2, 255, 255, 0, 0]
    want[4] = 1 if np.float32(np.float32(0.5 / 255) * np.float32(255)) >= 0.5 else 0
    want[5] = 2 if np.float32(np.float32(1.5 / 255) * np.float32(255)) >= 1.5 else 1
    want[6] = 255 if np.float32(np.float32(254.5 / 255) * np.float32(255)) >= 254.5 else 254
    assert got.tolist() == want


@pytest.mark.parametrize("dither,n", [(1, 2), (2, 4), (3, 8), (4, 16)])
def test_black_and_white_dithering(dither, n):
    """apply_dithering::<N>(false) (canvas.rs:47-52): black where any channel is below bayer[x % N][y % N], else white."""
    rng = np.random.RandomState(n)
    rgb = rng.uniform(-0.1, 1.1, size=(37, 53, 3)).astype(np.float32)
    b = bayer(n)
    x, y = np.meshgrid(np.arange(53), np.arange(37))
    d = b[x % n, y % n]
    want = np.where((rgb < d[..., None]).any(axis=2), 0, 255).astype(np.uint8)
    got = to_u8(rgb, dither)
    assert np.array_equal(got, np.repeat(want[..., None], 3, axis=2))


def test_colour_dithering_and_row_mapping():
    """apply_dithering::<4>(true) (canvas.rs:42-44): the matrix value is added, then quantised; rows of a part dither by
    their row in the whole image."""
    rng = np.random.RandomState(9)
    rgb = rng.uniform(0, 1, size=(24, 31, 3)).astype(np.float32)
    b = bayer(4)
    x, y = np.meshgrid(np.arange(31), np.arange(24))
    summed = rgb + b[x % 4, y % 4][..., None]
    assert np.array_equal(to_u8(rgb, 5), to_u8(summed, 0))
    rows = np.array([r for r in range(24) if (r // 2) % 3 == 1], dtype=np.uint32)
    assert np.array_equal(to_u8(rgb[rows], 5, rows), to_u8(rgb, 5)[rows])
    assert np.array_equal(to_u8(rgb[rows], 2, rows), to_u8(rgb, 2)[rows])
