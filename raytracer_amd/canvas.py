"""Canvas quantisation and PPM output rules of the reference (edge of the hot path).

``to_u8`` is `(0_f32.max(c).min(1.) * 255.).round() as u8` (src/tuples.rs:367-377,
src/canvas.rs:73-75): clamp, scale in f32, round half away from zero.
"""
from __future__ import annotations

import numpy as np


def to_u8(rgb: np.ndarray) -> np.ndarray:
    x = np.asarray(rgb, dtype=np.float32)
    x = np.where(np.isnan(x), np.float32(0), x)  # 0_f32.max(NaN) = 0
    x = np.minimum(np.maximum(x, np.float32(0)), np.float32(1)) * np.float32(255)
    return np.floor(x + np.float32(0.5)).astype(np.uint8)  # non-negative: round-half-up == half-away


def write_ppm(path: str, rgb: np.ndarray) -> None:
    """Canvas::to_ppm (src/canvas.rs:57-84): ASCII P3, maxval 255."""
    img = to_u8(rgb)
    h, w, _ = img.shape
    with open(path, "w") as f:
        f.write(f"P3\n{w} {h}\n255\n")
        for row in img:
            f.write(" ".join(str(int(v)) for v in row.reshape(-1)) + "\n")
