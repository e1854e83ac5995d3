"""Throughput of the native loaders (include/rt_loaders.h) on files of the benchmark's size: the synthetic dragon mesh
written as OBJ (871 200 triangles) and the ASCII PPM of a 3840x2160 frame.  Host only.  usage: loader_bench.py [--python]
(--python also times the harness's Python readers, which takes a minute)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import canvas, loaders, obj as pyobj, synth  # noqa: E402


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t)
    return min(ts), out


def main():
    with_python = "--python" in sys.argv
    mesh = synth.synth_dragon()[0]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "dragon.obj")
        pyobj.write_obj(path, mesh["positions"], mesh["normals"], mesh["vidx"], mesh["nidx"])
        data = open(path, "rb").read()
        print(f"OBJ: {len(data) / 1e6:.1f} MB, {len(mesh['vidx'])} triangles, {os.cpu_count()} CPUs", flush=True)
        for threads in (1, 4, 8, 16):
            t, g = best(lambda: loaders.parse_obj_bytes(data, threads))
            print(f"  native parse, {threads:2d} threads: {t * 1e3:8.1f} ms  {len(data) / t / 1e6:8.1f} MB/s", flush=True)
        assert np.array_equal(g[0]["vidx"], mesh["vidx"]) and np.array_equal(g[0]["nidx"], mesh["nidx"])
        if with_python:
            t, _ = best(lambda: pyobj.parse_obj(path), 1)
            print(f"  Python reader:             {t * 1e3:8.1f} ms  {len(data) / t / 1e6:8.1f} MB/s", flush=True)
        rng = np.random.RandomState(1)
        frame = rng.randint(0, 256, size=(2160, 3840, 3)).astype(np.uint8)
        for threads in (1, 4, 8, 16):
            t, text = best(lambda: loaders.ppm_encode(frame, threads))
            print(f"PPM 3840x2160 encode, {threads:2d} threads: {t * 1e3:8.1f} ms  {len(text) / t / 1e6:8.1f} MB/s of text "
                  f"({len(text) / 1e6:.1f} MB)", flush=True)
        t, _ = best(lambda: loaders.ppm_write(os.path.join(d, "f.ppm"), frame, 0))
        print(f"PPM write to a file, auto threads: {t * 1e3:8.1f} ms", flush=True)
        t, img = best(lambda: loaders.ppm_decode(text))
        print(f"PPM P3 decode of that text:        {t * 1e3:8.1f} ms", flush=True)
        assert np.array_equal(img, frame.astype(np.float32) / np.float32(255))
        if with_python:
            t, _ = best(lambda: canvas.write_ppm(os.path.join(d, "p.ppm"), frame.astype(np.float32) / 255), 1)
            print(f"Python writer:                     {t * 1e3:8.1f} ms", flush=True)


if __name__ == "__main__":
    main()
