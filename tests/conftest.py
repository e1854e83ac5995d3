import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def obj_files():
    return [os.path.join(ROOT, "scenes", "obj", f) for f in ("space_ship.obj", "teapot_low.obj")]


def scene_path(name):
    return os.path.join(ROOT, "scenes", f"{name}.yaml")


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", f"{name}.png")


def small_dragons(nu=64, nv=48):
    """dragons.yaml (6 instances, nested transformed groups, `divide: 10`, transparent shadow-less
    bbox cubes) over a small synthetic stand-in for dragon.obj: 2*nu*nv triangles per instance."""
    from raytracer_amd import synth
    from raytracer_amd.scene import Scene
    return Scene.from_file(scene_path("dragons"), meshes={"dragon.obj": synth.synth_dragon(nu, nv)})
