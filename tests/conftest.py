import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def cornell():
    from pt_amd.scene import Scene
    return Scene.load_npz(os.path.join(GOLDEN, "scene_cornell.npz"))


@pytest.fixture(scope="session")
def cube():
    from pt_amd.scene import Scene
    return Scene.load_npz(os.path.join(GOLDEN, "scene_cube.npz"))


@pytest.fixture(scope="session")
def helmet():
    """FlightHelmet geometry (5 instances, 94,722 triangles) with reduced base-colour textures."""
    from pt_amd.scene import Scene
    return Scene.load_npz(os.path.join(GOLDEN, "scene_helmet.npz"))


@pytest.fixture(scope="session")
def dragon():
    from pt_amd import procedural
    return procedural.dragon_standin()


@pytest.fixture(scope="session")
def soup():
    from pt_amd import procedural
    return procedural.triangle_soup(6000, seed=3, n_instances=4)


@pytest.fixture(scope="session")
def stacked():
    from pt_amd import procedural
    return procedural.stacked_copies()
