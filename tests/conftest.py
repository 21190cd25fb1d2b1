import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sio():
    from path_tracing_amd import scene_io
    return scene_io


@pytest.fixture(scope="session")
def input_scene(sio):
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    return sc, sio.flatten_for_pt(sc)


@pytest.fixture(scope="session")
def hpt():
    """The product library; GPU tests fail loudly when libhpt.so is missing."""
    import path_tracing_amd
    path_tracing_amd.load_library()
    return path_tracing_amd


def scene_by_name(sio, name):
    if name in ("input", "mis_test"):
        sc = sio.load_scene(os.path.join(GOLDEN, "scenes", name + ".txt"))
        return sio.flatten_for_pt(sc), (sc.eye, sc.look_at, sc.view_up)
    if name == "cornell_diffuse":
        return sio.cornell_diffuse(), (sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP)
    if name == "cornell_sphere_2k":
        return sio.cornell_with_sphere(2000), (sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP)
    raise KeyError(name)


def golden_cases():
    out = []
    for f in sorted(os.listdir(GOLDEN)):
        if f.startswith("pt_") and f.endswith(".npz"):
            out.append(f)
    return out


def rmse(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt((d * d).mean()))
