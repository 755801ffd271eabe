import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the hyphen-named package directory as module ``ppg_slam_amd``."""
    if "ppg_slam_amd" in sys.modules:
        return sys.modules["ppg_slam_amd"]
    path = os.path.join(ROOT, "ppg-slam_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("ppg_slam_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ppg_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    load_pkg()
    import importlib
    return importlib.import_module("ppg_slam_amd.synth")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def onet(orc):
    return orc.Net()


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def assert_keypoints_equivalent(xy_a, sc_a, xy_b, sc_b, tol=2e-5):
    """Two key-point lists computed by two correct fp32 CNN evaluations: the sets must be equal, and
    positions may differ only where the competing scores are within `tol` (the order of near-ties is
    decided by rounding noise; two correct fp32 runs differ by ~2e-6 in score)."""
    a = [tuple(int(v) for v in p) for p in xy_a]
    b = [tuple(int(v) for v in p) for p in xy_b]
    assert len(a) == len(b), (len(a), len(b))
    sa, sb = set(a), set(b)
    if sa != sb:
        # points at the cap boundary may swap in/out: they must be the lowest-scoring ones
        only_a = [sc_a[i] for i, p in enumerate(a) if p not in sb]
        only_b = [sc_b[i] for i, p in enumerate(b) if p not in sa]
        assert len(only_a) <= 2 and all(abs(v - min(sc_a)) < tol for v in only_a) and all(abs(v - min(sc_b)) < tol for v in only_b), \
            (len(only_a), len(only_b))
    bad = [i for i in range(len(a)) if a[i] != b[i] and abs(float(sc_a[i]) - float(sc_b[i])) > tol]
    assert not bad, bad[:5]
    return sum(x != y for x, y in zip(a, b))


def edge_coord_set(edge_idx, xy):
    """Edges as unordered pairs of end-point pixel coordinates (independent of key-point numbering)."""
    out = set()
    for s, e in edge_idx:
        p, q = tuple(int(v) for v in xy[int(s)]), tuple(int(v) for v in xy[int(e)])
        out.add((min(p, q), max(p, q)))
    return out
