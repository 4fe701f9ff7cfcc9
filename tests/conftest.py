import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    # the product library and the oracle are built in-tree; (re)build when sources are newer
    from voronoirt_amd import build as vbuild
    from oracle import oracle as orc
    vbuild.build_library()
    orc.build()


_build_once()


@pytest.fixture(scope="session")
def bcc_small():
    from voronoirt_amd import synth
    return synth.bcc_grid(8, 12, seed=2)


@pytest.fixture(scope="session")
def voro_small():
    from voronoirt_amd import synth
    return synth.voronoi_grid(1500, seed=5, bounds=(0.0, 2.0, 0.0, 1.0, 0.0, 1.0), scale_height=0.7)


@pytest.fixture(scope="session")
def golden():
    """Committed fixture: a 2000-site true Voronoi grid in the reference's file formats plus
    oracle outputs (tests/golden/make_fixtures.py)."""
    import json
    meta = json.load(open(os.path.join(GOLDEN, "voro2k_meta.json")))
    exp = np.load(os.path.join(GOLDEN, "voro2k_expected.npz"))
    sites = np.loadtxt(os.path.join(GOLDEN, "voro2k_sites.txt"))
    # file columns: id x y z  (src/io.jl:16-20) -> positions rows (z, x, y)
    pos = np.ascontiguousarray(sites[:, [3, 1, 2]])
    return {"meta": meta, "exp": exp, "pos": pos,
            "nbr_file": os.path.join(GOLDEN, "voro2k_neighbours.txt"),
            "bounds": tuple(meta["bounds"])}


def random_fields(n, nlam, seed, box=1.0):
    rng = np.random.default_rng(seed)
    S = 1.0 + rng.random((n, nlam))
    alpha = 10.0 ** rng.uniform(-3, 3, (n, 1)) * (1.0 + rng.random((n, nlam))) * (10.0 / box)
    return S, alpha
