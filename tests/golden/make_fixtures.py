#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container; needs scipy for the
Voronoi tessellation).  Inputs are written in the reference's own file formats -- voro++ site
file "id\\tx\\ty\\tz" (src/io.jl:16-20) and voro++ "%i %n" neighbour file
(rt_preprocessing/output_sites.cc:49) -- and the expected outputs come from the CPU oracle
(oracle/vrt_oracle.c), the literal restatement of the reference algorithm.  The reference itself
(Julia) cannot be executed in this image, so these vectors pin the build's own oracle and
kernels against regressions; they are not outputs of the reference ("parity unpinned")."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc          # noqa: E402
from voronoirt_amd import synth           # noqa: E402
from voronoirt_amd.api import read_quadrature  # noqa: E402


def main():
    n, seed = 2000, 20221003
    bounds = (-0.5, 2.5, 0.0, 1.2, 0.0, 1.0)
    pos, nbr, bounds = synth.voronoi_grid(n, seed=seed, bounds=bounds, scale_height=1.0)
    synth.write_sites_file(os.path.join(HERE, "voro2k_sites.txt"), pos)
    synth.write_neighbours_file(os.path.join(HERE, "voro2k_neighbours.txt"), nbr, seed=seed)
    # re-read what was written so the expected values belong to the committed text exactly
    sites_txt = np.loadtxt(os.path.join(HERE, "voro2k_sites.txt"))
    pos = np.ascontiguousarray(sites_txt[:, [3, 1, 2]])
    so = orc.read_cell(os.path.join(HERE, "voro2k_neighbours.txt"), n, pos, bounds)

    rng = np.random.default_rng(seed)
    nlam = 3
    S = 1.0 + rng.random((n, nlam))
    alpha = 10.0 ** rng.uniform(-3, 3, (n, 1)) * (1.0 + rng.random((n, nlam))) * 10.0
    n1u, n1d = int(so.layers_up[1] - 1), int(so.layers_down[1] - 1)
    I0_up = rng.random((n1u, nlam))
    I0_down = rng.random((n1d, nlam))
    w, th, ph, _ = read_quadrature("ul7n12.dat")
    out = {"layers_up": so.layers_up, "layers_down": so.layers_down, "perm_up": so.perm_up,
           "perm_down": so.perm_down, "S": S, "alpha": alpha, "I0_up": I0_up, "I0_down": I0_down}
    for a in (1, 2, 3, 4):          # two inclined/steep up rays, two down rays
        k = orc.direction(th[a], ph[a])
        up, dots, wt, r, st = orc.upwind_table(so, k)
        out[f"up_{a}"] = up
        out[f"status_{a}"] = st
        if th[a] > 90:
            out[f"I_{a}"] = orc.Delaunay_upII(k, S[:, 0], I0_up[:, 0], alpha[:, 0], so, 3)
        else:
            out[f"I_{a}"] = orc.Delaunay_downII(k, S[:, 0], I0_down[:, 0], alpha[:, 0], so, 3)
    out["J"] = orc.J_voronoi(w, th, ph, S, alpha, so, I0_up=I0_up, I0_down=I0_down)
    np.savez_compressed(os.path.join(HERE, "voro2k_expected.npz"), **out)
    json.dump({"n": n, "seed": seed, "bounds": list(bounds), "nlam": nlam, "quadrature": "ul7n12.dat",
               "angles": [1, 2, 3, 4], "n_sweeps": 3,
               "generator": "tests/golden/make_fixtures.py (synth.voronoi_grid + oracle)"},
              open(os.path.join(HERE, "voro2k_meta.json"), "w"), indent=1)
    print("wrote fixtures:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
