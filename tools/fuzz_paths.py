#!/usr/bin/env python3
"""Randomised cross-path screen (GPU): random small grids (jittered BCC lattices of odd shapes, true
Voronoi tessellations), wavelength counts, opacity layouts and direction subsets; J and the
per-angle intensities of the step and tile paths must agree with the level path to 1e-12 and with
the CPU oracle to 1e-10.  usage: python tools/fuzz_paths.py [cases] [seed] [big]
(big = 1: lattices of 34..62 cells across, i.e. layers of 2 300..7 700 sites, 3..8 sites per thread)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from voronoirt_amd import synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3 and sys.argv[3] == "1"
quads = ["ul7n12.dat", "ul9n20.dat", "ul2n3.dat", "n1.dat", "n2.dat"]
worst = 0.0
for case in range(cases):
    if big or rng.random() < 0.6:
        a, c = (int(rng.integers(34, 63)), int(rng.integers(2, 4))) if big else (int(rng.integers(3, 34)), int(rng.integers(2, 7)))
        pos, nbr, bounds = synth.bcc_grid(a, c, seed=int(rng.integers(1 << 30)))
        desc = f"bcc a={a} c={c}"
    else:
        nsites = int(rng.integers(300, 4000))
        pos, nbr, bounds = synth.voronoi_grid(nsites, int(rng.integers(1 << 30)),
                                               scale_height=None if rng.random() < 0.5 else 0.4)[:3]
        desc = f"voronoi n={nsites}"
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    w, th, ph, nq = vrt.read_quadrature(quads[int(rng.integers(len(quads)))])
    nlam = int(rng.choice([1, 2, 3, 5, 8, 13]))
    mode = int(rng.integers(3))
    S = 1 + rng.random((n, nlam))
    base = 10 ** rng.uniform(-8, -4, (n, 1))
    if mode == 0:
        al = base[:, 0].copy()
    elif mode == 1:
        al = base * (1 + rng.random((n, nlam)))
    else:
        al = base[None] * (1 + rng.random((nq, n, nlam)))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    n_sweeps = int(rng.choice([1, 2, 3, 3, 3, 4]))
    ks = vrt.quadrature_directions(th, ph)
    res = {}
    for path in ("levels", "steps", "tiles", "patches"):
        os.environ["VRT_PATH"] = path
        plan = vrt.FormalPlan(hs, ks, n_sweeps)
        J, I = plan.execute(S, al, weights=w, I0_up=I0, want_I=True)
        res[path] = (J.copy(), I.copy())
        plan.close()
    # the patch path's launch forms -- per-layer launches, chained with progress words, chained with the intensities as
    # their own flags -- against whatever the default chose: bit for bit
    os.environ["VRT_PATH"] = "patches"
    for form, env in (("launches", {"VRT_PATCH_CHAIN": "0"}), ("chain", {"VRT_PATCH_CHAIN": "1", "VRT_CHAIN_DATAFLAG": "0"}),
                      ("chain-df", {"VRT_PATCH_CHAIN": "1", "VRT_CHAIN_DATAFLAG": "1"})):
        os.environ.update(env)
        plan = vrt.FormalPlan(hs, ks, n_sweeps)
        Jf, If = plan.execute(S, al, weights=w, I0_up=I0, want_I=True)
        plan.close()
        for k in env:
            os.environ.pop(k)
        if not (np.array_equal(Jf, res["patches"][0]) and np.array_equal(If, res["patches"][1])):
            print(f"case {case} {desc}: patch path form '{form}' differs from the default form")
            sys.exit(1)
    # the single-wavelength level kernel (large layers / fp32 storage) on the same problem: bitwise the pair kernel
    os.environ["VRT_PATH"] = "steps"
    os.environ["VRT_STEP_SINGLE"] = "1"
    plan = vrt.FormalPlan(hs, ks, n_sweeps)
    J1, I1 = plan.execute(S, al, weights=w, I0_up=I0, want_I=True)
    plan.close()
    os.environ.pop("VRT_STEP_SINGLE")
    if not (np.array_equal(J1, res["steps"][0]) and np.array_equal(I1, res["steps"][1])):
        print(f"case {case} {desc}: single-wavelength level kernel differs from the pair kernel")
        sys.exit(1)
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, n_sweeps=n_sweeps, nthreads=4)
    scale = max(np.abs(ref).max(), 1e-300)
    e_or = np.abs(res["levels"][0] - ref).max() / scale
    e_st = max(np.abs(res["steps"][0] - res["levels"][0]).max() / scale,
               np.abs(res["steps"][1] - res["levels"][1]).max() / max(np.abs(res["levels"][1]).max(), 1e-300))
    e_ti = max(np.abs(res["tiles"][0] - res["levels"][0]).max() / scale,
               np.abs(res["tiles"][1] - res["levels"][1]).max() / max(np.abs(res["levels"][1]).max(), 1e-300))
    e_pa = max(np.abs(res["patches"][0] - res["levels"][0]).max() / scale,
               np.abs(res["patches"][1] - res["levels"][1]).max() / max(np.abs(res["levels"][1]).max(), 1e-300))
    worst = max(worst, e_or, e_st, e_ti, e_pa)
    ok = e_or < 1e-10 and e_st < 1e-12 and e_ti < 1e-12 and e_pa < 5e-12
    print(f"case {case:3d} {desc:18s} n={n:6d} quad={nq:2d} nlam={nlam:2d} alpha_mode={mode} sweeps={n_sweeps} "
          f"oracle {e_or:.1e} steps {e_st:.1e} tiles {e_ti:.1e} patches {e_pa:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
    hs.close()
os.environ.pop("VRT_PATH", None)
print(f"fuzz ok: {cases} cases, worst relative difference {worst:.2e}")
