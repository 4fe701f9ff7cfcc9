#!/usr/bin/env python3
"""How much of the result hangs on the ORDER of a cell's neighbours in the neighbour file?

voro++ prints a cell's neighbours in its own face order (`con.print_custom("%i %n", ...)`,
rt_preprocessing/output_sites.cc:49) and the reference's top-2 upwind rule is order dependent
(src/voronoi_utils.jl:378-386: a neighbour that was the running maximum when visited can never become the SECOND
upwind).  vrt_tessellate lists walls first, then neighbours by distance -- another order.  This probe shuffles every
row `shuffles` times and reports, against the unshuffled rows: the fraction of (site, angle) whose first / second
upwind changes, and the largest relative change of J (element-wise and max-norm).  CPU only (the oracle).
usage: python tools/row_order_probe.py [golden | strat:<n>] [shuffles] [quadrature]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import voronoirt_amd as vrt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.parity import rel  # noqa: E402
from voronoirt_amd import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "golden"
shuffles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
quad = sys.argv[3] if len(sys.argv) > 3 else "ul7n12.dat"
if what == "golden":
    G = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(G, "voro2k_meta.json")))
    sites = np.loadtxt(os.path.join(G, "voro2k_sites.txt"))
    pos = np.ascontiguousarray(sites[:, [3, 1, 2]])
    bounds = tuple(meta["bounds"])
    rows = [[int(v) for v in line.split()] for line in open(os.path.join(G, "voro2k_neighbours.txt"))]
    n = pos.shape[0]
    D = max(len(r) - 1 for r in rows)
    nbr = np.zeros((D + 1, n), dtype=np.int64)
    for r in rows:
        nbr[0, r[0] - 1] = len(r) - 1
        nbr[1:len(r), r[0] - 1] = r[1:]
else:
    n = int(what.split(":")[1])
    H = 2.0e6
    bounds = (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6)
    rng = np.random.default_rng(11)
    u = rng.random(n)
    Lz = bounds[1] - bounds[0]
    pos = np.stack([bounds[0] - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H))),
                    bounds[2] + rng.random(n) * (bounds[3] - bounds[2]),
                    bounds[4] + rng.random(n) * (bounds[5] - bounds[4])], axis=1)
    nbr = vrt.voro(pos, bounds)
w, th, ph, nq = vrt.read_quadrature(quad)
S, al = synth.synthetic_fields(pos, bounds, 1, seed=9)
threads = int(os.environ.get("PROBE_THREADS", "8"))


def solve(M):
    so = orc.make_sites(pos, M, bounds)
    ups = np.stack([orc.upwind_table(so, orc.direction(t, p))[0] for t, p in zip(th, ph)])      # (angles, n, 2)
    I0 = S[so.perm_up[: so.layers_up[1] - 1] - 1]
    J = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=threads)
    return ups, J


ups0, J0 = solve(nbr)
rng = np.random.default_rng(2022)
first = second = 0.0
worst = worst_mx = 0.0
for s in range(shuffles):
    M = nbr.copy()
    for i in range(n):
        c = int(M[0, i])
        M[1:c + 1, i] = rng.permutation(M[1:c + 1, i])
    ups, J = solve(M)
    f1 = float((ups[:, :, 0] != ups0[:, :, 0]).mean())
    f2 = float((ups[:, :, 1] != ups0[:, :, 1]).mean())
    e = rel(J, J0)
    first += f1 / shuffles
    second += f2 / shuffles
    worst = max(worst, float(e))
    worst_mx = max(worst_mx, e.maxnorm)
    print(f"shuffle {s}: first upwind changed {f1:.4%}, second {f2:.4%} of (site, angle); J changed by {e}", flush=True)
print(f"{what}: {n} sites, {nq} angles, {shuffles} shuffles of every row: first upwind {first:.4%}, second upwind "
      f"{second:.4%} of (site, angle) change on average; largest change of J {worst:.3e} element-wise, {worst_mx:.3e} max-norm")
