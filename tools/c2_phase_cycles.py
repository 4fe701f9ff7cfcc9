import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["VRT_TILE_DEBUG"] = "1"
import torch
import voronoirt_amd as vrt
from voronoirt_amd import synth, _lib
pos, nbr, bounds = synth.bcc_grid(37, 90, seed=1998)
hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
n = hs.n
rng = np.random.default_rng(1)
S = 1 + rng.random((n, 1)); al = (1e-2 * np.exp(-(pos[:, 0] - bounds[0]) / 0.7e6))[:, None] * (1 + rng.random((n, 1)))
for _ in range(3):
    J, _ = plan.execute(S, al, weights=w)
print(plan.last_path)
