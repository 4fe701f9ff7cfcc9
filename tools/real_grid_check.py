#!/usr/bin/env python3
"""diagnostics: a density-stratified TRUE Voronoi tessellation (the library's own vrt_tessellate,
periodic in x, y) at sizes where the largest BFS layer exceeds the pair kernel's 8 192 sites: which
path runs, how fast, parity of a wavelength sample against the oracle.
usage: python tools/real_grid_check.py [n_sites] [nlam] [scale_height_m]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from voronoirt_amd import _lib, synth  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
nlam = int(sys.argv[2]) if len(sys.argv) > 2 else 24
H = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0e6
bounds = (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6)
rng = np.random.default_rng(11)
u = rng.random(n_sites)
Lz = bounds[1] - bounds[0]
pos = np.stack([bounds[0] - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H))),      # density ~ exp(-z/H): sample_grids.jl:223-230
                bounds[2] + rng.random(n_sites) * (bounds[3] - bounds[2]),
                bounds[4] + rng.random(n_sites) * (bounds[5] - bounds[4])], axis=1)
t0 = time.time()
nbr = vrt.voro(pos, bounds)
print(f"tessellation of {n_sites} sites (vrt_tessellate): {time.time() - t0:.1f} s, D = {nbr.shape[0] - 1}", flush=True)
t0 = time.time()
hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
lu, ld_ = np.diff(hs.layers_up), np.diff(hs.layers_down)
print(f"grid handle {time.time() - t0:.2f} s; layers up {lu.size} (max {lu.max()}), down {ld_.size} (max {ld_.max()})", flush=True)
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
t0 = time.time()
plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
print(f"plan {time.time() - t0:.2f} s, {plan.num_levels} global levels", flush=True)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
n = hs.n
z = torch.as_tensor(pos[:, 0], device=dev)
S = 1 + 0.1 * torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
al = (1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6))[:, None] * (1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64))
n1 = int(hs.layers_up[1] - 1)
I0 = torch.rand((n1, nlam), generator=g, device=dev, dtype=torch.float64)
J = torch.zeros((n, nlam), device=dev, dtype=torch.float64)
stream = torch.cuda.current_stream().cuda_stream
for path in ((None,) if os.environ.get("REAL_GRID_DEFAULT_ONLY") else (None, "levels", "steps", "tiles", "patches")):
    plan.set_option("VRT_PATH", path or "auto")
    if (path == "tiles" and max(lu.max(), ld_.max()) > 8192) or (path == "steps" and max(lu.max(), ld_.max()) > 12288):
        continue
    for _ in range(2):
        plan.execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(), dI0_up=I0.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(), dI0_up=I0.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"path {plan.last_path}: {dt * 1e3:.2f} ms per J ({n * nq * nlam / dt / 1e9:.1f} G cell-updates/s)", flush=True)
plan.set_option("VRT_PATH", "auto")
for _ in range(2):      # J of the default path for the parity check
    plan.execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(), dI0_up=I0.data_ptr(), stream=stream)
torch.cuda.synchronize()
so = orc.make_sites(pos, nbr, bounds)
ls = min(nlam, 4)
ref = orc.J_voronoi(w, th, ph, S[:, :ls].cpu().numpy(), al[:, :ls].cpu().numpy(), so, I0_up=I0[:, :ls].cpu().numpy(), nthreads=8)
got = J[:, :ls].cpu().numpy()
print(f"parity vs oracle on {ls} wavelengths: max rel err {np.abs(got - ref).max() / np.abs(ref).max():.2e}")
