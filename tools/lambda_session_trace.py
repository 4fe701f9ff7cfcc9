#!/usr/bin/env python3
"""The library-owned Λ-iteration session (vrt_lambda_create / _iterate) at C4 size, a few iterations -- run under
`rocprofv3 --kernel-trace --stats` to list the kernels an iteration launches (profiles/r5/lambda_session_kernel_stats.csv:
no k_to_sweep_order, no k_combine_J between the steps: S and J stay in sweep order).  Prints the time per iteration."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt                     # noqa: E402
from voronoirt_amd import _lib, api, synth      # noqa: E402

C0, H_PLANCK, K_B = 2.99792458e8, 6.62607015e-34, 1.380649e-23
a, c = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (59, 143)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pos, nbr, bounds = synth.bcc_grid(a, c, seed=2022)
sites = vrt.VoronoiSites(pos, nbr, bounds, device=0)
n = sites.n
rng = np.random.default_rng(7)
nbb, nbf = 51, 20
lambda0 = 121.567e-9
q = np.concatenate([-np.geomspace(600, 0.05, nbb // 2), [0.0], np.geomspace(0.05, 600, nbb // 2)])
lam = np.concatenate([lambda0 * (1 + q * 2.5e3 / C0), np.linspace(22.8e-9, 91.17e-9, nbf), np.linspace(91.2e-9, 364.7e-9, nbf)])
blocks = np.array([0, nbb, nbb, nbb + nbf, nbb + nbf, nbb + 2 * nbf], dtype=np.int64)
z = pos[:, 0]
T = (5e3 + 1.5e4 * (z - bounds[0]) / (bounds[1] - bounds[0])) * (1 + 0.05 * rng.random(n))
doppler = lambda0 / C0 * np.sqrt(2 * K_B * T / 1.6735575e-27)
gamma = 4.702e8 + 10 ** rng.uniform(6, 10, n)
velocity = rng.normal(0, 8e3, (n, 3))
strat = np.exp(-(z - bounds[0]) / 0.7e6)
strength = 3e-2 * strat * doppler.mean() * (1 + 0.1 * rng.random(n))
lte = np.stack([10 ** rng.uniform(14, 19, n), 10 ** rng.uniform(8, 12, n), 10 ** rng.uniform(10, 16, n)])
Cmat = 10 ** rng.uniform(-2, 4, (n, 3, 3))
for d in range(3):
    Cmat[:, d, d] = 0.0
case = vrt.LineCase(lam=lam, blocks=blocks, lambda0=lambda0, c0=C0, velocity=velocity, doppler=doppler,
                    gamma_static=gamma, gamma_unsold=1e-9 * np.ones(n), alpha_cont=1e-4 * strat,
                    eps=10 ** rng.uniform(-2.5, -0.5, n), temperature=T, atom_density=lte.sum(axis=0),
                    B0=(1.0 + (z - bounds[0]) / (bounds[1] - bounds[0]))[:, None] * np.ones((1, lam.size)), lte=lte, C=Cmat,
                    planck2=2 * H_PLANCK * C0 ** 2 / lam ** 5, sigma_bf1=7.9e-22 * (lam[51:71] / lam[70]) ** 3,
                    sigma_bf2=1.4e-21 * (lam[71:91] / lam[90]) ** 3,
                    strength_const=float(np.median(strength / lte[0])), Bij=1.0, Bji=0.25,
                    sigma_bb_const=H_PLANCK * C0 / (4 * np.pi * lambda0) * 4.5e20, hc_over_kB=H_PLANCK * C0 / K_B,
                    pref_ij=2 * np.pi / (H_PLANCK * C0) / 1000.0, pref_ji=2 * np.pi / (H_PLANCK * C0))
L = _lib.load()
plan, wq = api._quadrature_plan(sites, "ul7n12.dat", 3)
lc, keep = case.c_struct()
h = ctypes.c_void_p()
api.check(L.vrt_lambda_create(plan._h, ctypes.byref(lc), api._d(api._f64(wq)), ctypes.byref(h)))
d = ctypes.c_double()
api.check(L.vrt_lambda_iterate(h, ctypes.byref(d)))
t0 = time.perf_counter()
hist = []
for _ in range(iters):
    api.check(L.vrt_lambda_iterate(h, ctypes.byref(d)))
    hist.append(d.value)
dt = (time.perf_counter() - t0) / iters
print(f"vrt_lambda_iterate: {dt * 1e3:.2f} ms per iteration at {lam.size} wavelengths, {n} sites, 12 angles "
      f"(VRT_LAMBDA_NATIVE={os.environ.get('VRT_LAMBDA_NATIVE', '1')}); criterion {hist}")
L.vrt_lambda_destroy(h)
sites.close()
