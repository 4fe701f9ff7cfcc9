#!/usr/bin/env python3
"""diagnostics: the host-pointer line entry (vrt_plan_execute_line) at C4 size, phase by phase (-DVRT_DIAG build prints them)
and end to end, with a fresh and with a reused (already touched) output array.  usage: python tools/host_entry_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt
from voronoirt_amd import synth, _lib, api
C0 = 2.99792458e8
pos, nbr, bounds = synth.bcc_grid(59, 143, seed=2022)
sites = vrt.VoronoiSites(pos, nbr, bounds, device=0)
n = sites.n
rng = np.random.default_rng(7)
nbb = 51
lambda0 = 121.567e-9
q = np.concatenate([-np.geomspace(600, 0.05, nbb // 2), [0.0], np.geomspace(0.05, 600, nbb // 2)])
lam = lambda0 * (1 + q * 2.5e3 / C0)
z = pos[:, 0]
T = 5e3 + 1.5e4 * (z - bounds[0]) / (bounds[1] - bounds[0])
doppler = lambda0 / C0 * np.sqrt(2 * 1.380649e-23 * T / 1.6735575e-27)
gamma = 4.702e8 + 10 ** rng.uniform(6, 10, n)
velocity = rng.normal(0, 8e3, (n, 3))
strat = np.exp(-(z - bounds[0]) / 0.7e6)
strength = 3e-2 * strat * doppler.mean()
alpha_cont = 1e-4 * strat
S = np.ascontiguousarray(1.0 + rng.random((n, nbb)))
n1 = int(sites.layers_up[1] - 1)
I0 = np.ascontiguousarray(S[sites.perm_up[:n1] - 1])
plan, w = api._quadrature_plan(sites, "ul7n12.dat", 3)
L = _lib.load()
d = api._d
J = np.zeros((n, nbb))
def call(Jout):
    api.check(L.vrt_plan_execute_line(plan._h, nbb, nbb, d(lam), lambda0, C0, d(np.ascontiguousarray(velocity)), d(doppler), d(gamma),
                                      d(np.ascontiguousarray(strength)), d(alpha_cont), d(S), d(I0), None, d(api._f64(w)), d(Jout)))
call(J)
for label, fresh in (("reused output array", False), ("fresh np.zeros output (first touch inside the call)", True)):
    ts = []
    for _ in range(4):
        Jout = np.zeros((n, nbb)) if fresh else J
        t0 = time.perf_counter()
        call(Jout)
        ts.append(time.perf_counter() - t0)
    print(f"{label}: {min(ts) * 1e3:.1f} ms per J (median {sorted(ts)[len(ts) // 2] * 1e3:.1f})", flush=True)
assert np.isfinite(J).all()
