#!/usr/bin/env python3
"""diagnostics: one device-resident Λ-iteration of the line case at C4 size (995 566 sites, ul7n12,
51 line + 2 x 20 continuum wavelengths), entry point by entry point ("Keeping a
Λ-iteration on the device"):

    vrt_line_opacity_dev -> vrt_plan_execute_dev (native α) -> vrt_lambda_update_dev -> vrt_rates_populations_dev

HIP-event time of each call on the launch stream + the bytes each one has to move (its own floor).
Synthetic inputs with physical magnitudes (tests/test_physics.py's Ly-α-like atom).

    python tools/iteration_breakdown.py [--a 59 --c 143] [--reps 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt                     # noqa: E402
from voronoirt_amd import _lib, api, synth      # noqa: E402

C0, H_PLANCK, K_B = 2.99792458e8, 6.62607015e-34, 1.380649e-23

ap = argparse.ArgumentParser()
ap.add_argument("--a", type=int, default=59)
ap.add_argument("--c", type=int, default=143)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--velocity", default="noise", choices=["noise", "smooth"],
                help="noise: independent normal 8 km/s per SITE (the field of every earlier round: neighbouring sites of a wave fall into "
                     "different regions of the Voigt function's four-region form); smooth: 8 km/s of large-scale flow (a few Fourier modes of "
                     "the box) + 0.5 km/s of noise, as a simulated atmosphere's velocity field is")
args = ap.parse_args()

dev = torch.device("cuda", 0)
pos, nbr, bounds = synth.bcc_grid(args.a, args.c, seed=2022)
sites = vrt.VoronoiSites(pos, nbr, bounds, device=0)
n = sites.n
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
plan = vrt.FormalPlan(sites, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])

rng = np.random.default_rng(7)
nbb, nbf = 51, 20
lambda0 = 121.567e-9
q = np.concatenate([-np.geomspace(600, 0.05, nbb // 2), [0.0], np.geomspace(0.05, 600, nbb // 2)])
lam = np.concatenate([lambda0 * (1 + q * 2.5e3 / C0), np.linspace(22.8e-9, 91.17e-9, nbf), np.linspace(91.2e-9, 364.7e-9, nbf)])
blocks = np.array([0, nbb, nbb, nbb + nbf, nbb + nbf, nbb + 2 * nbf], dtype=np.int64)
nlam_all = lam.size
z = pos[:, 0]
T = (5e3 + 1.5e4 * (z - bounds[0]) / (bounds[1] - bounds[0])) * (1 + 0.05 * rng.random(n))   # smooth in height, as an atmosphere is
doppler = lambda0 / C0 * np.sqrt(2 * K_B * T / 1.6735575e-27)
gamma = 4.702e8 + 10 ** rng.uniform(6, 10, n)
velocity = rng.normal(0, 8e3, (n, 3))
if args.velocity == "smooth":
    xyz = (pos - np.array([bounds[0], bounds[2], bounds[4]])) / np.array([bounds[1] - bounds[0], bounds[3] - bounds[2], bounds[5] - bounds[4]])
    velocity = rng.normal(0, 0.5e3, (n, 3))
    for comp in range(3):
        for _ in range(4):
            kvec = rng.integers(1, 4, 3)
            velocity[:, comp] += 4e3 * np.sin(2 * np.pi * (xyz @ kvec) + rng.uniform(0, 2 * np.pi))
strat = np.exp(-(z - bounds[0]) / 0.7e6)
strength = 3e-2 * strat * doppler.mean() * (1 + 0.1 * rng.random(n))     # Δτ between neighbours spans the branches
alpha_cont = 1e-4 * strat
lte = np.stack([10 ** rng.uniform(14, 19, n), 10 ** rng.uniform(8, 12, n), 10 ** rng.uniform(10, 16, n)])
Cmat = 10 ** rng.uniform(-2, 4, (n, 3, 3))
for d in range(3):
    Cmat[:, d, d] = 0.0
atom = lte.sum(axis=0)
planck2 = 2 * H_PLANCK * C0 ** 2 / lam ** 5
sig1 = 7.9e-22 * (lam[51:71] / lam[70]) ** 3
sig2 = 1.4e-21 * (lam[71:91] / lam[90]) ** 3

t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d_vel, d_dop, d_gam, d_str, d_ac = t(velocity), t(doppler), t(gamma), t(strength), t(alpha_cont)
d_T, d_lte, d_C, d_atom = t(T), t(lte), t(Cmat), t(atom)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
S = 1.0 + torch.rand((n, nbb), generator=gen, device=dev, dtype=torch.float64)
B = 1.0 + torch.rand((n, nbb), generator=gen, device=dev, dtype=torch.float64)
eps = 1e-3 + 0.1 * torch.rand((n, nbb), generator=gen, device=dev, dtype=torch.float64)
S_new = torch.empty_like(S)
J = torch.zeros((n, nlam_all), device=dev, dtype=torch.float64)          # line block first, continuum blocks as given
J[:, nbb:] = 1e-6 * torch.rand((n, 2 * nbf), generator=gen, device=dev, dtype=torch.float64)
n1 = int(sites.layers_up[1] - 1)
I0 = S[torch.as_tensor(sites.perm_up[:n1] - 1, device=dev)].contiguous()
native = torch.empty(plan.native_alpha_count(nbb), device=dev, dtype=torch.float64)
d_R = torch.empty((n, 3, 3), device=dev, dtype=torch.float64)
d_pop = torch.empty((3, n), device=dev, dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream


def opacity():
    plan.line_opacity_dev(lam[:nbb], lambda0, C0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                          d_str.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=st)


def solve():
    # J of the line block into the first nbb columns of the (n, 91) array (ld = 91)
    plan.execute_dev(nbb, nbb, S.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w, dJ=Jline.data_ptr(),
                     dI0_up=I0.data_ptr(), stream=st)


def update():
    api.lambda_update_dev(sites, nbb, nbb, Jline.data_ptr(), B.data_ptr(), eps.data_ptr(), S.data_ptr(),
                          S_new.data_ptr(), stream=st)


def rates():
    api.rates_populations_dev(sites, lam, blocks, nlam_all, J.data_ptr(), planck2, lambda0, C0, d_dop.data_ptr(),
                              d_gam.data_ptr(), H_PLANCK * C0 / (4 * np.pi * lambda0) * 4.5e20, sig1, sig2,
                              d_T.data_ptr(), d_lte.data_ptr(), H_PLANCK * C0 / K_B, 2 * np.pi / (H_PLANCK * C0) / 1000.0,
                              2 * np.pi / (H_PLANCK * C0), d_C.data_ptr(), d_atom.data_ptr(), d_R.data_ptr(),
                              d_pop.data_ptr(), stream=st)


Jline = torch.zeros((n, nbb), device=dev, dtype=torch.float64)
A = nq
steps = (
    ("vrt_line_opacity_dev", opacity, 8.0 * (A * n * (nbb + 1) + 8 * n)),               # writes α of every angle
    ("vrt_plan_execute_dev (J)", solve, float(n) * A * nbb * (40.0 + 40.0 / nbb)),      # §8d algorithmic bytes
    ("vrt_lambda_update_dev", update, 8.0 * 5 * n * nbb),                               # J, B, ε, S_old in, S_new out
    ("vrt_rates_populations_dev", rates, 8.0 * (n * nlam_all + 30 * n)),               # J in; R, populations, C out
)
print(f"{n} sites, {A} angles, {nbb} line + {2 * nbf} continuum wavelengths")
total = 0.0
for name, fn, nbytes in steps:
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    total += ms
    print(f"{name:28s} {ms:8.3f} ms   {nbytes / 1e9:7.2f} GB it must move -> {nbytes / ms / 1e6:7.1f} GB/s of those")
print(f"{'one iteration':28s} {total:8.3f} ms   (path {plan.last_path}; S and J in the caller's (n, nlam) layout)")
assert torch.isfinite(Jline).all() and torch.isfinite(d_pop).all()

# ---- the same iteration with S and J kept in SWEEP ORDER between the steps (what vrt_lambda_iterate does) ------------------
cnt = plan.native_plane_count(nbb)
S_nat = [torch.empty(cnt, device=dev, dtype=torch.float64) for _ in range(2)]
B_up = torch.empty(cnt, device=dev, dtype=torch.float64)
Jn = [torch.zeros(cnt, device=dev, dtype=torch.float64) for _ in range(2)]
plan.to_native_dev(nbb, nbb, S.data_ptr(), S_nat[0].data_ptr(), S_nat[1].data_ptr(), stream=st)
plan.to_native_dev(nbb, nbb, B.data_ptr(), B_up.data_ptr(), 0, stream=st)
eps_site = eps[:, 0].contiguous()
cnt_all = plan.native_plane_count(nlam_all)
J_all = [torch.zeros(cnt_all, device=dev, dtype=torch.float64) for _ in range(2)]
plan.to_native_dev(nlam_all, nlam_all, J.data_ptr(), J_all[0].data_ptr(), 0, stream=st)      # (the continuum part of J: as given)


def solve_native():
    plan.execute_native_dev(nbb, S_nat[0].data_ptr(), S_nat[1].data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w,
                            dJ_up=Jn[0].data_ptr(), dJ_down=Jn[1].data_ptr(), dI0_up=I0.data_ptr(), stream=st)


def update_native():
    api.lambda_update_native_dev(sites, nbb, Jn[0].data_ptr(), Jn[1].data_ptr(), B_up.data_ptr(), eps_site.data_ptr(),
                                 S_nat[0].data_ptr(), S_nat[1].data_ptr(), stream=st)


def rates_native():
    api.rates_populations_native_dev(sites, lam, blocks, J_all[0].data_ptr(), J_all[1].data_ptr(), planck2, lambda0, C0,
                                     d_dop.data_ptr(), d_gam.data_ptr(), H_PLANCK * C0 / (4 * np.pi * lambda0) * 4.5e20, sig1, sig2,
                                     d_T.data_ptr(), d_lte.data_ptr(), H_PLANCK * C0 / K_B, 2 * np.pi / (H_PLANCK * C0) / 1000.0,
                                     2 * np.pi / (H_PLANCK * C0), d_C.data_ptr(), d_atom.data_ptr(), d_R.data_ptr(),
                                     d_pop.data_ptr(), stream=st)


steps_n = (
    ("vrt_line_opacity_dev", opacity, 8.0 * (A * n * (nbb + 1) + 8 * n)),
    ("vrt_plan_execute_native_dev", solve_native, float(n) * A * nbb * (40.0 + 40.0 / nbb)),
    ("vrt_lambda_update_native_dev", update_native, 8.0 * (6 * n * nbb + n)),              # J_up, J_down, B, S_old in; S_up, S_down out
    ("vrt_rates_populations_native", rates_native, 8.0 * (2 * n * nlam_all + 30 * n)),
)
total_n = 0.0
for name, fn, nbytes in steps_n:
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    total_n += ms
    print(f"{name:28s} {ms:8.3f} ms   {nbytes / 1e9:7.2f} GB it must move -> {nbytes / ms / 1e6:7.1f} GB/s of those")
print(f"{'one iteration':28s} {total_n:8.3f} ms   (path {plan.last_path}; S and J in sweep order: no layout change inside the loop)")
assert all(torch.isfinite(x).all() for x in Jn) and torch.isfinite(d_pop).all()
plan.close()

# ---- the same from HOST arrays (what the Julia shim calls): PCIe-inclusive wall times ------------------
import time  # noqa: E402
case = vrt.LineCase(lam=lam, blocks=blocks, lambda0=lambda0, c0=C0, velocity=velocity, doppler=doppler,
                    gamma_static=gamma, gamma_unsold=1e-9 * np.ones(n), alpha_cont=alpha_cont,
                    eps=10 ** rng.uniform(-2.5, -0.5, n), temperature=T, atom_density=atom,
                    B0=(1.0 + (z - bounds[0]) / (bounds[1] - bounds[0]))[:, None] * np.ones((1, nlam_all)), lte=lte, C=Cmat,
                    planck2=planck2, sigma_bf1=sig1, sigma_bf2=sig2,
                    strength_const=float(np.median(strength / lte[0])), Bij=1.0, Bji=0.25,
                    sigma_bb_const=H_PLANCK * C0 / (4 * np.pi * lambda0) * 4.5e20, hc_over_kB=H_PLANCK * C0 / K_B,
                    pref_ij=2 * np.pi / (H_PLANCK * C0) / 1000.0, pref_ji=2 * np.pi / (H_PLANCK * C0))
S_h = np.ascontiguousarray(case.B0[:, :nbb])
line_only = vrt.LineCase(**{**{k: getattr(case, k) for k in vrt.LineCase.FIELDS}, "lam": lam[:nbb], "B0": S_h})
vrt.J_lambda_voronoi_line(S_h, lte, sites, line_only, "ul7n12.dat")            # warm-up: plan, staging buffers
# the C entry itself, as the Julia shim calls it: its arguments prepared before the clock starts (γ and the line
# strength of the populations are the caller's numpy / Julia work either way)
L0 = _lib.load()
plan_l, wq_l = api._quadrature_plan(sites, "ul7n12.dat", 3)
gam_h = api._f64(line_only.gamma(lte))
str_h = api._f64(line_only.strength_const * (lte[0] * line_only.Bij - lte[1] * line_only.Bji))
vel_h, dop_h, ac_h, lam_h = api._f64(velocity), api._f64(doppler), api._f64(alpha_cont), api._f64(lam[:nbb])
n1h = int(sites.layers_up[1] - 1)
I0_h = api._f64(S_h[sites.perm_up[:n1h] - 1])
Jh = np.zeros((n, nbb))
def line_call():
    api.check(L0.vrt_plan_execute_line(plan_l._h, nbb, nbb, api._d(lam_h), float(lambda0), float(C0), api._d(vel_h), api._d(dop_h),
                                       api._d(gam_h), api._d(str_h), api._d(ac_h), api._d(S_h), api._d(I0_h), None,
                                       api._d(api._f64(wq_l)), api._d(Jh)))
line_call()
t0 = time.perf_counter()
for _ in range(3):
    line_call()
dt = (time.perf_counter() - t0) / 3
up = 8.0 * (n * nbb + 7 * n) / 1e9
print(f"vrt_plan_execute_line, host arrays in -> J out ({nbb} wavelengths): {dt * 1e3:.1f} ms per J "
      f"(uploads {up:.2f} GB + downloads {8.0 * n * nbb / 1e9:.2f} GB; alpha_tot (nlam, n, n_angles) would be {8.0 * n * nbb * A / 1e9:.1f} GB)")
assert np.isfinite(Jh).all()
t0 = time.perf_counter()
Jl, Sl, pl, hist = vrt.Lambda_voronoi_host(0.0, 4, sites, case, "ul7n12.dat")
dt = time.perf_counter() - t0
print(f"vrt_lambda_create + 4 x vrt_lambda_iterate + vrt_lambda_get ({nlam_all} wavelengths): {dt * 1e3:.0f} ms in all; history {hist}")
L = _lib.load()
import ctypes  # noqa: E402
plan_h, wq = api._quadrature_plan(sites, "ul7n12.dat", 3)
lc, keep = case.c_struct()
h = ctypes.c_void_p()
api.check(L.vrt_lambda_create(plan_h._h, ctypes.byref(lc), api._d(api._f64(wq)), ctypes.byref(h)))
d = ctypes.c_double()
api.check(L.vrt_lambda_iterate(h, ctypes.byref(d)))
t0 = time.perf_counter()
for _ in range(3):
    api.check(L.vrt_lambda_iterate(h, ctypes.byref(d)))
print(f"vrt_lambda_iterate: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per Λ-iteration at {nlam_all} wavelengths (only the criterion's scalar crosses PCIe)")
L.vrt_lambda_destroy(h)
sites.close()
