"""The two steps either side of the formal solve that the library also runs on the device
(SURVEY.md 8f rows 2 and 4): the per-angle opacity prologue and the rates / populations epilogue.

Parity status: UNPINNED upstream -- the reference's arithmetic for them goes through Transparency.jl
(absent, version unpinned) and Unitful conversions, and it holds no test for them.  Pinned here:
  * the Voigt profile (Humlíček's w4, the algorithm Transparency.jl documents) against
    scipy.special.wofz on a committed fixture, at w4's own accuracy (1e-4 relative),
  * the oracle's restatement of the reference's formulas against an independent numpy transcription,
  * the HIP kernels against the oracle at 1e-12 (-m gpu)."""
import os

import numpy as np
import pytest
from oracle.parity import rel as _rel

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import _lib, api, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C0 = 2.99792458e8
H_PLANCK = 6.62607015e-34
K_B = 1.380649e-23


def test_voigt_w4_against_wofz_fixture():
    fx = np.load(os.path.join(GOLDEN, "voigt_wofz.npz"))
    got = np.array([orc.humlicek_w4(v, a).real for a, v in zip(fx["a"], fx["v"])])
    rel = np.abs(got - fx["H"]) / fx["H"]
    assert rel.max() < 1e-4, rel.max()             # Humlíček 1982: relative accuracy 1e-4
    # normalisation of the profile: ∫ H(a, v) dv = sqrt(π)
    v = np.linspace(-400, 400, 400001)
    Hv = np.array([orc.humlicek_w4(x, 0.1).real for x in v[::40]])
    assert abs(np.trapezoid(Hv, v[::40]) / np.sqrt(np.pi) - 1) < 2e-3
    assert orc.voigt_profile(0.3, 1.2, 2.5e-12) == orc.humlicek_w4(1.2, 0.3).real / (np.sqrt(np.pi) * 2.5e-12)


def _line_case(n, seed, nbb=51, nbf=20):
    """A Ly-α-like 2-level + continuum atom (src/line.jl:232-247) on n sites, SI numbers."""
    rng = np.random.default_rng(seed)
    lambda0 = 121.567e-9
    # bound-bound sampling like sample_λ_line (log-spaced wings), bound-free blocks linear
    q = np.concatenate([-np.geomspace(600, 0.05, nbb // 2), [0.0], np.geomspace(0.05, 600, nbb // 2)])
    lam_bb = lambda0 * (1 + q * 2.5e3 / C0)
    lam_bf1 = np.linspace(22.8e-9, 91.17e-9, nbf)
    lam_bf2 = np.linspace(91.2e-9, 364.7e-9, nbf)
    lam = np.concatenate([lam_bb, lam_bf1, lam_bf2])
    blocks = np.array([0, nbb, nbb, nbb + nbf, nbb + nbf, nbb + 2 * nbf], dtype=np.int64)
    T = rng.uniform(4e3, 2e4, n)
    m_H = 1.6735575e-27
    doppler = lambda0 / C0 * np.sqrt(2 * K_B * T / m_H)
    gamma = 4.702e8 + 10 ** rng.uniform(6, 10, n)
    velocity = rng.normal(0, 8e3, (n, 3))
    n_i = 10 ** rng.uniform(14, 19, n)
    n_j = n_i * 10 ** rng.uniform(-9, -5, n)
    Bij = 4.5e20
    strength = H_PLANCK * C0 / (4 * np.pi * lambda0) * (n_i * Bij - n_j * Bij * 0.25)
    alpha_cont = 10 ** rng.uniform(-9, -5, n)
    return dict(lambda0=lambda0, lam=lam, blocks=blocks, T=T, doppler=doppler, gamma=gamma, velocity=velocity,
                strength=strength, alpha_cont=alpha_cont, n_i=n_i, n_j=n_j, Bij=Bij, rng=rng)


def test_oracle_line_opacity_against_numpy_transcription():
    c = _line_case(300, 1)
    k = orc.direction(112.8, 335.8)
    got = orc.line_opacity(k, c["lam"][:51], c["lambda0"], C0, c["velocity"], c["doppler"], c["gamma"],
                           c["strength"], c["alpha_cont"])
    from scipy.special import wofz
    v_los = c["velocity"] @ (-k)
    lam = c["lam"][:51][None, :]
    a = c["gamma"][:, None] * lam ** 2 / (4 * np.pi * C0 * c["doppler"][:, None])
    v = (lam - c["lambda0"] + c["lambda0"] * v_los[:, None] / C0) / c["doppler"][:, None]
    ref = c["strength"][:, None] * wofz(v + 1j * a).real / (np.sqrt(np.pi) * c["doppler"][:, None]) + c["alpha_cont"][:, None]
    assert np.abs(got / ref - 1).max() < 1e-4        # w4 vs the exact Faddeeva function


def _rates_case(n, seed):
    c = _line_case(n, seed)
    rng = c["rng"]
    nlam = c["lam"].size
    J = 10 ** rng.uniform(-12, -3, (n, nlam))
    planck2 = 2 * H_PLANCK * C0 ** 2 / c["lam"] ** 5
    lte = np.stack([c["n_i"], c["n_j"] * 3.0, c["n_i"] * 10 ** rng.uniform(-6, 0, n)])       # (3, n) == Julia (n, 3)
    sig1 = 7.9e-22 * (c["lam"][51:71] / c["lam"][70]) ** 3
    sig2 = 1.4e-21 * (c["lam"][71:91] / c["lam"][90]) ** 3
    C = 10 ** rng.uniform(-2, 4, (n, 3, 3))
    for d in range(3):
        C[:, d, d] = 0.0
    atom = lte.sum(axis=0) * rng.uniform(0.9, 1.1, n)
    return c, dict(J=J, planck2=planck2, lte=lte, sig1=sig1, sig2=sig2, C=C, atom=atom,
                   sigma_bb_const=H_PLANCK * C0 / (4 * np.pi * c["lambda0"]) * c["Bij"],
                   hc_over_kB=H_PLANCK * C0 / K_B, pref_ij=2 * np.pi / (H_PLANCK * C0) / 1000.0,
                   pref_ji=2 * np.pi / (H_PLANCK * C0))


def test_oracle_rates_and_populations_against_numpy_transcription():
    c, r = _rates_case(200, 2)
    lam = c["lam"]
    R = orc.calculate_R(lam, c["blocks"], r["J"], r["planck2"], c["lambda0"], C0, c["doppler"], c["gamma"],
                        r["sigma_bb_const"], r["sig1"], r["sig2"], c["T"], r["lte"], r["hc_over_kB"],
                        r["pref_ij"], r["pref_ji"])
    # independent transcription of Rij / Rji (rates.jl:226-364) with numpy's own trapezoid
    def trap(y, x):
        return ((y[:, 1:] + y[:, :-1]) * np.diff(x)[None, :]).sum(axis=1)
    for lev, (lo, hi), sig in ((1, (51, 71), r["sig1"]), (2, (71, 91), r["sig2"])):
        l = lam[lo:hi]
        G = (r["lte"][lev - 1] / r["lte"][2])[:, None] * np.exp(-r["hc_over_kB"] / (l[None, :] * c["T"][:, None]))
        rij = r["pref_ij"] * trap(l[None, :] * sig[None, :] * r["J"][:, lo:hi], l)
        rji = r["pref_ji"] * trap(sig[None, :] * G * l[None, :] * (r["planck2"][None, lo:hi] + r["J"][:, lo:hi]), l)
        assert np.abs(R[:, 2, lev - 1] / rij - 1).max() < 1e-12      # Julia R[lev, 3, i]
        assert np.abs(R[:, lev - 1, 2] / rji - 1).max() < 1e-12      # Julia R[3, lev, i]
    l = lam[:51]
    a = c["gamma"][:, None] * l[None, :] ** 2 / (4 * np.pi * C0 * c["doppler"][:, None])
    v = (l[None, :] - c["lambda0"]) / c["doppler"][:, None]
    H = np.array([[orc.humlicek_w4(v[i, j], a[i, j]).real for j in range(51)] for i in range(v.shape[0])])
    sig = r["sigma_bb_const"] * H / (np.sqrt(np.pi) * c["doppler"][:, None])
    G = (r["lte"][0] / r["lte"][1])[:, None] * np.exp(-r["hc_over_kB"] / (l[None, :] * c["T"][:, None]))
    assert np.abs(R[:, 1, 0] / (r["pref_ij"] * trap(l[None, :] * sig * r["J"][:, :51], l)) - 1).max() < 1e-12
    assert np.abs(R[:, 0, 1] / (r["pref_ji"] * trap(sig * G * l[None, :] * (r["planck2"][None, :51] + r["J"][:, :51]), l)) - 1).max() < 1e-12
    assert (R[:, 0, 0] == 0).all() and (R[:, 1, 1] == 0).all() and (R[:, 2, 2] == 0).all()
    # get_revised_populations (populations.jl:191-221): A x = b per site, with numpy's solver
    pops = orc.revised_populations(R, r["C"], r["atom"])
    P = R + r["C"]                                 # P[i, c, r] == Julia P[r+1, c+1, i+1]
    Pj = lambda rr, cc: P[:, cc - 1, rr - 1]
    A = np.zeros((R.shape[0], 2, 2))
    A[:, 0, 0] = Pj(1, 2) + Pj(2, 1) + Pj(2, 3)
    A[:, 0, 1] = Pj(1, 2) - Pj(3, 2)
    A[:, 1, 1] = Pj(1, 3) + Pj(3, 1) + Pj(3, 2)
    A[:, 1, 0] = Pj(1, 3) - Pj(2, 3)
    b = np.stack([r["atom"] * Pj(1, 2), r["atom"] * Pj(1, 3)], axis=1)
    x = np.linalg.solve(A, b[:, :, None])[:, :, 0]
    assert np.abs(pops[1] / x[:, 0] - 1).max() < 1e-9 and np.abs(pops[2] / x[:, 1] - 1).max() < 1e-9
    assert np.allclose(pops.sum(axis=0), r["atom"], rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("pair_block", [1, 8])
def test_gpu_line_opacity_native_layout_and_sweep(voro_small, pair_block, monkeypatch):
    """vrt_line_opacity_dev writes α_tot of every angle in the native layout; the same α through the
    caller's layout gives the same J bit for bit, and both match the oracle's α / J.  Also with eight
    wavelength pairs of a site side by side (VRT_PAIR_BLOCK=8: 26 pairs in blocks of 8, 8, 8, 2)."""
    import torch
    monkeypatch.setenv("VRT_PAIR_BLOCK", str(pair_block))
    pos, nbr, bounds = voro_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = hs.n
    c = _line_case(n, 3)
    nlam = 51
    lam = c["lam"][:nlam]
    # scale the opacities to the box so that Δτ spans the branches
    scale = 3e4 / c["strength"].max() * c["doppler"].mean()
    strength, alpha_cont = c["strength"] * scale, c["alpha_cont"] * 1e5
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_dop, d_gam, d_str, d_ac = t(c["velocity"]), t(c["doppler"]), t(c["gamma"]), t(strength), t(alpha_cont)
    native = torch.full((plan.native_alpha_count(nlam),), float("nan"), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.line_opacity_dev(lam, c["lambda0"], C0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                          d_str.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=st)
    torch.cuda.synchronize()
    alpha_ref = np.stack([orc.line_opacity(orc.direction(th[a], ph[a]), lam, c["lambda0"], C0, c["velocity"],
                                           c["doppler"], c["gamma"], strength, alpha_cont) for a in range(nq)])
    assert np.isfinite(native.cpu().numpy()).all()
    assert plan.native_pair_block == pair_block
    nat = plan.native_to_site_major(native.cpu().numpy(), nlam, nq)      # [angle][pos][l]
    for a in range(nq):
        order = hs.storage_order(1 if th[a] > 90 else -1) - 1
        assert np.abs(nat[a] / alpha_ref[a][order] - 1).max() < 1e-12, a
    rng = np.random.default_rng(4)
    S = 1 + rng.random((n, nlam))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    Sd, I0d = t(S), t(I0)
    J = torch.empty((n, nlam), dtype=torch.float64, device=dev)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w, dJ=J.data_ptr(),
                     dI0_up=I0d.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    ref = orc.J_voronoi(w, th, ph, S, alpha_ref, so, I0_up=I0, nthreads=4)
    assert _rel(J.cpu().numpy(), ref) < 1e-10
    plan.close()
    hs.close()


@pytest.mark.gpu
def test_gpu_f32_storage_accepts_native_per_angle_alpha(voro_small):
    """fp32 VALUE path with per-angle α: `vrt_line_opacity_dev_f32` (fused prologue) and
    `vrt_plan_alpha_to_native_dev_f32` (caller's (n_angles, n, ld) floats) fill the same native float buffer bit
    for bit, and the sweep over it stays within the fp32 storage tolerance (5e-6) of the fp64 oracle."""
    import torch
    pos, nbr, bounds = voro_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = hs.n
    c = _line_case(n, 3)
    nlam = 51
    lam = c["lam"][:nlam]
    scale = 3e4 / c["strength"].max() * c["doppler"].mean()
    strength, alpha_cont = c["strength"] * scale, c["alpha_cont"] * 1e5
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_dop, d_gam, d_str, d_ac = t(c["velocity"]), t(c["doppler"]), t(c["gamma"]), t(strength), t(alpha_cont)
    count = plan.native_alpha_count(nlam)
    native = torch.full((count,), float("nan"), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.line_opacity_dev(lam, c["lambda0"], C0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                          d_str.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=st, f32=True)
    alpha_ref = np.stack([orc.line_opacity(orc.direction(th[a], ph[a]), lam, c["lambda0"], C0, c["velocity"],
                                           c["doppler"], c["gamma"], strength, alpha_cont) for a in range(nq)])
    ld = nlam + 2
    a32 = np.zeros((nq, n, ld), dtype=np.float32)
    a32[:, :, :nlam] = alpha_ref
    native2 = torch.full((count,), float("nan"), dtype=torch.float32, device=dev)
    a32d = t(a32)
    plan.alpha_to_native_dev(nlam, ld, a32d.data_ptr(), native2.data_ptr(), stream=st, f32=True)
    torch.cuda.synchronize()
    nat, nat2 = native.cpu().numpy(), native2.cpu().numpy()
    assert np.isfinite(nat).all()
    v1 = plan.native_to_site_major(nat, nlam, nq)                  # the pad wavelength of the last pair is unspecified
    v2 = plan.native_to_site_major(nat2, nlam, nq)
    assert np.abs(v1 / v2 - 1).max() < 2.5e-7                      # device fp64 -> float vs numpy fp64 -> float: ≤ 1 ulp
    for a in range(nq):
        order = hs.storage_order(1 if th[a] > 90 else -1) - 1
        assert np.array_equal(v2[a], a32[a][order][:, :nlam]), a
    rng = np.random.default_rng(4)
    S = 1 + rng.random((n, nlam))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    Sd, I0d = t(S.astype(np.float32)), t(I0.astype(np.float32))
    ref = orc.J_voronoi(w, th, ph, S, alpha_ref, so, I0_up=I0, nthreads=4)
    for buf in (native, native2):
        J = torch.full((n, nlam), float("nan"), dtype=torch.float32, device=dev)
        plan.execute_dev(nlam, nlam, Sd.data_ptr(), buf.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w, dJ=J.data_ptr(),
                         dI0_up=I0d.data_ptr(), stream=st, f32=True)
        torch.cuda.synchronize()
        assert plan.last_path == "patches"
        assert _rel(J.cpu().numpy(), ref) < 5e-6
    plan.close()
    hs.close()


@pytest.mark.gpu
def test_gpu_physics_entries_with_growing_wavelength_arrays(voro_small):
    """The wavelength-sized host arrays of the physics entry points go through a device scratch of the grid that
    grows on demand (first capacity 256 doubles): calls with 10, then 400, then 10 wavelengths, on two streams --
    the reallocation must not invalidate the event that orders the scratch's reuse."""
    import torch
    pos, nbr, bounds = voro_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    n = hs.n
    c = _line_case(n, 9)
    w, th, ph, nq = vrt.read_quadrature("ul2n3.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_dop, d_gam, d_str, d_ac = t(c["velocity"]), t(c["doppler"]), t(c["gamma"]), t(c["strength"]), t(c["alpha_cont"])
    side = torch.cuda.Stream()
    for rep, nlam in enumerate((10, 400, 10, 700, 51)):
        lam = c["lambda0"] * (1 + np.linspace(-3e-4, 3e-4, nlam))
        native = torch.full((plan.native_alpha_count(nlam),), float("nan"), dtype=torch.float64, device=dev)
        stream = side if rep % 2 else torch.cuda.current_stream()
        stream.wait_stream(torch.cuda.current_stream())
        plan.line_opacity_dev(lam, c["lambda0"], C0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                              d_str.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        nat = plan.native_to_site_major(native.cpu().numpy(), nlam, nq)
        for a in (0, nq - 1):
            ref = orc.line_opacity(orc.direction(th[a], ph[a]), lam[[0, nlam - 1]], c["lambda0"], C0, c["velocity"],
                                   c["doppler"], c["gamma"], c["strength"], c["alpha_cont"])
            order = hs.storage_order(1 if th[a] > 90 else -1) - 1
            got = nat[a][:, [0, nlam - 1]]
            assert np.abs(got / ref[order] - 1).max() < 1e-12, (nlam, a)
    plan.close()
    hs.close()


@pytest.mark.gpu
def test_gpu_rates_and_populations_against_oracle(bcc_small):
    import torch
    pos, nbr, bounds = bcc_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    n = hs.n
    c, r = _rates_case(n, 5)
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ld = c["lam"].size + 3
    Jd = torch.full((n, ld), float("nan"), dtype=torch.float64, device=dev)
    Jd[:, : c["lam"].size] = t(r["J"])
    d_dop, d_gam, d_T, d_lte, d_C, d_atom = t(c["doppler"]), t(c["gamma"]), t(c["T"]), t(r["lte"]), t(r["C"]), t(r["atom"])
    d_R = torch.empty((n, 3, 3), dtype=torch.float64, device=dev)
    d_pop = torch.empty((3, n), dtype=torch.float64, device=dev)
    api.rates_populations_dev(hs, c["lam"], c["blocks"], ld, Jd.data_ptr(), r["planck2"], c["lambda0"], C0,
                              d_dop.data_ptr(), d_gam.data_ptr(), r["sigma_bb_const"], r["sig1"], r["sig2"],
                              d_T.data_ptr(), d_lte.data_ptr(), r["hc_over_kB"], r["pref_ij"], r["pref_ji"],
                              d_C.data_ptr(), d_atom.data_ptr(), d_R.data_ptr(), d_pop.data_ptr(),
                              stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    R_ref = orc.calculate_R(c["lam"], c["blocks"], r["J"], r["planck2"], c["lambda0"], C0, c["doppler"], c["gamma"],
                            r["sigma_bb_const"], r["sig1"], r["sig2"], c["T"], r["lte"], r["hc_over_kB"],
                            r["pref_ij"], r["pref_ji"])
    R = d_R.cpu().numpy()
    m = R_ref != 0
    assert np.array_equal(R == 0, ~m)
    assert np.abs(R[m] / R_ref[m] - 1).max() < 1e-12
    pops_ref = orc.revised_populations(R, r["C"], r["atom"])          # same R in: isolates the 2 x 2 solve
    assert np.array_equal(d_pop.cpu().numpy(), pops_ref)
    hs.close()


# ---- the whole Λ-iteration, device-resident, against the same loop driven by the oracle --------------
def _lambda_case(pos, bounds, seed):
    """Inputs of Λ_voronoi's loop on n sites: a 2-level + continuum atom with 21 line + 2 x 6 continuum
    wavelengths, magnitudes chosen so that the box is optically thick at line centre and thin in the
    wings, radiative and collisional rates are comparable and the populations stay positive."""
    n = pos.shape[0]
    c = _line_case(n, seed, nbb=21, nbf=6)
    rng = c["rng"]
    lam, nlam = c["lam"], c["lam"].size
    z = (pos[:, 0] - bounds[0]) / (bounds[1] - bounds[0])
    T = 6e3 + 6e3 * z + 200 * rng.random(n)
    doppler = c["lambda0"] / C0 * np.sqrt(2 * K_B * T / 1.6735575e-27)
    n1 = 1e16 * np.exp(-3 * z) * (1 + 0.1 * rng.random(n))
    lte = np.stack([n1, n1 * 1e-3 * (1 + rng.random(n)), n1 * 1e-2 * (1 + rng.random(n))])
    B0 = (1.0 + z)[:, None] * (1 + 0.05 * rng.random((n, nlam)))
    Cm = 10 ** rng.uniform(-1, 1, (n, 3, 3))
    for d in range(3):
        Cm[:, d, d] = 0.0
    box = bounds[1] - bounds[0]
    Bij = 1.0
    strength_const = 60.0 / box * doppler.mean() / n1.mean()       # line-centre τ of order 100 across the box
    return vrt.LineCase(
        lam=lam, blocks=c["blocks"], lambda0=c["lambda0"], c0=C0, velocity=rng.normal(0, 3e3, (n, 3)), doppler=doppler,
        gamma_static=4.702e8 + 10 ** rng.uniform(7, 8.7, n), gamma_unsold=10 ** rng.uniform(-8.5, -7.5, n), alpha_cont=0.05 / box * np.exp(-2 * z), eps=10 ** rng.uniform(-2.5, -0.5, n),
        temperature=T, atom_density=lte.sum(axis=0), B0=B0, lte=lte, C=Cm, planck2=2.0 * (c["lambda0"] / lam) ** 5,
        sigma_bf1=1e-21 * (lam[21:27] / lam[26]) ** 3, sigma_bf2=2e-21 * (lam[27:33] / lam[32]) ** 3,
        strength_const=strength_const, Bij=Bij, Bji=0.25 * Bij, sigma_bb_const=2e-32,
        hc_over_kB=H_PLANCK * C0 / K_B, pref_ij=2e36, pref_ji=2e37)


def _oracle_lambda_iteration(case, so, quadrature, maxiter, eps_conv=0.0):
    """Λ_voronoi's loop (src/lambda_iteration.jl:253-283) with the oracle's restatements."""
    w, th, ph, nq = vrt.read_quadrature(quadrature)
    pops = case.lte.copy()
    S_new, S_old = case.B0.copy(), np.zeros_like(case.B0)
    bottom = so.perm_up[: so.layers_up[1] - 1] - 1
    hist, diff, i = [], np.inf, 0
    while diff > eps_conv and i < maxiter:
        S_old = S_new.copy()
        # γ_constant of the current populations (lambda_iteration.jl:72-75) and αline_λ's population factor
        gamma, strength = orc.line_terms(case.gamma_static, case.gamma_unsold, pops, case.strength_const, case.Bij, case.Bji)
        alpha = np.stack([orc.line_opacity(orc.direction(th[a], ph[a]), case.lam, case.lambda0, case.c0, case.velocity,
                                           case.doppler, gamma, strength, case.alpha_cont) for a in range(nq)])
        J = orc.J_voronoi(w, th, ph, S_old, alpha, so, I0_up=case.B0[bottom], nthreads=8)
        S_new = (1 - case.eps)[:, None] * J + case.eps[:, None] * case.B0
        diff = float(np.abs(1 - S_old / S_new).max())
        R = orc.calculate_R(case.lam, case.blocks, J, case.planck2, case.lambda0, case.c0, case.doppler, gamma,
                            case.sigma_bb_const, case.sigma_bf1, case.sigma_bf2, case.temperature, case.lte,
                            case.hc_over_kB, case.pref_ij, case.pref_ji)
        pops = orc.revised_populations(R, case.C, case.atom_density)
        hist.append(diff)
        i += 1
    return J, S_new, pops, hist


def test_oracle_lambda_iteration_is_well_posed(voro_small):
    """The synthetic line case keeps the loop physical: finite fields, positive populations that sum
    to the atom density, line-centre optical depths across the box above 10 and wing depths below 1,
    and a contracting criterion."""
    pos, nbr, bounds = voro_small
    so = orc.make_sites(pos, nbr, bounds)
    case = _lambda_case(pos, bounds, 11)
    J, S, pops, hist = _oracle_lambda_iteration(case, so, "ul7n12.dat", 3)
    assert np.isfinite(J).all() and np.isfinite(S).all() and (pops > 0).all()
    assert np.allclose(pops.sum(axis=0), case.atom_density, rtol=1e-12)
    assert hist[2] < hist[1] < hist[0]      # (hist[0] is 1/ε - 1 at the never-solved site perm_up[n], where J = 0)
    assert np.abs(pops[1] / case.lte[1] - 1).max() > 1e-3          # the radiation field moved the populations
    strength = case.strength_const * (case.lte[0] * case.Bij - case.lte[1] * case.Bji)
    al = orc.line_opacity(orc.direction(180.0, 0.0), case.lam, case.lambda0, case.c0, case.velocity, case.doppler,
                          case.gamma(case.lte), strength, case.alpha_cont)
    g0, g3 = case.gamma(case.lte), case.gamma(pops)
    assert np.abs(g3 / g0 - 1).max() > 1e-4                        # γ did follow the populations
    assert np.array_equal(orc.line_terms(case.gamma_static, case.gamma_unsold, pops, case.strength_const, case.Bij,
                                         case.Bji)[0], g3)
    tau = al.mean(axis=0) * (bounds[1] - bounds[0])
    assert tau[10] > 10 and tau[0] < 1 and tau[20] < 1


@pytest.mark.gpu
def test_gpu_lambda_voronoi_device_resident_loop(voro_small):
    """vrt.Lambda_voronoi (opacity -> J -> S_new + criterion -> rates + populations, all on the device,
    populations fed back into the opacity) against the same loop driven by the oracle."""
    pos, nbr, bounds = voro_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    case = _lambda_case(pos, bounds, 11)
    J, S, pops, hist = vrt.Lambda_voronoi(0.0, 4, hs, case, "ul7n12.dat")
    J_ref, S_ref, pops_ref, hist_ref = _oracle_lambda_iteration(case, so, "ul7n12.dat", 4)
    assert len(hist) == 4
    assert np.abs(J - J_ref).max() < 1e-9 * np.abs(J_ref).max() and np.abs(S / S_ref - 1).max() < 1e-9   # (J = 0 at perm_up[n])
    assert np.abs(pops / pops_ref - 1).max() < 1e-9
    assert np.allclose(hist, hist_ref, rtol=1e-8)
    # the convergence test stops it like the reference's criterion (lambda_iteration.jl:325-349): it starts
    # from |1 - S_old / S_new| = |1 - 0 / B| = 1, so a tolerance of 1 or more does not iterate at all
    assert vrt.Lambda_voronoi(1.0, 10, hs, case, "ul7n12.dat")[3] == []
    below = [i for i, h in enumerate(hist) if h < 1.0]
    if below:
        eps = hist[below[0]] * 1.0001
        expect = next(i for i, h in enumerate(hist) if h <= eps) + 1
        J2, S2, pops2, hist2 = vrt.Lambda_voronoi(eps, 10, hs, case, "ul7n12.dat")
        assert len(hist2) == expect
        assert np.abs(S2 / _oracle_lambda_iteration(case, so, "ul7n12.dat", expect)[1] - 1).max() < 1e-9
    # the same loop for a host without device arrays: library-owned device state, one call per iteration (the session
    # keeps S and J in sweep order between its steps: no layout change inside the loop, the same numbers)
    Jh, Sh, ph_, hh = vrt.Lambda_voronoi_host(0.0, 4, hs, case, "ul7n12.dat")
    assert np.array_equal(Jh, J) and np.array_equal(Sh, S) and np.array_equal(ph_, pops) and hh == hist
    # ... and through the public sweep-order entry points (vrt_plan_execute_native_dev, vrt_lambda_update_native_dev,
    # vrt_rates_populations_native_dev): bit for bit again
    Jn, Sn, pn, hn = vrt.Lambda_voronoi(0.0, 4, hs, case, "ul7n12.dat", native=True)
    assert np.array_equal(Jn, J) and np.array_equal(Sn, S) and np.array_equal(pn, pops) and hn == hist
    assert vrt.Lambda_voronoi_host(1.0, 10, hs, case, "ul7n12.dat")[3] == []
    hs.close()


@pytest.mark.gpu
def test_gpu_J_line_from_host_arrays(voro_small):
    """vrt_plan_execute_line: the body of J_λ_voronoi's line method (src/lambda_iteration.jl:72-111) from HOST arrays in
    one call -- seven per-site vectors + S in, J out, α_tot (nλ, n, n_angles) made on the device -- against the
    oracle's α_tot + J_λ_voronoi; γ from the given populations."""
    pos, nbr, bounds = voro_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    case = _lambda_case(pos, bounds, 12)
    rng = np.random.default_rng(2)
    pops = case.lte * (1 + 0.2 * rng.random(case.lte.shape))
    S = case.B0 * (1 + 0.3 * rng.random(case.B0.shape))
    J = vrt.J_lambda_voronoi_line(S, pops, hs, case, "ul7n12.dat")
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    gamma, strength = orc.line_terms(case.gamma_static, case.gamma_unsold, pops, case.strength_const, case.Bij, case.Bji)
    alpha = np.stack([orc.line_opacity(orc.direction(th[a], ph[a]), case.lam, case.lambda0, case.c0, case.velocity,
                                       case.doppler, gamma, strength, case.alpha_cont) for a in range(nq)])
    bottom = so.perm_up[: so.layers_up[1] - 1] - 1
    ref = orc.J_voronoi(w, th, ph, S, alpha, so, I0_up=case.B0[bottom], nthreads=8)
    assert np.abs(J - ref).max() < 1e-10 * np.abs(ref).max()
    hs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_gpu_multi_device_lambda_session_and_line_entry(voro_small, devices):
    """vrt_multi_lambda_* (BASELINE configs[3]: Λ-iteration across the devices of a node) and vrt_multi_execute_line:
    wavelength blocks per device handle, every handle makes the α_tot of its own wavelengths and its share of the six
    rate integrals, ONE all-reduce of the shares (6 n doubles; handles on one device: kernel adds), the statistical
    equilibrium solved by every handle.  J and S_new of a block are produced by exactly one handle from replicated
    populations; the rate integrals are regrouped per wavelength (Σ_l W_l f_l instead of the trapezoid segments), so the
    session equals the one-device session to rounding (1e-12) and the oracle-driven loop like it does (1e-9)."""
    pos, nbr, bounds = voro_small
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    case = _lambda_case(pos, bounds, 11)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    dirs = [1 if t > 90 else -1 for t in th]
    mp = vrt.MultiDevicePlan(pos, nbr, bounds, vrt.quadrature_directions(th, ph), dirs=dirs, devices=devices)
    J, S, pops, hist = mp.lambda_iteration(0.0, 4, case, w)
    J1, S1, pops1, hist1 = vrt.Lambda_voronoi_host(0.0, 4, hs, case, "ul7n12.dat")
    assert len(hist) == 4 and np.allclose(hist, hist1, rtol=1e-12)
    assert _rel(J, J1) < 1e-12 and _rel(S, S1) < 1e-12 and _rel(pops, pops1) < 1e-12
    J_ref, S_ref, pops_ref, hist_ref = _oracle_lambda_iteration(case, so, "ul7n12.dat", 4)
    assert np.abs(J - J_ref).max() < 1e-9 * np.abs(J_ref).max() and np.abs(pops / pops_ref - 1).max() < 1e-9
    assert mp.lambda_iteration(1.0, 10, case, w)[3] == []          # the criterion starts at 1 (lambda_iteration.jl:325-349)
    # J_λ_voronoi of the line case from host arrays, wavelength blocks over the handles: each block bit for bit what the
    # one-handle entry gives (every wavelength is solved by exactly one handle)
    rng = np.random.default_rng(2)
    pp = case.lte * (1 + 0.2 * rng.random(case.lte.shape))
    Sx = case.B0 * (1 + 0.3 * rng.random(case.B0.shape))
    Jl = mp.execute_line(Sx, pp, case, w, so.perm_up, int(so.layers_up[1] - 1))
    assert np.array_equal(Jl, vrt.J_lambda_voronoi_line(Sx, pp, hs, case, "ul7n12.dat"))
    mp.close()
    hs.close()


@pytest.mark.gpu
def test_gpu_rccl_calls_execute_on_a_one_rank_communicator(voro_small, monkeypatch):
    """VRT_MULTI_FORCE_RCCL=1: the multi-device object builds a ONE-rank communicator (ncclCommInitAll over device 0), so
    the RCCL legs of csrc/vrt_multi.cpp run on a one-GPU box: ncclReduce of the angle shards' partial J
    (lambda_iteration.jl:102,107) and ncclAllReduce of the six rate-integral shares (rates.jl:154-201) inside group
    calls, stream-ordered behind the sweeps.  A one-rank sum is the identity: results bit for bit those of the same object
    without the communicator."""
    pos, nbr, bounds = voro_small
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    case = _lambda_case(pos, bounds, 11)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    dirs = [1 if t > 90 else -1 for t in th]
    k = vrt.quadrature_directions(th, ph)
    rng = np.random.default_rng(5)
    nlam = 6
    S = 1 + rng.random((n, nlam))
    al = 5 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    I0u = rng.random((so.layers_up[1] - 1, nlam))
    out = {}
    for force in ("0", "1"):
        monkeypatch.setenv("VRT_MULTI_FORCE_RCCL", force)
        mp = vrt.MultiDevicePlan(pos, nbr, bounds, k, dirs=dirs, devices=(0,))
        assert mp.uses_rccl == (force == "1")
        mp.set_shard("angle")                                  # one device holds every angle: ncclReduce(root 0) of its own J
        Ja = mp.execute(S, al, w, I0_up=I0u)
        assert mp.last_shard == "angle"
        mp.set_shard("lambda")
        Jl = mp.execute(S, al, w, I0_up=I0u)
        it = mp.lambda_iteration(0.0, 3, case, w)              # ncclAllReduce of the 6 n shares, every iteration
        out[force] = (Ja, Jl) + tuple(it[:3]) + (np.array(it[3]),)
        mp.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0u, nthreads=4)
    assert _rel(out["1"][0], ref) < 1e-10 and len(out["1"][5]) == 3


@pytest.mark.gpu
def test_gpu_torch_distributed_nccl_world_of_one(tmp_path):
    """bench.py's collectives through torch.distributed's "nccl" backend (= RCCL) with a world of ONE rank
    (VRT_BENCH_FORCE_DIST=1): process-group init, the all-reduce of the angle mode and the all-gather of the wavelength
    blocks execute inside the timed step; J equals the plain one-process run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--workload", "C2", "--nlam", "3", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-secondary", "--no-critical-path"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ref = tmp_path / "ref.npy"
    subprocess.run(base + ["--dump-J", str(ref)], check=True, env=env, capture_output=True, timeout=600)
    for shard in ("angle", "lambda-strong"):
        got = tmp_path / f"{shard}.npy"
        e = dict(env, VRT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run(base + ["--shard", shard, "--dump-J", str(got)], check=True, env=e, capture_output=True, text=True, timeout=600)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["collective"]["backend"].startswith("nccl") and line["collective"]["bytes_per_step"] > 0, line.get("collective")
        assert np.array_equal(np.load(got), np.load(ref))
