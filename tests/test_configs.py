"""Every configuration of BASELINE.json under `pytest -m gpu`, by name, against the oracle:

  C1  compare_searchlight.jl's regular-grid searchlight, 60^3, n1.dat (θ = 180°, ϕ = 0°)
  C2  ~250k-site Voronoi (246 420), ul7n12 x 1 λ -- the whole J against the oracle
  C3  ~1M sites, ul9n20 x 20 λ, shared α -- a sample of directions x wavelengths + properties
  C4  ~1M sites, ul7n12 x 51 λ, PER-ANGLE α (caller layout and the native layout)
  C5  4M sites (4 011 544), ul9n20 x 100 λ, fp32 storage -- properties + up / down single solves at
      wavelengths 1 and 97 against the fp64 oracle at 5e-6
  +   the in-process tessellation at 250 000 and 1 000 000 density-stratified sites

Full-size parity uses the same seeded inputs on both sides; where the oracle would take minutes
the comparison is on a sample of (direction, wavelength) problems -- every problem is independent
(lambda_iteration.jl:84-111) -- plus size-independent properties (linearity in (S, I_0),
determinism, bounds)."""
import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import _lib, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-10        # north star, fp64
RTOL_F32 = 5e-6     # fp32 storage against the fp64 oracle


from oracle.parity import rel as _rel     # element-wise: |a - b| < tol (|b| + smallest non-zero |b|) for EVERY element


def _fields(pos, bounds, nlam, seed, alpha0=1e-2):
    rng = np.random.default_rng(seed)
    n = pos.shape[0]
    z = pos[:, 0]
    S = 1.0 + 0.5 * np.sin(2 * np.pi * (z - bounds[0]) / (bounds[1] - bounds[0]))[:, None] + 0.1 * rng.random((n, nlam))
    al = (alpha0 * np.exp(-(z - bounds[0]) / 0.7e6))[:, None] * (1.0 + 0.1 * rng.random((n, nlam)))
    return S, al


# ---- C1 ---------------------------------------------------------------------------------------------
def test_C1_searchlight_regular_60_cubed_n1():
    """searchlight_regular (src/compare_searchlight.jl:154-225) at 60^3 with quadratures/n1.dat:
    α = S = 0, a disk of radius 0.1 lit on the bottom plane, the single vertical up ray.  Known
    answer from the code: every plane repeats I_0 (the ray hits grid points exactly, α = 0 gives
    e = 1, a = b = 0), so I_top == I_0 and Σ I is conserved ("Bottom ... Top ...", :209)."""
    w, th, ph, nq = vrt.read_quadrature("n1.dat")
    assert nq == 1 and th[0] == 180.0 and ph[0] == 0.0
    n = 60
    z = x = y = np.linspace(0, 1, n)
    S = np.zeros((n, n, n))
    al = np.zeros((n, n, n))
    I0 = np.zeros((n, n))
    for i in range(1, n + 1):                   # compare_searchlight.jl:180-190
        for j in range(1, n + 1):
            if np.sqrt((i / n - 0.5) ** 2 + (j / n - 0.5) ** 2) < 0.1:
                I0[j - 1, i - 1] = 1.0
    k = vrt.direction(th[0], ph[0])
    I = vrt.short_characteristics_up(k, S, I0, al, z, x, y, 3)          # (ny, nx, nz)
    ref = orc.short_characteristics_up(orc.direction(th[0], ph[0]), S, I0, al, z, x, y, 3)
    assert np.abs(I - ref).max() <= 1e-15
    top = I[1:-1, 1:-1, -1]                                              # [end, 2:end-1, 2:end-1]
    assert np.array_equal(top, I0[1:-1, 1:-1])
    assert abs(top.sum() - I0[1:-1, 1:-1].sum()) == 0.0 and top.sum() > 100


# ---- C2 ---------------------------------------------------------------------------------------------
def test_C2_continuum_250k_full_J():
    a, c = synth.BCC_CONFIGS["C2"]
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=1998)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    assert hs.n == 246420
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    S, al = _fields(pos, bounds, 1, 11)
    I0 = S[so.perm_up[: so.layers_up[1] - 1] - 1]
    # the continuum caller: one wavelength, α per site (lambda_continuum.jl:27-56)
    J = vrt.J_lambda_voronoi(S, al[:, 0].copy(), hs, "ul7n12.dat", I0_up=I0)
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    ref = orc.J_voronoi(w, th, ph, S, al[:, 0].copy(), so, I0_up=I0, nthreads=8)
    assert _rel(J, ref) < RTOL
    # upwind ids of one inclined angle, bit-exact at this size
    plan = vrt.FormalPlan(hs, [vrt.direction(th[0], ph[0])], 3, dirs=[-1])
    up, dots, ww, r = plan.upwind(0)
    up_o, dots_o, w_o, r_o, _ = orc.upwind_table(so, orc.direction(th[0], ph[0]))
    assert np.array_equal(up, up_o) and np.array_equal(dots, dots_o) and np.array_equal(r, r_o)
    plan.close()
    hs.close()


# ---- the ~1M-site grid of C3 / C4 ------------------------------------------------------------------
@pytest.fixture(scope="module")
def grid_1m():
    a, c = synth.BCC_CONFIGS["C4"]
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=2022)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    yield hs, so, pos, bounds
    hs.close()


def test_C3_1M_ul9n20_20_wavelengths(grid_1m):
    import torch
    hs, so, pos, bounds = grid_1m
    n, nlam = hs.n, 20
    assert n == 995566
    w, th, ph, nq = vrt.read_quadrature("ul9n20.dat")
    assert nq == 20
    S, al = _fields(pos, bounds, nlam, 3)
    n1 = int(so.layers_up[1] - 1)
    I0 = S[so.perm_up[:n1] - 1].copy()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    dev = torch.device("cuda", 0)
    Sd, Ad, I0d = (torch.from_numpy(x).to(dev) for x in (S, al, I0))
    Iout = torch.empty((nq, n, nlam), dtype=torch.float64, device=dev)
    Jd = torch.empty((n, nlam), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                     dI0_up=I0d.data_ptr(), dI_out=Iout.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    # a sample: 2 directions (one up, one down) x 2 wavelengths against the oracle
    ups = [i for i in range(nq) if th[i] > 90]
    downs = [i for i in range(nq) if th[i] < 90]
    for a_i, l in ((ups[3], 0), (ups[3], 13), (downs[7], 5), (downs[7], 19)):
        k = orc.direction(th[a_i], ph[a_i])
        if th[a_i] > 90:
            ref = orc.Delaunay_upII(k, S[:, l].copy(), I0[:, l].copy(), al[:, l].copy(), so, 3)
        else:
            ref = orc.Delaunay_downII(k, S[:, l].copy(), np.zeros(so.layers_down[1] - 1), al[:, l].copy(), so, 3)
        assert _rel(Iout[a_i, :, l].cpu().numpy(), ref) < RTOL, (a_i, l)
    # J is the weighted sum of the per-angle intensities (angle order of the reference)
    Jsum = torch.zeros_like(Jd)
    for i in range(nq):
        Jsum += w[i] * Iout[i]
    assert (Jsum - Jd).abs().max().item() <= 1e-13 * Jd.abs().max().item()
    # determinism
    J2 = torch.empty_like(Jd)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J2.data_ptr(),
                     dI0_up=I0d.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(J2, Jd)
    plan.close()


def test_C4_1M_per_angle_alpha_51_wavelengths(grid_1m):
    """Full size, per-angle α, 51 λ: a 2-wavelength sample of J against the oracle (all 12 angles),
    through the caller's (n_angles, n, nλ) layout AND the native layout -- bitwise equal results."""
    import torch
    hs, so, pos, bounds = grid_1m
    n, nlam = hs.n, 51
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    z = torch.as_tensor(pos[:, 0], device=dev)
    S = 1.0 + 0.5 * torch.sin(2 * np.pi * (z - bounds[0]) / (bounds[1] - bounds[0]))[:, None] \
        + 0.1 * torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
    strat = 1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6)
    lam = torch.arange(nlam, device=dev, dtype=torch.float64)
    alpha = torch.empty((nq, n, nlam), device=dev, dtype=torch.float64)
    for a_i in range(nq):
        psi = 1.0 + 9.0 * torch.exp(-((lam - 25.0 - 1.3 * np.cos(2 * np.pi * a_i / nq)) / 8.5) ** 2)
        alpha[a_i] = strat[:, None] * (1.0 + 0.1 * torch.rand((n, nlam), generator=g, device=dev,
                                                             dtype=torch.float64)) * psi[None, :]
    n1 = int(so.layers_up[1] - 1)
    I0 = S[torch.as_tensor(so.perm_up[:n1] - 1, device=dev)].contiguous()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    st = torch.cuda.current_stream().cuda_stream
    J = torch.empty((n, nlam), dtype=torch.float64, device=dev)
    plan.execute_dev(nlam, nlam, S.data_ptr(), alpha.data_ptr(), _lib.ALPHA_ANGLE_SITE_LAM, w,
                     dJ=J.data_ptr(), dI0_up=I0.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    native = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
    plan.alpha_to_native_dev(nlam, nlam, alpha.data_ptr(), native.data_ptr(), stream=st)
    Jn = torch.empty_like(J)
    plan.execute_dev(nlam, nlam, S.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w,
                     dJ=Jn.data_ptr(), dI0_up=I0.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(J, Jn)
    # ... and with S and J in the sweep's own order (vrt_plan_execute_native_dev: what the device-resident Λ-iteration and
    # bench.py's headline step run): J_up + J_down is the same J bit for bit at full size
    cnt = plan.native_plane_count(nlam)
    S_up, S_dn, J_up, J_dn = (torch.empty(cnt, dtype=torch.float64, device=dev) for _ in range(4))
    plan.to_native_dev(nlam, nlam, S.data_ptr(), S_up.data_ptr(), S_dn.data_ptr(), stream=st)
    plan.execute_native_dev(nlam, S_up.data_ptr(), S_dn.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w,
                            dJ_up=J_up.data_ptr(), dJ_down=J_dn.data_ptr(), dI0_up=I0.data_ptr(), stream=st)
    plan.J_from_native_dev(nlam, nlam, J_up.data_ptr(), J_dn.data_ptr(), Jn.data_ptr(), stream=st)
    torch.cuda.synchronize()
    plan.check()
    assert plan.last_path == "patches" and torch.equal(J, Jn)
    del S_up, S_dn, J_up, J_dn
    # the native layout is what the header says: pair q = l/2 of the block [q0, q0 + w) it falls in, at storage
    # position pos of angle a, is pair element q0 n + pos w + (q - q0) (26 pairs in blocks of B = 8: 8, 8, 8, 2)
    B = plan.native_pair_block
    npad = 52
    for a_i, l in ((7, 33), (2, 50), (11, 0)):
        order = hs.storage_order(1 if th[a_i] > 90 else -1) - 1
        q = l // 2
        widths = [B] * (26 // B) + [1 << b for b in range(B.bit_length() - 2, -1, -1) if (26 % B) & (1 << b)]
        q0 = 0
        for wd in widths:
            if q < q0 + wd:
                break
            q0 += wd
        blk = native[a_i * npad * n:(a_i + 1) * npad * n].view(-1, 2)[q0 * n:(q0 + wd) * n].view(n, wd, 2)
        assert torch.equal(blk[:, q - q0, l % 2], alpha[a_i, torch.as_tensor(order, device=dev), l])
    sel = [0, 37]
    ref = orc.J_voronoi(w, th, ph, S[:, sel].cpu().numpy(), alpha[:, :, sel].cpu().numpy(), so,
                        I0_up=I0[:, sel].cpu().numpy(), nthreads=8)
    assert _rel(J[:, sel].cpu().numpy(), ref) < RTOL
    plan.close()


# ---- C5 ---------------------------------------------------------------------------------------------
def test_C5_4M_sites_fp32_storage_100_wavelengths():
    """BASELINE configs[4] at its full size: 4 011 544 sites x ul9n20 x 100 wavelengths, fp32 storage (S, α, J
    1.6 GB each, the per-angle intensities 32 GB).  Fields are generated on the device; the oracle solves one up
    and one down ray at wavelength indices either side of the 64-wavelength knee (1 and 97) on the float32-
    rounded inputs; determinism, bounds and linearity at full size."""
    import torch
    a, c = synth.BCC_CONFIGS["C5"]
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=1998)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    n = hs.n
    assert n == 4011544
    w, th, ph, nq = vrt.read_quadrature("ul9n20.dat")
    nlam = 100
    so = orc.make_sites(pos, nbr, bounds)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    z = torch.as_tensor(pos[:, 0], device=dev, dtype=torch.float32)
    Sd = (1.0 + 0.5 * torch.sin(2 * np.pi * (z - bounds[0]) / (bounds[1] - bounds[0]))[:, None]
          + 0.1 * torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float32)).contiguous()
    Ad = ((1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6))[:, None]
          * (1.0 + 0.1 * torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float32))).contiguous()
    n1 = int(so.layers_up[1] - 1)
    I0d = Sd[torch.as_tensor(so.perm_up[:n1] - 1, device=dev)].contiguous()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    st = torch.cuda.current_stream().cuda_stream

    def solve(Sx, I0x, want_I=False):
        J = torch.empty((n, nlam), dtype=torch.float32, device=dev)
        Io = torch.empty((nq, n, nlam), dtype=torch.float32, device=dev) if want_I else None
        plan.execute_dev(nlam, nlam, Sx.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(),
                         dI0_up=I0x.data_ptr(), dI_out=Io.data_ptr() if want_I else 0, stream=st, f32=True)
        torch.cuda.synchronize()
        return J, Io

    J1, Io = solve(Sd, I0d, want_I=True)
    assert plan.last_path == "patches"
    up_i = next(i for i in range(nq) if th[i] > 90 and th[i] < 130)
    dn_i = next(i for i in range(nq) if th[i] < 90 and th[i] > 50)
    for l, a_i in ((1, up_i), (97, dn_i), (97, up_i)):
        S64, al64 = (x[:, l].cpu().numpy().astype(np.float64) for x in (Sd, Ad))      # what the device was given
        k = orc.direction(th[a_i], ph[a_i])
        if th[a_i] > 90:
            ref = orc.Delaunay_upII(k, S64, I0d[:, l].cpu().numpy().astype(np.float64), al64, so, 3)
        else:
            ref = orc.Delaunay_downII(k, S64, np.zeros(so.layers_down[1] - 1), al64, so, 3)
        assert _rel(Io[a_i, :, l].cpu().numpy().astype(np.float64), ref) < RTOL_F32, (l, a_i)
    del Io
    torch.cuda.empty_cache()
    # properties at full size: determinism, bounds (convex combinations of S and I_0), linearity
    J2, _ = solve(Sd, I0d)
    assert torch.equal(J1, J2)
    assert J1.min().item() >= 0.0 and J1.max().item() <= float(Sd.max().item()) * (1 + 1e-5)
    J3, _ = solve(Sd * 2.0, I0d * 2.0)
    assert ((J3 - 2.0 * J1).abs().max() / J3.abs().max()).item() < 1e-5
    del J2, J3
    # ... and the same step with S, alpha and J in sweep order (vrt_plan_execute_native_dev_f32: no layout change inside): bit for bit
    cnt = plan.native_plane_count(nlam)
    S_nat = [torch.empty(cnt, dtype=torch.float32, device=dev) for _ in range(2)]
    J_nat = [torch.full((cnt,), -1.0, dtype=torch.float32, device=dev) for _ in range(2)]
    A_nat = torch.empty(2 * cnt, dtype=torch.float32, device=dev)
    plan.to_native_dev(nlam, nlam, Sd.data_ptr(), S_nat[0].data_ptr(), S_nat[1].data_ptr(), stream=st, f32=True)
    plan.to_native_dev(nlam, nlam, Ad.data_ptr(), A_nat.data_ptr(), A_nat.data_ptr() + 4 * cnt, stream=st, f32=True)
    plan.execute_native_dev(nlam, S_nat[0].data_ptr(), S_nat[1].data_ptr(), A_nat.data_ptr(), _lib.ALPHA_SITE_LAM_NATIVE, w,
                            dJ_up=J_nat[0].data_ptr(), dJ_down=J_nat[1].data_ptr(), dI0_up=I0d.data_ptr(), stream=st, f32=True)
    Jn = torch.empty((n, nlam), dtype=torch.float32, device=dev)
    plan.J_from_native_dev(nlam, nlam, J_nat[0].data_ptr(), J_nat[1].data_ptr(), Jn.data_ptr(), stream=st, f32=True)
    torch.cuda.synchronize()
    assert torch.equal(Jn, J1)
    plan.close()
    hs.close()


# ---- in-process tessellation -> device path --------------------------------------------
def _stratified_sites(n, seed, H=2.0e6):
    """density ~ exp(-z / H) like sample_from_invNH_invT (src/sample_grids.jl:223-230) in the Bifrost-sized box"""
    bounds = (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6)
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    Lz = bounds[1] - bounds[0]
    pos = np.stack([bounds[0] - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H))),
                    bounds[2] + rng.random(n) * (bounds[3] - bounds[2]),
                    bounds[4] + rng.random(n) * (bounds[5] - bounds[4])], axis=1)
    return pos, bounds


def test_tessellated_250k_stratified_sites_full_J():
    """`voro` (rt_preprocessing/output_sites.cc:35-49, here vrt_tessellate) -> VoronoiSites -> J_λ_voronoi on
    250 000 density-stratified sites (layers of up to ~11 000 sites): ul7n12 x 4 wavelengths, the whole J
    against the oracle on the default device path."""
    pos, bounds = _stratified_sites(250000, 11)
    nbr = vrt.voro(pos, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    assert max(np.diff(hs.layers_up).max(), np.diff(hs.layers_down).max()) > 8192      # beyond the pair-tile kernels
    S, al = _fields(pos, bounds, 4, 21)
    I0 = S[so.perm_up[: so.layers_up[1] - 1] - 1]
    J = vrt.J_lambda_voronoi(S, al, hs, "ul7n12.dat", I0_up=I0)
    assert list(hs._plans.values())[0].last_path == "patches"
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
    assert _rel(J, ref) < RTOL
    hs.close()


def test_tessellated_1M_stratified_sites_sample_of_directions():
    """The reference's production shape (compare_line.jl:64-68: 1 050 232 sites sampled ~ N_H / T): 1 000 000
    density-stratified sites tessellated in-process, BFS layers of ~30 000 sites -- beyond every
    whole-layer kernel, so the layers are cut into patches.  One up and one down direction x 2 wavelengths
    against the oracle; J = Σ w I; per-angle alpha in the native layout is accepted."""
    import torch
    pos, bounds = _stratified_sites(1000000, 12)
    nbr = vrt.voro(pos, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    assert max(np.diff(hs.layers_up).max(), np.diff(hs.layers_down).max()) > 18432
    n, nlam = hs.n, 6
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    S, al = _fields(pos, bounds, nlam, 5)
    n1 = int(so.layers_up[1] - 1)
    I0 = S[so.perm_up[:n1] - 1].copy()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    dev = torch.device("cuda", 0)
    Sd, Ad, I0d = (torch.from_numpy(x).to(dev) for x in (S, al, I0))
    Iout = torch.empty((nq, n, nlam), dtype=torch.float64, device=dev)
    Jd = torch.empty((n, nlam), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                     dI0_up=I0d.data_ptr(), dI_out=Iout.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    ups = [i for i in range(nq) if th[i] > 90]
    downs = [i for i in range(nq) if th[i] < 90]
    for a_i, l in ((ups[1], 0), (downs[4], 5)):
        k = orc.direction(th[a_i], ph[a_i])
        if th[a_i] > 90:
            ref = orc.Delaunay_upII(k, S[:, l].copy(), I0[:, l].copy(), al[:, l].copy(), so, 3)
        else:
            ref = orc.Delaunay_downII(k, S[:, l].copy(), np.zeros(so.layers_down[1] - 1), al[:, l].copy(), so, 3)
        assert _rel(Iout[a_i, :, l].cpu().numpy(), ref) < RTOL, (a_i, l)
    Jsum = torch.zeros_like(Jd)
    for i in range(nq):
        Jsum += w[i] * Iout[i]
    assert (Jsum - Jd).abs().max().item() <= 1e-13 * Jd.abs().max().item()
    # the native per-angle alpha layout works on such a grid (the device-resident Λ-iteration's input)
    native = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
    A3 = Ad[None].expand(nq, n, nlam).contiguous()
    plan.alpha_to_native_dev(nlam, nlam, A3.data_ptr(), native.data_ptr(), stream=st)
    Jn = torch.empty_like(Jd)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w, dJ=Jn.data_ptr(),
                     dI0_up=I0d.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(Jn, Jd)
    plan.close()
    hs.close()


# ---- N > 1 code path of bench.py, rehearsed on one GPU ------------------------------------------------
@pytest.mark.parametrize("shard", ["lambda-strong", "angle"])
def test_bench_two_ranks_reproduce_the_one_rank_J(tmp_path, shard):
    """`python bench.py --gpus 2` launches its own two ranks (torch.distributed.run, before any GPU
    call), shards the FIXED tiny problem by wavelength blocks (all-gather of J) or by angles
    (all-reduce of J) and must reproduce the single-rank J.  VRT_BENCH_REHEARSE=1: both ranks share
    GPU 0 and the collectives run over gloo (RCCL refuses two ranks on one device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VRT_BENCH_REHEARSE="1")
    env.pop("VRT_PATH", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    outs = {}
    for gpus in (1, 2):
        dump = tmp_path / f"J_{gpus}.npy"
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--workload", "tiny",
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-critical-path",
               "--shard", shard, "--dump-J", str(dump)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        outs[gpus] = (json.loads(line), np.load(dump))
    (j1, J1), (j2, J2) = outs[1], outs[2]
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert j2["config"]["shard"] == shard and J1.shape == J2.shape
    c = j2["collective"]                                      # the collective is inside the timed step, and visible
    assert c["bytes_per_step"] == J1.size * 8 and c["ms"] > 0 and ("all_gather" if shard == "lambda-strong" else "all_reduce") in c["op"]
    # same cell-update count for the fixed job at either rank count
    assert abs(j2["value"] * j2["ms_per_step"] - j1["value"] * j1["ms_per_step"]) < 1e-6 * j1["value"] * j1["ms_per_step"]
    if shard == "lambda-strong":
        assert np.array_equal(J1, J2)            # every wavelength is solved by exactly one rank
    else:
        assert np.abs(J1 - J2).max() <= 1e-13 * np.abs(J1).max()   # the all-reduce changes the summation order


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_default_is_the_fixed_workload(launcher):
    """The driver's scaling run: `bench.py --gpus 2` with the DEFAULT sharding -- the fixed workload split by
    wavelength blocks, J all-gathered inside the timed step (north star: RCCL on J each step) -- started by
    bench.py itself or, as the driver does, under `python -m torch.distributed.run`.  One JSON line from rank 0,
    n_gpus 2, scaling "strong", the one-rank cell-update count per step; `--shard lambda` is the opt-in weak mode."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VRT_BENCH_REHEARSE="1")
    env.pop("VRT_PATH", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
            "--no-critical-path"]
    bench = os.path.join(root, "bench.py")
    r1 = subprocess.run([sys.executable, bench, "--gpus", "1"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    if launcher == "self":
        cmd = [sys.executable, bench, "--gpus", "2"] + args
    else:
        import socket
        with socket.socket() as sk:                           # a free port for the rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), bench, "--gpus", "2"] + args
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    j1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["steps"] == 2 and j2["warmup"] == 1
    assert j2["config"]["shard"] == "lambda-strong" and "all_gather" in j2["collective"]["op"]
    u1, u2 = j1["value"] * j1["ms_per_step"], j2["value"] * j2["ms_per_step"]
    assert abs(u2 - u1) < 1e-6 * u2                           # the same fixed job at either rank count
    if launcher == "self":
        r3 = subprocess.run([sys.executable, bench, "--gpus", "2", "--shard", "lambda"] + args, env=env, capture_output=True,
                            text=True, timeout=600)
        assert r3.returncode == 0, r3.stderr[-2000:]
        j3 = json.loads([ln for ln in r3.stdout.splitlines() if ln.startswith("{")][-1])
        assert j3["scaling"] == "weak" and "none" in j3["collective"]["op"]
        assert abs(j3["value"] * j3["ms_per_step"] - 2 * u1) < 1e-6 * u2      # weak: both ranks' updates
