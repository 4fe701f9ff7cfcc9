"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full
size -- through size-independent properties.

Bars (BASELINE.json north_star): neighbour indices (and everything integer: layers, permutations,
upwind ids) bit-exact; intensities within 1e-10 relative in fp64 (RTOL below).  The Delaunay
lines, dot products and path lengths are also bit-exact because fp64 +,*,/,sqrt are correctly
rounded on the device and the build disables FMA contraction on both sides."""
import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import _lib, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-10     # fp64 tolerance of the north star


from oracle.parity import rel as _rel     # element-wise: |a - b| < tol (|b| + smallest non-zero |b|) for EVERY element


def _valid_line_mask(so):
    m = np.zeros(so.delaunay_lines.shape[:2], dtype=bool)
    for j in range(so.D):
        m[:, j] = (j < so.neighbours[0]) & (so.neighbours[j + 1] > 0)
    return m


@pytest.fixture(scope="module")
def grids(bcc_small, voro_small):
    out = {}
    for name, (pos, nbr, bounds) in (("bcc", bcc_small), ("voronoi", voro_small)):
        out[name] = (vrt.VoronoiSites(pos, nbr, bounds, device=0), orc.make_sites(pos, nbr, bounds))
    return out


@pytest.fixture(params=["tiles", "steps", "levels", "patches"])
def path(request, monkeypatch, grids):
    """Every device path: the fused patch kernel (default), the layer-step kernels, the persistent
    layer-tile kernel and the one-launch-per-level kernels (general fallback).  The library reads
    VRT_PATH when a plan is created; plans the module's grids already hold follow set_option."""
    monkeypatch.setenv("VRT_PATH", request.param)
    for hs, _ in grids.values():
        hs.set_option("VRT_PATH", request.param)
    yield request.param
    for hs, _ in grids.values():
        hs.set_option("VRT_PATH", "auto")
        hs._options.pop("VRT_PATH")          # later plans take the environment's preset again


def test_native_library_is_the_compute_path():
    L = _lib.load()
    assert L.vrt_device_count() >= 1
    assert _lib.LIB_PATH.endswith("voronoirt_amd/libvrt_hip.so")


@pytest.mark.parametrize("name", ["bcc", "voronoi"])
def test_grid_and_delaunay_lines_bit_exact(grids, name):
    hs, so = grids[name]
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    m = _valid_line_mask(so)
    lines = hs.Delaunay_lines
    assert np.array_equal(lines[m], so.delaunay_lines[m])
    assert (lines[~m] == 0).all()


@pytest.mark.parametrize("name", ["bcc", "voronoi"])
@pytest.mark.parametrize("quad", ["ul7n12.dat", "ul9n20.dat", "n1.dat"])
def test_upwind_table_bit_exact(grids, name, quad):
    hs, so = grids[name]
    w, th, ph, nq = vrt.read_quadrature(quad)
    ks = vrt.quadrature_directions(th, ph)
    plan = vrt.FormalPlan(hs, ks, 3, dirs=[1 if t > 90 else -1 for t in th])
    for a in range(nq):
        assert np.array_equal(ks[a], orc.direction(th[a], ph[a]))
        up, dots, wt, r, st = orc.upwind_table(so, ks[a])
        gup, gd, gw, gr = plan.upwind(a)
        ok = st == 0
        assert np.array_equal(gup[ok], up[ok]), "upwind neighbour ids must be bit-exact"
        assert (gup[~ok] == 0).all()
        assert np.array_equal(gd[ok], dots[ok])
        assert np.array_equal(gr[ok], r[ok])
        assert np.allclose(gw[ok], wt[ok], rtol=4e-15, atol=1e-300)      # pow(): ulp-level
    plan.close()


def test_upwind_rows_longer_than_16_and_order_dependence():
    """Rows with more than 16 neighbours span several 16-lane chunks; permuting a row changes
    the reference's order-dependent choice and the kernel must follow it."""
    pos, nbr, bounds = synth.voronoi_grid(1200, seed=8, bounds=(0.0, 1.0, 0.0, 1.0, 0.0, 1.0))
    assert nbr[0].max() > 16
    rng = np.random.default_rng(0)
    for trial in range(3):
        nb = nbr.copy()
        for i in range(nb.shape[1]):
            c = nb[0, i]
            nb[1:c + 1, i] = rng.permutation(nb[1:c + 1, i])
        hs = vrt.VoronoiSites(pos, nb, bounds, device=0)
        so = orc.make_sites(pos, nb, bounds)
        k = orc.direction(112.824260481870382, 335.790538127899197)
        plan = vrt.FormalPlan(hs, [k], 3)
        up, dots, wt, r, st = orc.upwind_table(so, k)
        gup, gd, gw, gr = plan.upwind(0)
        ok = st == 0
        assert np.array_equal(gup[ok], up[ok]) and np.array_equal(gd[ok], dots[ok])
        plan.close()
        hs.close()


@pytest.mark.parametrize("name", ["bcc", "voronoi"])
def test_single_solves_match_oracle(grids, name, path):
    """Delaunay_upII / Delaunay_downII for the 12 directions of ul7n12, random S, α spanning all
    three linear_weights branches, random boundary intensity (also for the down rays)."""
    hs, so = grids[name]
    n = so.n
    rng = np.random.default_rng(11)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-3, 3, n) / (so.bounds[3] - so.bounds[2]) * 10
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for t, p in zip(th, ph):
        k = vrt.direction(t, p)
        if t > 90:
            I0 = rng.random(so.layers_up[1] - 1)
            ref = orc.Delaunay_upII(k, S, I0, alpha, so, 3)
            got = vrt.Delaunay_upII(k, S, I0, alpha, hs, 3)
        else:
            I0 = rng.random(so.layers_down[1] - 1)
            ref = orc.Delaunay_downII(k, S, I0, alpha, so, 3)
            got = vrt.Delaunay_downII(k, S, I0, alpha, hs, 3)
        assert _rel(got, ref) < RTOL
        assert got[(so.perm_up if t > 90 else so.perm_down)[-1] - 1] == 0.0   # never-visited site
    # second call with the same k hits the plan cache and must give the same bits
    k = vrt.direction(th[0], ph[0])
    I0 = np.zeros(so.layers_down[1] - 1)
    a = vrt.Delaunay_downII(k, S, I0, alpha, hs, 3)
    b = vrt.Delaunay_downII(k, S, I0, alpha, hs, 3)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n_sweeps", [1, 2, 4])
def test_other_sweep_counts(grids, n_sweeps, path):
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(5)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-2, 2, n) * 5
    k = vrt.direction(101.810709392034880, 235.428463450411130)
    I0 = rng.random(so.layers_up[1] - 1)
    assert _rel(vrt.Delaunay_upII(k, S, I0, alpha, hs, n_sweeps),
                orc.Delaunay_upII(k, S, I0, alpha, so, n_sweeps)) < RTOL


def test_up_solver_with_down_pointing_k(grids):
    """The reference's Delaunay_upII uses perm_up with whatever k it is handed; so does the
    drop-in (vrt_plan_create_ex with an explicit direction)."""
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(2)
    S, alpha = 1 + rng.random(n), np.full(n, 1e-6)
    k = vrt.direction(60.0, 20.0)                 # θ < 90 handed to the "up" solver
    I0 = rng.random(so.layers_up[1] - 1)
    assert _rel(vrt.Delaunay_upII(k, S, I0, alpha, hs, 3), orc.Delaunay_upII(k, S, I0, alpha, so, 3)) < RTOL


def test_golden_vectors(golden, path):
    g = golden
    exp = g["exp"]
    n = g["meta"]["n"]
    hs = vrt.read_cell(g["nbr_file"], n, g["pos"], g["bounds"], device=0)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), exp[key]), key
    w, th, ph, _ = vrt.read_quadrature(g["meta"]["quadrature"])
    S, al = exp["S"], exp["alpha"]
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    for a in g["meta"]["angles"]:
        gup, *_ = plan.upwind(a)
        ok = exp[f"status_{a}"] == 0
        assert np.array_equal(gup[ok], exp[f"up_{a}"][ok])
        k = vrt.direction(th[a], ph[a])
        if th[a] > 90:
            I = vrt.Delaunay_upII(k, S[:, 0].copy(), exp["I0_up"][:, 0].copy(), al[:, 0].copy(), hs, 3)
        else:
            I = vrt.Delaunay_downII(k, S[:, 0].copy(), exp["I0_down"][:, 0].copy(), al[:, 0].copy(), hs, 3)
        assert _rel(I, exp[f"I_{a}"]) < RTOL
    J, _ = plan.execute(S, al, weights=w, I0_up=exp["I0_up"], I0_down=exp["I0_down"])
    assert _rel(J, exp["J"]) < RTOL
    plan.close()


@pytest.mark.parametrize("nlam", [1, 3, 51, 64, 70])
def test_J_all_alpha_layouts(grids, nlam, path):
    """J_λ_voronoi with α per site, per (site, λ) and per (angle, site, λ); ragged λ counts."""
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(100 + nlam)
    S = 1 + rng.random((n, nlam))
    base = 10 ** rng.uniform(-3, 3, n) * 5
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    variants = [base, base[:, None] * (1 + rng.random((n, nlam)))]
    if nlam <= 51:
        variants.append(np.stack([base[:, None] * (1 + rng.random((n, nlam))) for _ in range(nq)]))
    for al in variants:
        if al.ndim == 1 and nlam > 1:
            ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
        else:
            ref = orc.J_voronoi(w, th, ph, S, al if al.ndim > 1 else al.reshape(n, 1), so,
                                I0_up=I0, nthreads=8, alpha_mode=None if al.ndim > 1 else 1)
        got = vrt.J_lambda_voronoi(S, al, hs, "ul7n12.dat", I0_up=I0)
        assert got.shape == (n, nlam) and _rel(got, ref) < RTOL


def test_per_angle_intensities_and_theta90_skipped(grids, path):
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(3)
    nlam = 4
    S = 1 + rng.random((n, nlam))
    al = 10 ** rng.uniform(-3, 3, (n, 1)) * (1 + rng.random((n, nlam))) / 6e6 * 10
    theta = np.array([150.0, 90.0, 35.0])
    phi = np.array([10.0, 0.0, 200.0])
    w = np.array([0.3, 0.4, 0.3])
    ks = vrt.quadrature_directions(theta, phi)
    plan = vrt.FormalPlan(hs, ks, 3, dirs=[1, 0, -1])
    I0u = rng.random((so.layers_up[1] - 1, nlam))
    I0d = rng.random((so.layers_down[1] - 1, nlam))
    J, I = plan.execute(S, al, weights=w, I0_up=I0u, I0_down=I0d, want_I=True)
    assert (I[1] == 0).all()                                  # θ = 90: no solve, no contribution
    for l in range(nlam):
        ref_u = orc.Delaunay_upII(ks[0], S[:, l], I0u[:, l], al[:, l], so, 3)
        ref_d = orc.Delaunay_downII(ks[2], S[:, l], I0d[:, l], al[:, l], so, 3)
        assert _rel(I[0][:, l], ref_u) < RTOL and _rel(I[2][:, l], ref_d) < RTOL
    ref = orc.J_voronoi(w, theta, phi, S, al, so, I0_up=I0u, I0_down=I0d)
    assert _rel(J, ref) < RTOL
    assert _rel(J, w[0] * I[0] + w[2] * I[2]) < 1e-15
    plan.close()


def test_searchlight_transparent_grid(grids):
    """The reference's searchlight test (src/compare_searchlight.jl:10-152): α = S = 0, a disk of
    radius 0.1 lit on the boundary layer, all 12 directions.  The beam must arrive unattenuated
    in total no larger than it started and match the oracle."""
    pos, nbr, bounds = synth.voronoi_grid(4000, seed=21)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    S = np.zeros(n)
    al = np.zeros(n)
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for t, p in zip(th, ph):
        up = t > 90
        perm, lay = (so.perm_up, so.layers_up) if up else (so.perm_down, so.layers_down)
        idx = perm[: lay[1] - 1] - 1
        lit = np.sqrt((pos[idx, 1] - 0.5) ** 2 + (pos[idx, 2] - 0.5) ** 2) < 0.1   # :77-82
        I0 = lit.astype(float)
        k = vrt.direction(t, p)
        got = (vrt.Delaunay_upII if up else vrt.Delaunay_downII)(k, S, I0, al, hs, 3)
        ref = (orc.Delaunay_upII if up else orc.Delaunay_downII)(k, S, I0, al, so, 3)
        assert _rel(got, ref) < RTOL
        assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-14
    hs.close()


def test_error_behaviour(grids):
    hs, so = grids["bcc"]
    n = so.n
    k = vrt.direction(150.0, 0.0)
    with pytest.raises(vrt.VrtError) as e:      # Julia: DimensionMismatch at I[perm[1:n1]] = I_0
        vrt.Delaunay_upII(k, np.zeros(n), np.zeros(3), np.zeros(n), hs, 3)
    assert e.value.code == _lib.VRT_EINVAL
    with pytest.raises(vrt.VrtError) as e:
        vrt.FormalPlan(hs, [[0.5, 0.5, 0.5]])               # not a unit vector
    assert e.value.code == _lib.VRT_EINVAL
    with pytest.raises(vrt.VrtError):
        vrt.FormalPlan(hs, [[-1.0, 0.0, 0.0]], n_sweeps=0)
    with pytest.raises(ValueError):
        vrt.Delaunay_upII(k, np.zeros(n - 1), np.zeros(so.layers_up[1] - 1), np.zeros(n), hs, 3)
    with pytest.raises(vrt.VrtError) as e:
        vrt.VoronoiSites(so.positions, so.neighbours, so.bounds, device=99)
    assert e.value.code == _lib.VRT_EINVAL


def test_site_without_upwind_is_reported():
    """A visited site all of whose neighbours are walls has no upwind neighbour; the reference
    would index with an uninitialised value, the library reports VRT_EGRID."""
    pos, nbr, bounds = synth.regular_lattice_grid(3, 3, 4)
    nbr = nbr.copy()
    n = nbr.shape[1]
    victim = int(np.argmax((pos[:, 0] > 0.3) & (pos[:, 0] < 0.5)))   # second plane from the bottom
    c = nbr[0, victim]
    ids = nbr[1:c + 1, victim]
    # keep one neighbour so the layering still reaches it, but place it straight ABOVE (dot = -1)
    above = [v for v in ids if v > 0 and pos[v - 1, 0] > pos[victim, 0]]
    if not above:
        pytest.skip("victim is in the top layer")
    nbr[1:, victim] = 0
    nbr[1, victim] = above[0]
    nbr[0, victim] = 1
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    k = np.array([-1.0, 0.0, 0.0])
    rc, dots, idx = orc.smallest_angle(victim, so, k)
    assert rc == -1                                      # d = -1 is not > -1: nothing selected
    assert so.perm_up[-1] - 1 != victim                  # it IS visited by the sweep
    with pytest.raises(vrt.VrtError) as e:
        vrt.FormalPlan(hs, [k], 3)
    assert e.value.code == _lib.VRT_EGRID
    hs.close()


def test_execute_dev_with_torch_tensors_and_padding(grids, path):
    """Device-pointer entry point on a torch stream, with a padded leading dimension."""
    import torch
    hs, so = grids["voronoi"]
    n = so.n
    nlam, ld = 5, 8
    rng = np.random.default_rng(9)
    S = 1 + rng.random((n, nlam))
    al = 10 ** rng.uniform(-3, 3, (n, 1)) * (1 + rng.random((n, nlam))) * 5
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    dev = torch.device("cuda", 0)
    Sd = torch.full((n, ld), float("nan"), dtype=torch.float64, device=dev)
    Ad = torch.full((n, ld), float("nan"), dtype=torch.float64, device=dev)
    Sd[:, :nlam] = torch.from_numpy(S).to(dev)
    Ad[:, :nlam] = torch.from_numpy(al).to(dev)
    I0d = torch.from_numpy(I0).to(dev).contiguous()
    Jd = torch.full((n, ld), -7.0, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        plan.execute_dev(nlam, ld, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w,
                         dJ=Jd.data_ptr(), dI0_up=I0d.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    ms, launches = plan.last_sweep_timing()
    # tiles: one persistent launch, preceded by the chip-wide coefficient launch when layers are <= 4096 sites
    assert ms > 0 and (launches == plan.num_levels if path == "levels" else (launches in (1, 2) if path == "tiles" else launches > 0))
    J = Jd.cpu().numpy()
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=4)
    assert _rel(J[:, :nlam], ref) < RTOL
    assert (J[:, nlam:] == -7.0).all()                    # padding columns untouched
    plan.close()


# ---- BASELINE.json full size: properties that need no oracle run ---------------------------------
@pytest.fixture(scope="module")
def full_grid():
    a, c = synth.BCC_CONFIGS["C4"]
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=2022)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    return hs, pos, bounds


def test_full_size_properties(full_grid):
    """~1M sites (995 566), ul7n12: (1) linearity of the formal solution in (S, I_0) at fixed α,
    (2) a constant source function with I_0 = S reproduces itself wherever the sweep converged
    and never overshoots, (3) J of a unit field is bounded by Σw = 1, (4) determinism."""
    import torch
    hs, pos, bounds = full_grid
    n = hs.n
    assert n == 995566
    nlam = 2
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    assert plan.num_nodes <= 3 * n * nq
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    z = torch.as_tensor(pos[:, 0], device=dev)
    al = (1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6))[:, None] * \
        (1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64))
    n1 = int(hs.layers_up[1] - 1)
    stream = torch.cuda.current_stream().cuda_stream

    def solve(S, I0):
        J = torch.empty((n, nlam), dtype=torch.float64, device=dev)
        plan.execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w,
                         dJ=J.data_ptr(), dI0_up=I0.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        return J

    S1 = 1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
    S2 = 1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
    A1 = torch.rand((n1, nlam), generator=g, device=dev, dtype=torch.float64)
    A2 = torch.rand((n1, nlam), generator=g, device=dev, dtype=torch.float64)
    J1, J2 = solve(S1, A1), solve(S2, A2)
    J12 = solve(S1 + 2.0 * S2, A1 + 2.0 * A2)
    lin = (J12 - (J1 + 2.0 * J2)).abs().max().item() / J12.abs().max().item()
    assert lin < 1e-12
    assert torch.equal(solve(S1, A1), J1)                          # bitwise reproducible
    ones = torch.ones((n, nlam), dtype=torch.float64, device=dev)
    Jc = solve(ones, torch.ones((n1, nlam), dtype=torch.float64, device=dev))
    assert Jc.max().item() <= 1.0 + 1e-12 and Jc.min().item() >= 0.0
    plan.close()


def test_full_size_sample_against_oracle(full_grid):
    """One up and one down direction at full size against the oracle (a few seconds of CPU)."""
    hs, pos, bounds = full_grid
    rng = np.random.default_rng(4)
    n = hs.n
    nbr = hs.neighbours
    so = orc.make_sites(pos, nbr, bounds)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    S = 1 + rng.random(n)
    al = 1e-2 * np.exp(-(pos[:, 0] - bounds[0]) / 0.7e6) * (1 + rng.random(n))
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for a in (1, 0):
        k = vrt.direction(th[a], ph[a])
        if th[a] > 90:
            I0 = rng.random(so.layers_up[1] - 1)
            ref = orc.Delaunay_upII(k, S, I0, al, so, 3)
            got = vrt.Delaunay_upII(k, S, I0, al, hs, 3)
        else:
            I0 = np.zeros(so.layers_down[1] - 1)
            ref = orc.Delaunay_downII(k, S, I0, al, so, 3)
            got = vrt.Delaunay_downII(k, S, I0, al, hs, 3)
        assert _rel(got, ref) < RTOL


def _large_layer_case(a, nlam, seed):
    pos, nbr, bounds = synth.bcc_grid(a, 3, seed=seed)          # 2 a^2 sites per layer
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    rng = np.random.default_rng(8)
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    return hs, so, S, al, I0


def test_layers_of_8712_sites_run_on_the_single_wavelength_step_kernels(monkeypatch):
    """Layers above 8192 sites exceed the wavelength-PAIR level kernel of the layer-step path (16-byte tile
    slots, 15 registers of coefficients per site); up to 12 288 sites they run on its single-wavelength
    one (fp64, tile in sorted order).  The default path is the patch kernel, which has no such limit.
    Same results on every path that holds the grid, to the 1e-10 bar."""
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so, S, al, I0 = _large_layer_case(66, 12, 5)           # 8712 sites per layer
    assert int(np.diff(so.layers_up).max()) == 8712
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == "patches"
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
    assert _rel(J, ref) < RTOL
    plan.set_option("VRT_PATH", "steps")
    J1, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == "steps" and _rel(J1, ref) < RTOL
    plan.set_option("VRT_PATH", "tiles")                      # the persistent tile kernel cannot hold them
    with pytest.raises(vrt.VrtError):
        plan.execute(S, al, weights=w, I0_up=I0)
    plan.set_option("VRT_PATH", "levels")
    J2, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert _rel(J2, ref) < RTOL
    plan.close()
    hs.close()


def test_layers_of_17298_sites_fp32_on_steps_fp64_not(monkeypatch):
    """17 298-site layers (C5's are 17 672): with fp32 storage the layer-step path's single-wavelength kernel
    holds them (float tile and coefficients, 18 sites per thread); with fp64 it does not (12 288) and
    VRT_PATH = steps is refused.  The default (patch kernel) takes both."""
    import torch
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so, S, al, I0 = _large_layer_case(93, 6, 6)            # 2 * 93^2 = 17 298 sites per layer
    assert int(np.diff(so.layers_up).max()) == 17298
    n, nlam = so.n, S.shape[1]
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    ref = orc.J_voronoi(w, th, ph, S.astype(np.float32).astype(np.float64), al.astype(np.float32).astype(np.float64),
                        so, I0_up=I0.astype(np.float32).astype(np.float64), nthreads=8)
    dev = torch.device("cuda", 0)
    Sd, Ad, I0d = (torch.from_numpy(x.astype(np.float32)).to(dev).contiguous() for x in (S, al, I0))
    for want in ("patches", "steps"):
        plan.set_option("VRT_PATH", "auto" if want == "patches" else want)
        Jd = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
        plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                         dI0_up=I0d.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, f32=True)
        torch.cuda.synchronize()
        assert plan.last_path == want
        assert _rel(Jd.cpu().numpy().astype(np.float64), ref) < 5e-6
    plan.set_option("VRT_PATH", "auto")
    J, _ = plan.execute(S, al, weights=w, I0_up=I0)           # fp64
    assert plan.last_path == "patches"
    assert _rel(J, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)) < RTOL
    plan.set_option("VRT_PATH", "steps")                      # fp64: too large for the step kernels
    with pytest.raises(vrt.VrtError):
        plan.execute(S, al, weights=w, I0_up=I0)
    plan.close()
    hs.close()


@pytest.mark.parametrize("a, nl", [(100, 4), (123, 3)])
def test_layers_of_20000_and_30000_sites_run_on_the_patch_kernel(monkeypatch, a, nl):
    """Beyond 18 432 sites per layer nothing holds a layer in one workgroup (the reference's
    density-stratified grids have ~30 000-site layers at 1 M sites, src/sample_grids.jl:223-230):
    the fused patch kernel cuts every layer into patches and solves each, with the halo of its
    in-layer dependency cone, in a workgroup of its own.  fp64 and fp32 storage, against the oracle
    and against the general level path; per-angle alpha in the native layout is accepted."""
    import torch
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so, S, al, I0 = _large_layer_case(a, nl, 7)
    assert int(np.diff(so.layers_up).max()) == 2 * a * a
    n, nlam = so.n, S.shape[1]
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == "patches"
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
    assert _rel(J, ref) < RTOL
    plan.set_option("VRT_PATH", "levels")
    J2, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == "levels" and _rel(J2, J) < 5e-11
    plan.set_option("VRT_PATH", "steps")                      # no layer-step kernel holds such a layer
    with pytest.raises(vrt.VrtError):
        plan.execute(S, al, weights=w, I0_up=I0)
    plan.set_option("VRT_PATH", "auto")
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    Sd, Ad, I0d = (torch.from_numpy(x.astype(np.float32)).to(dev).contiguous() for x in (S, al, I0))
    Jd = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                     dI0_up=I0d.data_ptr(), stream=st, f32=True)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    assert _rel(Jd.cpu().numpy().astype(np.float64), ref) < 5e-6
    # per-angle alpha, caller layout -> native layout -> the same J bit for bit
    al3 = np.stack([al * (1 + 0.05 * i) for i in range(nq)])
    S64, A64, I064 = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (S, al3, I0))
    Ja, Jn = torch.zeros((n, nlam), dtype=torch.float64, device=dev), torch.zeros((n, nlam), dtype=torch.float64, device=dev)
    plan.execute_dev(nlam, nlam, S64.data_ptr(), A64.data_ptr(), _lib.ALPHA_ANGLE_SITE_LAM, w, dJ=Ja.data_ptr(),
                     dI0_up=I064.data_ptr(), stream=st)
    native = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
    plan.alpha_to_native_dev(nlam, nlam, A64.data_ptr(), native.data_ptr(), stream=st)
    plan.execute_dev(nlam, nlam, S64.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w, dJ=Jn.data_ptr(),
                     dI0_up=I064.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert plan.last_path == "patches" and torch.equal(Ja, Jn)
    refa = orc.J_voronoi(w, th, ph, S[:, :2].copy(), al3[:, :, :2].copy(), so, I0_up=I0[:, :2].copy(), nthreads=8)
    assert _rel(Ja[:, :2].cpu().numpy(), refa) < RTOL
    plan.close()
    hs.close()


def test_wall_layer_larger_than_every_other_layer(monkeypatch):
    """A stratified grid whose boundary layer is its LARGEST layer (1 047 sites against 1 044): the persistent
    tile kernel keeps that layer in LDS as the first "previous" layer, so its tile stride must count layer 1
    too.  Up-only single-angle plans on every path against the oracle."""
    pos, nbr, bounds = synth.voronoi_grid(3000, seed=5, bounds=(0.0, 0.5, 0.0, 1.0, 0.0, 1.0), scale_height=0.06)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    sizes = np.diff(so.layers_up)
    assert sizes[0] == sizes.max() and sizes[0] > sizes[1:].max()
    rng = np.random.default_rng(2)
    n = so.n
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-2, 2, n)
    I0 = rng.random(so.layers_up[1] - 1)
    for t, p in ((109.7, 193.6), (152.7, 315.5)):
        k = vrt.direction(t, p)
        ref = orc.Delaunay_upII(k, S, I0, alpha, so, 3)
        for path_ in ("tiles", "patches", "steps", "levels"):
            hs.set_option("VRT_PATH", path_)
            plan = vrt.FormalPlan(hs, [k], 3, dirs=[1])
            for pre in (1, 0):
                plan.set_option("VRT_TILE_PRE", pre)
                J, _ = plan.execute(S, alpha, weights=[1.0], I0_up=I0)
                assert plan.last_path == path_ and _rel(J[:, 0], ref) < RTOL, (path_, pre)
            plan.close()
    hs.close()


def test_default_path_choice(grids, monkeypatch):
    """The fused patch kernel is the default; a lone (angle, wavelength) problem on small layers takes the
    two launches of the persistent tile kernel.  Options: unknown names and creation-only ones are refused."""
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(1)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    S = 1 + rng.random((n, 24))
    al = 5 * 10 ** rng.uniform(-2, 2, (n, 24))
    plan.execute(S, al, weights=w)
    assert plan.last_path == "patches"
    plan.execute(S[:, :2].copy(), al[:, :2].copy(), weights=w)
    assert plan.last_path == "patches"
    with pytest.raises(vrt.VrtError):
        plan.set_option("VRT_NO_SUCH_OPTION", 1)
    with pytest.raises(vrt.VrtError):
        plan.set_option("VRT_PATH", "sideways")
    with pytest.raises(vrt.VrtError):
        plan.set_option("VRT_PATCH_NT", 256)                  # shapes what plan creation built
    plan.close()
    one = vrt.FormalPlan(hs, [vrt.direction(th[0], ph[0])], 3)
    one.execute(S[:, :1].copy(), al[:, :1].copy(), weights=[1.0])
    assert one.last_path == "tiles"
    one.close()


@pytest.mark.parametrize("f32_path", ["patches", "steps", "levels"])
def test_fp32_value_path_against_fp64_oracle(grids, f32_path, monkeypatch):
    """BASELINE config C5's path: S, α, I_0, J stored as float32, arithmetic in fp64.  Checked
    against the fp64 oracle fed with the same float32-rounded inputs; the remaining difference is
    the float32 rounding of the stored intensities (tolerance 5e-6 relative, fp64 path: 1e-10)."""
    import torch
    if f32_path == "patches":
        monkeypatch.delenv("VRT_PATH", raising=False)          # the default
    else:
        monkeypatch.setenv("VRT_PATH", f32_path)               # read when the plan below is created
    hs, so = grids["voronoi"]
    n = so.n
    nlam = 7
    rng = np.random.default_rng(12)
    S = (1 + rng.random((n, nlam))).astype(np.float32)
    al = (5 * 10 ** rng.uniform(-3, 3, (n, 1)) * (1 + rng.random((n, nlam)))).astype(np.float32)
    I0 = rng.random((so.layers_up[1] - 1, nlam)).astype(np.float32)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    dev = torch.device("cuda", 0)
    Sd, Ad, I0d = (torch.from_numpy(a).to(dev).contiguous() for a in (S, al, I0))
    Jd = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
    Id = torch.zeros((nq, n, nlam), dtype=torch.float32, device=dev)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                     dI0_up=I0d.data_ptr(), dI_out=Id.data_ptr(),
                     stream=torch.cuda.current_stream().cuda_stream, f32=True)
    torch.cuda.synchronize()
    assert plan.last_path == f32_path
    ref = orc.J_voronoi(w, th, ph, S.astype(np.float64), al.astype(np.float64), so,
                        I0_up=I0.astype(np.float64), nthreads=4)
    J = Jd.cpu().numpy().astype(np.float64)
    assert _rel(J, ref) < 5e-6
    assert _rel(J, ref) > 1e-12          # it really is the float32 path
    k = vrt.direction(th[1], ph[1])
    Iref = orc.Delaunay_upII(k, S[:, 0].astype(np.float64), I0[:, 0].astype(np.float64),
                             al[:, 0].astype(np.float64), so, 3)
    assert _rel(Id[1, :, 0].cpu().numpy().astype(np.float64), Iref) < 5e-6
    plan.close()


def _c_caller_lattice():
    nx, ny, nz = 4, 5, 6
    n = nx * ny * nz
    pos = np.zeros((n, 3))
    nbr = np.zeros((7, n), dtype=np.int64)
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                s = (i * ny + j) * nz + k
                pos[s] = ((k + 0.5) / nz, (i + 0.5) / nx, (j + 0.5) / ny)
                nbr[0, s] = 6
                nbr[1:, s] = [((i + 1) % nx * ny + j) * nz + k + 1, ((i - 1) % nx * ny + j) * nz + k + 1,
                              (i * ny + (j + 1) % ny) * nz + k + 1, (i * ny + (j - 1) % ny) * nz + k + 1,
                              s + 2 if k + 1 < nz else -6, s if k > 0 else -5]
    return n, orc.make_sites(pos, nbr, (0, 1, 0, 1, 0, 1))


def _build_c_caller(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "voronoirt_amd")
    exe = tmp_path / "c_caller"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_caller.c"), "-o", str(exe),
                           "-L", libdir, "-lvrt_hip", f"-Wl,-rpath,{libdir}", "-lm"])
    return exe


def test_plain_c_caller_matches_oracle(tmp_path):
    """examples/c_caller.c -- a C host standing in for the Julia caller -- gives the oracle's answer."""
    import subprocess
    exe = _build_c_caller(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = np.array([float(ln.split()[1]) for ln in out.stdout.strip().splitlines()])
    n, so = _c_caller_lattice()
    ref = orc.Delaunay_upII(orc.direction(150.0, 30.0), np.ones(n), np.full(so.layers_up[1] - 1, 3.0),
                            np.full(n, 2.0), so, 3)
    assert _rel(got, ref) < RTOL


def test_plain_c_caller_plays_J_lambda_voronoi(tmp_path):
    """Scenario 2 of examples/c_caller.c is the caller julia/VoronoiRT_hip.jl's J_line is (the body
    of J_λ_voronoi, lambda_iteration.jl:60-113, with the formal solves batched): 12 angles x 5
    wavelengths, per-angle α_tot (nλ, n, n_angles), I_0 for the up rays, one vrt_plan_execute."""
    import subprocess
    exe = _build_c_caller(tmp_path)
    out = subprocess.run([str(exe), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    n, so = _c_caller_lattice()
    nlam, na = 5, 12
    got = np.zeros((n, nlam))
    for ln in out.stdout.strip().splitlines():
        i, l, v = ln.split()
        got[int(i) - 1, int(l) - 1] = float(v)
    ii, ll = np.meshgrid(np.arange(n), np.arange(nlam), indexing="ij")
    S = 1.0 + 0.1 * ((ii * 7 + ll * 3) % 11)
    alpha = np.stack([0.5 + 0.05 * ((ii + 2 * ll + 3 * a) % 13) for a in range(na)])
    n1 = int(so.layers_up[1] - 1)
    pp, l1 = np.meshgrid(np.arange(n1), np.arange(nlam), indexing="ij")
    I0 = 2.0 + 0.25 * ((pp + l1) % 5)
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    ref = orc.J_voronoi(w, th, ph, S, alpha, so, I0_up=I0, nthreads=2)
    assert _rel(got, ref) < RTOL


@pytest.mark.parametrize("handles", [0, 2])
def test_plain_c_caller_plays_Lambda_voronoi(tmp_path, handles):
    """Scenario 3 of examples/c_caller.c is the caller julia/VoronoiRT_hip.jl's Λ_voronoi is: the reference's loop
    (src/lambda_iteration.jl:205-300) over vrt_lambda_create / _iterate / _get with host arrays -- library-owned
    device state, one call per iteration, γ following the populations.  Scenario 4 (handles = 2) is the same loop
    across the devices of a node, vrt_multi_create + vrt_multi_lambda_* (here: two handles on device 0).  Against the
    same loop driven by the oracle."""
    import subprocess
    from test_physics import _lambda_case, _oracle_lambda_iteration
    exe = _build_c_caller(tmp_path)
    n, so = _c_caller_lattice()
    case = _lambda_case(so.positions, (0.0, 1.0, 0.0, 1.0, 0.0, 1.0), 3)
    maxiter = 3
    nlam = case.lam.size
    blob = tmp_path / "lambda_inputs.bin"
    with open(blob, "wb") as f:
        np.array([n, nlam, maxiter, handles], dtype=np.int64).tofile(f)
        np.asarray(case.blocks, dtype=np.int64).tofile(f)
        np.array([case.lambda0, case.c0, case.strength_const, case.Bij, case.Bji, case.sigma_bb_const, case.hc_over_kB,
                  case.pref_ij, case.pref_ji]).tofile(f)
        for a in (case.lam, case.velocity, case.doppler, case.gamma_static, case.gamma_unsold, case.alpha_cont, case.eps,
                  case.temperature, case.atom_density, case.B0, case.lte, case.C, case.planck2, case.sigma_bf1,
                  case.sigma_bf2):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    out = subprocess.run([str(exe), "4" if handles else "3", str(blob)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    hist, J, S, P = [], np.zeros((n, nlam)), np.zeros((n, nlam)), np.zeros((3, n))
    for ln in out.stdout.strip().splitlines():
        t = ln.split()
        if t[0] == "hist":
            hist.append(float(t[2]))
        elif t[0] == "JS":
            J[int(t[1]) - 1, int(t[2]) - 1], S[int(t[1]) - 1, int(t[2]) - 1] = float(t[3]), float(t[4])
        else:
            P[:, int(t[1]) - 1] = [float(x) for x in t[2:5]]
    J_ref, S_ref, pops_ref, hist_ref = _oracle_lambda_iteration(case, so, "ul7n12.dat", maxiter)
    assert len(hist) == maxiter and np.allclose(hist, hist_ref, rtol=1e-8)
    assert np.abs(J - J_ref).max() < 1e-9 * np.abs(J_ref).max() and np.abs(S / S_ref - 1).max() < 1e-9
    assert np.abs(P / pops_ref - 1).max() < 1e-9


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_multi_device_object_of_the_c_abi(voro_small, devices):
    """vrt_multi_*: one process, several device handles.  On a one-GPU box the same device is
    listed two or three times: the wavelength-block sharding must reproduce the one-handle J bit for bit (every
    wavelength is solved by exactly one handle), the angle sharding to summation order (its all-reduce is
    replaced by kernel adds here; with distinct devices it is ONE RCCL all-reduce)."""
    pos, nbr, bounds = voro_small
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    dirs = [1 if t > 90 else -1 for t in th]
    mp = vrt.MultiDevicePlan(pos, nbr, bounds, vrt.quadrature_directions(th, ph), dirs=dirs, devices=devices)
    assert not mp.uses_rccl or len(set(devices)) == len(devices) > 1
    rng = np.random.default_rng(len(devices))
    for nlam, per_angle in ((5, False), (7, True), (1, False)):
        S = 1 + rng.random((n, nlam))
        al = 5 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
        if per_angle:
            al = np.stack([al * (1 + 0.1 * rng.random((n, nlam))) for _ in range(nq)])
        I0u, I0d = rng.random((so.layers_up[1] - 1, nlam)), rng.random((so.layers_down[1] - 1, nlam))
        ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0u, I0_down=I0d, nthreads=4)
        mp.set_shard("auto")
        J = mp.execute(S, al, w, I0_up=I0u, I0_down=I0d)
        assert mp.last_shard == ("lambda" if nlam >= len(devices) else "angle")
        assert _rel(J, ref) < RTOL
        mp.set_shard("lambda")
        Jl = mp.execute(S, al, w, I0_up=I0u, I0_down=I0d)
        mp.set_shard("angle")
        Ja = mp.execute(S, al, w, I0_up=I0u, I0_down=I0d)
        assert mp.last_shard == "angle" and _rel(Ja, ref) < RTOL and _rel(Ja, Jl) < 1e-13
        if len(devices) == 1:
            hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
            plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=dirs)
            J1, _ = plan.execute(S, al, weights=w, I0_up=I0u, I0_down=I0d)
            assert np.array_equal(J1, Jl)
            plan.close()
            hs.close()
    # the wavelength count changes from call to call (the chained launch's item sets are rebuilt every time) while the
    # handles' launches share the chip: every J still right (regression: progress words zeroed on the null stream)
    cases = {}
    for nlam in (1, 2, 3):
        S = 1 + rng.random((n, nlam))
        al = 5 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
        I0u = rng.random((so.layers_up[1] - 1, nlam))
        cases[nlam] = (S, al, I0u, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0u, nthreads=4))
    for rep in range(24):
        S, al, I0u, ref = cases[(1, 3, 2, 3, 1, 2, 2)[rep % 7]]
        mp.set_shard("angle" if rep % 3 else "lambda")
        assert _rel(mp.execute(S, al, w, I0_up=I0u), ref) < RTOL, rep
    mp.close()
    with pytest.raises(vrt.VrtError):
        vrt.MultiDevicePlan(pos, nbr, bounds, vrt.quadrature_directions(th, ph), dirs=dirs, devices=(0, 97))


def test_degenerate_grids_and_plans(path):
    """Edge cases on every device path: a single-layer grid (every cell touches the wall: no sweep
    at all, I = I_0 except the never-visited last site), a two-layer grid, one angle / one
    wavelength, and a plan whose only direction is horizontal (θ = 90: skipped, J = 0)."""
    rng = np.random.default_rng(3)
    for nz in (1, 2, 3):
        pos, nbr, bounds = synth.regular_lattice_grid(4, 3, nz, seed=nz, jitter=0.2)
        hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
        so = orc.make_sites(pos, nbr, bounds)
        n = so.n
        assert np.array_equal(hs.layers_up, so.layers_up) and np.array_equal(hs.perm_down, so.perm_down)
        S, al = 1 + rng.random(n), 3 * rng.random(n)
        for theta, phi in ((150.0, 20.0), (40.0, 250.0)):
            k = vrt.direction(theta, phi)
            up = theta > 90
            lay = so.layers_up if up else so.layers_down
            I0 = 1 + rng.random(lay[1] - 1)
            got = (vrt.Delaunay_upII if up else vrt.Delaunay_downII)(k, S, I0, al, hs, 3)
            ref = (orc.Delaunay_upII if up else orc.Delaunay_downII)(k, S, I0, al, so, 3)
            assert _rel(got, ref) < RTOL
            if nz == 1:
                perm = so.perm_up if up else so.perm_down
                assert np.array_equal(got[perm[:-1] - 1], I0) and got[perm[-1] - 1] == 0.0
        # θ = 90 only: nothing to solve, J = 0 (lambda_iteration.jl:98,104 skip it)
        plan = vrt.FormalPlan(hs, [vrt.direction(90.0, 10.0)], 3, dirs=[0])
        J, I = plan.execute(np.ones((n, 2)), np.ones((n, 2)), weights=[1.0], want_I=True)
        assert (J == 0).all() and (I == 0).all()
        plan.close()
        hs.close()


def test_device_resident_lambda_iteration(grids):
    """A whole Λ-iteration without host round trips (the loop of Λ_voronoi,
    src/lambda_iteration.jl:253-283, with the physics reduced to given ε and B): J on the device,
    S_new = (1-ε)J + εB and the convergence measure on the device, repeated; against the same loop
    driven by the CPU oracle."""
    import torch
    from voronoirt_amd.api import lambda_update_dev
    hs, so = grids["voronoi"]
    n = so.n
    nlam = 24                                              # 288 problems: layer-step kernels
    rng = np.random.default_rng(21)
    B = 1 + rng.random((n, nlam))
    eps = 10 ** rng.uniform(-3, 0, n)                      # destruction probability per site
    al = 5 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    n1 = so.layers_up[1] - 1
    bottom = so.perm_up[:n1] - 1
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    Bd, epsd, ald = (torch.from_numpy(a).to(dev).contiguous() for a in (B, eps, al))
    S_old = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
    S_new = Bd.clone()                                     # S_new = B initially (LTE start)
    Jd = torch.zeros_like(S_new)
    bottom_d = torch.from_numpy(bottom).to(dev)
    S_ref_new, S_ref_old = B.copy(), np.zeros_like(B)
    hist_gpu, hist_ref = [], []
    for it in range(4):
        S_old.copy_(S_new)
        I0 = S_old[bottom_d].contiguous()
        plan.execute_dev(nlam, nlam, S_old.data_ptr(), ald.data_ptr(), _lib.ALPHA_SITE_LAM, w,
                         dJ=Jd.data_ptr(), dI0_up=I0.data_ptr(), stream=stream)
        hist_gpu.append(lambda_update_dev(hs, nlam, nlam, Jd.data_ptr(), Bd.data_ptr(), epsd.data_ptr(),
                                          S_old.data_ptr(), S_new.data_ptr(), stream))
        S_ref_old = S_ref_new.copy()
        J_ref = orc.J_voronoi(w, th, ph, S_ref_old, al, so, I0_up=S_ref_old[bottom], nthreads=8)
        S_ref_new = (1 - eps)[:, None] * J_ref + eps[:, None] * B
        hist_ref.append(float(np.abs(1 - S_ref_old / S_ref_new).max()))
    assert plan.last_path == "patches"
    assert _rel(S_new.cpu().numpy(), S_ref_new) < RTOL
    assert np.allclose(hist_gpu, hist_ref, rtol=1e-9)
    assert hist_gpu[-1] < hist_gpu[0]                      # the iteration contracts
    # NaN propagates like Julia's maximum
    S_new[0, 0] = float("nan")
    S_old.copy_(S_new)
    d = lambda_update_dev(hs, nlam, nlam, Jd.data_ptr(), Bd.data_ptr(), epsd.data_ptr(), S_old.data_ptr(),
                          S_new.data_ptr(), stream)
    assert np.isnan(d)
    plan.close()


def test_lambda_update_large_with_leading_dimension(grids):
    """vrt_lambda_update_dev beyond one pass of its grid-stride loop (> 1 M elements) and with a
    padded leading dimension: S_new and the convergence scalar against numpy, padding untouched."""
    import torch
    from voronoirt_amd.api import lambda_update_dev
    hs, so = grids["bcc"]
    n = so.n
    nlam = max(8, 3_000_000 // n + 1)
    ld = nlam + 3
    rng = np.random.default_rng(5)
    J, B, S_old = (1 + rng.random((n, ld)) for _ in range(3))
    eps = 10 ** rng.uniform(-3, 0, n)
    dev = torch.device("cuda", 0)
    Jd, Bd, Sd, ed = (torch.from_numpy(a).to(dev) for a in (J, B, S_old, eps))
    out = torch.full((n, ld), -7.0, dtype=torch.float64, device=dev)
    d = lambda_update_dev(hs, nlam, ld, Jd.data_ptr(), Bd.data_ptr(), ed.data_ptr(), Sd.data_ptr(), out.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    ref = (1 - eps)[:, None] * J[:, :nlam] + eps[:, None] * B[:, :nlam]
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :nlam], ref)
    assert (got[:, nlam:] == -7.0).all()
    assert d == np.abs(1 - S_old[:, :nlam] / ref).max()


def test_concurrent_single_solves_on_one_handle(grids):
    """The reference calls Delaunay_*II concurrently from Threads.@threads with a shared `sites`
    (lambda_iteration.jl:91-107); the drop-in must be re-entrant on one grid handle."""
    import threading
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(77)
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    jobs = []
    for j in range(12):
        S = 1 + rng.random(n)
        al = 1e-6 * 10 ** rng.uniform(-2, 2, n)
        up = th[j] > 90
        lay = so.layers_up if up else so.layers_down
        I0 = rng.random(lay[1] - 1)
        jobs.append((vrt.direction(th[j], ph[j]), S, I0, al, up))
    out = [None] * len(jobs)

    def work(j):
        k, S, I0, al, up = jobs[j]
        for _ in range(3):
            out[j] = (vrt.Delaunay_upII if up else vrt.Delaunay_downII)(k, S, I0, al, hs, 3)

    threads = [threading.Thread(target=work, args=(j,)) for j in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j, (k, S, I0, al, up) in enumerate(jobs):
        ref = (orc.Delaunay_upII if up else orc.Delaunay_downII)(k, S, I0, al, so, 3)
        assert _rel(out[j], ref) < RTOL


def test_level_path_with_hipgraph_replay(grids, monkeypatch):
    """VRT_GRAPH=1: the level-launch sequence is captured once and replayed; same results, also
    when the buffers (hence the captured arguments) change between calls."""
    monkeypatch.setenv("VRT_PATH", "levels")
    monkeypatch.setenv("VRT_GRAPH", "1")
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(31)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    for nlam in (3, 3, 5):
        S = 1 + rng.random((n, nlam))
        al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
        I0 = rng.random((so.layers_up[1] - 1, nlam))
        J, _ = plan.execute(S, al, weights=w, I0_up=I0)
        assert plan.last_path == "levels"
        assert _rel(J, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=4)) < RTOL
    plan.close()


@pytest.mark.parametrize("a,c,K", [(26, 4, 2), (36, 3, 4), (40, 4, 4), (52, 3, 8)])
def test_every_sites_per_thread_variant(a, c, K, path):
    """The LDS-tile kernels are instantiated for 2, 4 and 8 sites per thread (layers up to 2048 /
    4096 / 8192 sites; the persistent tile kernel also as 768 threads x 2 or 4 sites for layers up
    to 1536 / 3072 sites: a = 26 and a = 36); each variant against the oracle on every path."""
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=a)
    per_layer = 2 * a * a
    assert (K // 2) * 1024 < per_layer <= K * 1024
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    n = so.n
    rng = np.random.default_rng(K)
    nlam = 3
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    w, th, ph, nq = vrt.read_quadrature("ul2n3.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == path
    assert _rel(J, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=4)) < RTOL
    plan.close()
    hs.close()


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_level_kernel_instantiation_of_the_step_path(grids, K, monkeypatch):
    """k_step_levels is instantiated for 1..8 sites per thread (14 VGPRs of register-resident
    coefficients per site and wavelength pair; K = 8 spills).  VRT_STEP_K forces a larger K than
    the grid needs, so every instantiation runs on the small fixture; odd nlam pads a pair."""
    monkeypatch.setenv("VRT_PATH", "steps")
    monkeypatch.setenv("VRT_STEP_K", str(K))
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(100 + K)
    nlam = 5
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    w, th, ph, nq = vrt.read_quadrature("ul2n3.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J, _ = plan.execute(S, al, weights=w, I0_up=I0)
    assert plan.last_path == "steps"
    assert _rel(J, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=4)) < RTOL
    plan.close()


@pytest.mark.parametrize("K", [2, 4, 6, 8, 10, 12, 14, 16, 18])
@pytest.mark.parametrize("f32", [False, True])
def test_every_single_wavelength_level_kernel_instantiation(grids, K, f32, monkeypatch):
    """k_step_levels1 (one wavelength per workgroup, tile in sorted order) is instantiated for even
    sites-per-thread counts up to 12 (fp64 storage) / 18 (fp32 storage).  VRT_STEP_SINGLE=1 selects it
    on a small grid, VRT_STEP_K forces the instantiation; fp64 results equal the pair kernel's bit
    for bit (same arithmetic per wavelength)."""
    import torch
    if K > 12 and not f32:
        pytest.skip("fp64 storage: at most 12 sites per thread")
    monkeypatch.setenv("VRT_PATH", "steps")
    monkeypatch.setenv("VRT_STEP_K", str(K))
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(200 + K)
    nlam = 5
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    I0 = rng.random((so.layers_up[1] - 1, nlam))
    w, th, ph, nq = vrt.read_quadrature("ul2n3.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    if not f32:
        J_pair, _ = plan.execute(S, al, weights=w, I0_up=I0)
        plan.set_option("VRT_STEP_SINGLE", 1)
        J, _ = plan.execute(S, al, weights=w, I0_up=I0)
        assert plan.last_path == "steps"
        assert np.array_equal(J, J_pair)
        assert _rel(J, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=4)) < RTOL
    else:
        dev = torch.device("cuda", 0)
        S32, al32, I032 = (x.astype(np.float32) for x in (S, al, I0))
        Sd, Ad, I0d = (torch.from_numpy(x).to(dev).contiguous() for x in (S32, al32, I032))
        Jd = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
        plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                         dI0_up=I0d.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, f32=True)
        torch.cuda.synchronize()
        assert plan.last_path == "steps"
        ref = orc.J_voronoi(w, th, ph, S32.astype(np.float64), al32.astype(np.float64), so,
                            I0_up=I032.astype(np.float64), nthreads=4)
        assert _rel(Jd.cpu().numpy().astype(np.float64), ref) < 5e-6
    plan.close()


def test_step_path_thread_assignment_does_not_change_results(grids, monkeypatch):
    """Every internal stream count and block-to-XCD mapping of the layer-step path must give
    bit-identical J: the Gauss-Seidel order lives in the visit levels, not in which workgroup runs
    where or when."""
    monkeypatch.setenv("VRT_PATH", "steps")
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(5)
    nlam = 4
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    ks = vrt.quadrature_directions(th, ph)
    out = []
    for env in ({}, {"VRT_STEP_STREAMS": "1"}, {"VRT_STEP_STREAMS": "3", "VRT_STEP_XCD": "0"}, {"VRT_STEP_XCD": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        plan = vrt.FormalPlan(hs, ks, 3)
        J, _ = plan.execute(S, al, weights=w)
        out.append(J.copy())
        plan.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    assert _rel(out[0], orc.J_voronoi(w, th, ph, S, al, so, nthreads=4)) < RTOL
    for J in out[1:]:
        assert np.array_equal(J, out[0])


@pytest.mark.parametrize("dpath", ["steps", "patches"])
def test_diagnostic_environment_variables_cannot_change_results(grids, monkeypatch, dpath):
    """The timing diagnostics that switch memory traffic off live behind -DVRT_DIAG in a separate
    library (voronoirt_amd/libvrt_hip_diag.so, tools/flags_sweep.sh); the product library ignores
    their options, whether preset in the environment of a new plan or set on a live one."""
    monkeypatch.setenv("VRT_PATH", dpath)
    assert _lib.LIB_PATH.endswith("libvrt_hip.so")
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(6)
    nlam = 6
    S = 1 + rng.random((n, nlam))
    al = 1e-6 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J0, _ = plan.execute(S, al, weights=w)
    plan.set_option("VRT_DEBUG_FLAGS", 31)
    plan.set_option("VRT_DEBUG_SKIP_LEVELS", 1)
    J1, _ = plan.execute(S, al, weights=w)
    assert plan.last_path == dpath and np.array_equal(J0, J1)
    monkeypatch.setenv("VRT_DEBUG_FLAGS", "31")
    monkeypatch.setenv("VRT_DEBUG_SKIP_LEVELS", "1")
    monkeypatch.setenv("VRT_TILE_DEBUG", "1")
    plan2 = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    J2, _ = plan2.execute(S, al, weights=w)
    assert np.array_equal(J0, J2)
    assert _rel(J0, orc.J_voronoi(w, th, ph, S, al, so, nthreads=4)) < RTOL
    plan.close()
    plan2.close()


def test_theta_90_is_skipped_when_the_direction_is_inferred_from_k(grids):
    """vrt_plan_create (dirs == NULL) infers the sweep direction from k_z; the library's own
    vrt_direction(90, ϕ) gives k_z = cos(π/2) = 6.1e-17, not 0, and must still count as θ = 90
    (skipped, lambda_iteration.jl:98,104) instead of being solved as a horizontal 'down' ray."""
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(3)
    S = 1 + rng.random((n, 2))
    al = 1e-6 * (1 + rng.random((n, 2)))
    ks = np.stack([vrt.direction(150.0, 20.0), vrt.direction(90.0, 33.0), vrt.direction(40.0, 200.0)])
    assert 0 < ks[1, 0] < 1e-15
    w = np.array([0.3, 0.4, 0.3])
    plan = vrt.FormalPlan(hs, ks, 3)                       # no dirs: inferred
    J, I = plan.execute(S, al, weights=w, want_I=True)
    assert (I[1] == 0).all()                               # the horizontal direction was skipped
    ref = orc.J_voronoi(w, np.array([150.0, 90.0, 40.0]), np.array([20.0, 33.0, 200.0]), S, al, so, nthreads=2)
    assert _rel(J, ref) < RTOL
    plan.close()


def test_non_finite_input_stays_where_the_reference_keeps_it(grids, path):
    """One site with S = Inf: the reference propagates Inf/NaN only to sites that actually read it
    (0 * Inf never arises for an upwind outside the site's own layer, whose intensity reads 0).  All
    three device paths must agree with the oracle on WHICH sites are finite."""
    hs, so = grids["voronoi"]
    n = so.n
    rng = np.random.default_rng(17)
    S = 1 + rng.random(n)
    al = 5 * 10 ** rng.uniform(-3, 1, n)
    bad = int(so.perm_up[so.layers_up[3]] - 1)             # a site in the 4th layer
    S[bad] = np.inf
    k = vrt.direction(140.0, 70.0)
    I0 = rng.random(so.layers_up[1] - 1)
    with np.errstate(all="ignore"):
        ref = orc.Delaunay_upII(k, S, I0, al, so, 3)
    got = vrt.Delaunay_upII(k, S, I0, al, hs, 3)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    m = np.isfinite(ref)
    assert m.sum() > 0.5 * n and (~m).sum() >= 1
    assert _rel(got[m], ref[m]) < RTOL


def test_closing_the_grid_closes_its_plans(bcc_small):
    """VoronoiSites.close() first closes every FormalPlan built on it (a plan holds a pointer to the
    grid); closing such a plan afterwards, or letting it be collected, is a no-op."""
    pos, nbr, bounds = bcc_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    plan = vrt.FormalPlan(hs, [vrt.direction(160.0, 10.0)], 3)
    hs.close()
    assert plan._h is None
    plan.close()
    with pytest.raises(Exception):
        plan.execute(np.ones((hs.n, 1)), np.ones(hs.n))


@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
def test_results_do_not_depend_on_the_storage_order(grid, bcc_small, voro_small, monkeypatch):
    """VRT_STORE_ORDER (strips of rows, default; other strip widths; the Morton curve of rounds 1-4) decides where a site's
    values LIVE in the planes, not the order of the Gauss-Seidel visits (that is the schedule's): J and the per-angle
    intensities are bit for bit the same, on the chained launch and on the per-layer launches."""
    import torch
    pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    n, nlam = so.n, 6
    rng = np.random.default_rng(77)
    S = 1.0 + rng.random((n, nlam))
    al = 10.0 ** rng.uniform(-3, 1, (n, nlam))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    n1 = int(so.layers_up[1] - 1)
    I0 = S[so.perm_up[:n1] - 1].copy()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    dS, dA, dI0 = t(S), t(al), t(I0)
    got = {}
    for order in ("strips", "strips:7", "morton"):
        monkeypatch.setenv("VRT_STORE_ORDER", order)
        hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
        for chain in (1, 0):
            plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
            plan.set_option("VRT_PATH", "patches")
            plan.set_option("VRT_PATCH_CHAIN", chain)
            J = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
            Io = torch.zeros((nq, n, nlam), dtype=torch.float64, device=dev)
            plan.execute_dev(nlam, nlam, dS.data_ptr(), dA.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(), dI0_up=dI0.data_ptr(),
                             dI_out=Io.data_ptr(), stream=st)
            torch.cuda.synchronize()
            assert plan.last_path == "patches"
            got[(order, chain)] = (J.cpu().numpy(), Io.cpu().numpy())
            plan.close()
        hs.close()
    ref = got[("strips", 1)]
    for key, val in got.items():
        assert np.array_equal(val[0], ref[0]) and np.array_equal(val[1], ref[1]), key
    oracle = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
    assert _rel(ref[0], oracle) < RTOL


@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
@pytest.mark.parametrize("nlam, chain", [(6, 1), (12, 0), (13, 0), (20, 0)])
def test_sweep_order_float_S_and_J_are_the_caller_layout_results_bit_for_bit(grids, monkeypatch, grid, nlam, chain):
    """The float forms (vrt_plan_execute_native_dev_f32 and its three layout helpers): S read from and J reduced into
    float plane sets in the plan's float pair blocks.  J_up + J_down equals vrt_plan_execute_dev_f32's J bit for bit --
    per-site, sweep-order per-(site, wavelength) and native per-angle alpha, the four-wavelength kernel (even pair counts)
    and the pair kernel (13 wavelengths: 7 pairs), chained and per-layer launches -- and stays within 5e-6 of the fp64
    oracle; the helpers round-trip."""
    import torch
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so = grids[grid]
    n = so.n
    rng = np.random.default_rng(100 + nlam)
    r32 = lambda a: np.asarray(a, dtype=np.float32)
    S = r32(1.0 + rng.random((n, nlam)))
    al1 = r32(10.0 ** rng.uniform(-3, 1, n))
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    n1 = int(so.layers_up[1] - 1)
    I0 = S[so.perm_up[:n1] - 1].copy()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    plan.set_option("VRT_PATCH_CHAIN", chain)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    dS, dA1, dI0 = t(S), t(al1), t(I0)
    np_ = plan.native_plane_count(nlam)
    S_up, S_dn = (torch.full((np_,), 7.0, dtype=torch.float32, device=dev) for _ in range(2))
    plan.to_native_dev(nlam, nlam, dS.data_ptr(), S_up.data_ptr(), S_dn.data_ptr(), stream=st, f32=True)
    for d, buf in ((1, S_up), (-1, S_dn)):
        back = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
        plan.from_native_dev(d, nlam, nlam, buf.data_ptr(), back.data_ptr(), stream=st, f32=True)
        assert torch.equal(back, dS)
    for mode in ("site", "site_lam", "native"):
        if mode == "site":
            dal, am, dal_c, am_c = dA1, _lib.ALPHA_SITE, dA1, _lib.ALPHA_SITE
            al_ref = np.repeat(al1[:, None], nlam, axis=1)
        elif mode == "site_lam":
            al_ref = r32(al1[:, None] * (1 + 0.02 * np.arange(nlam)[None, :]))
            dal_c, am_c = t(al_ref), _lib.ALPHA_SITE_LAM
            dal = torch.full((2 * np_,), 5.0, dtype=torch.float32, device=dev)
            plan.to_native_dev(nlam, nlam, dal_c.data_ptr(), dal.data_ptr(), dal.data_ptr() + 4 * np_, stream=st, f32=True)
            am = _lib.ALPHA_SITE_LAM_NATIVE
        else:
            al_ref = r32(np.stack([np.repeat(al1[:, None], nlam, axis=1) * (1 + 0.03 * i + 0.01 * np.arange(nlam)[None, :]) for i in range(nq)]))
            d3 = t(al_ref)
            dal = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float32, device=dev)
            plan.alpha_to_native_dev(nlam, nlam, d3.data_ptr(), dal.data_ptr(), stream=st, f32=True)
            am = _lib.ALPHA_ANGLE_NATIVE
            dal_c, am_c = dal, am
        J = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
        plan.execute_dev(nlam, nlam, dS.data_ptr(), dal_c.data_ptr(), am_c, w, dJ=J.data_ptr(), dI0_up=dI0.data_ptr(), stream=st, f32=True)
        launches = plan.last_launches
        J_up, J_dn = (torch.full((np_,), -3.0, dtype=torch.float32, device=dev) for _ in range(2))
        plan.execute_native_dev(nlam, S_up.data_ptr(), S_dn.data_ptr(), dal.data_ptr(), am, w, dJ_up=J_up.data_ptr(),
                                dJ_down=J_dn.data_ptr(), dI0_up=dI0.data_ptr(), stream=st, f32=True)
        assert plan.last_path == "patches" and plan.last_launches == launches
        Jn = torch.zeros((n, nlam), dtype=torch.float32, device=dev)
        plan.J_from_native_dev(nlam, nlam, J_up.data_ptr(), J_dn.data_ptr(), Jn.data_ptr(), stream=st, f32=True)
        torch.cuda.synchronize()
        plan.check()
        assert torch.equal(J, Jn), mode
        if mode != "native" or nlam <= 6:
            ref = orc.J_voronoi(w, th, ph, S.astype(np.float64), al_ref.astype(np.float64), so, I0_up=I0.astype(np.float64), nthreads=8)
            assert _rel(Jn.cpu().numpy().astype(np.float64), ref) < 5e-6
    plan.close()


@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
@pytest.mark.parametrize("nlam, pathopt", [(3, "auto"), (7, "auto"), (22, "auto"), (5, "steps"), (22, "patches-launches")])
def test_sweep_order_S_and_J_are_the_caller_layout_results_bit_for_bit(grids, monkeypatch, grid, nlam, pathopt):
    """vrt_plan_execute_native_dev: S read from and J reduced into the sweep's own per-direction plane sets -- what a
    device-resident Λ-iteration keeps between its steps (lambda_iteration.jl:261-263 produces S, rates.jl:154-201 consume J).
    J_up + J_down equals the J of vrt_plan_execute_dev bit for bit (chained launch, per-layer launches, the steps
    path; odd and even wavelength counts; per-site, sweep-order per-(site, wavelength) and native per-angle alpha), the layout helpers round-trip, an
    up-only plan hands back a zero J_down, and the J itself is checked against the oracle."""
    import torch
    monkeypatch.delenv("VRT_PATH", raising=False)
    hs, so = grids[grid]
    n = so.n
    rng = np.random.default_rng(nlam)
    S = 1.0 + rng.random((n, nlam))
    al1 = 10.0 ** rng.uniform(-3, 1, n)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    n1 = int(so.layers_up[1] - 1)
    I0 = S[so.perm_up[:n1] - 1].copy()
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    if pathopt == "steps":
        plan.set_option("VRT_PATH", "steps")
    if pathopt == "patches-launches":
        plan.set_option("VRT_PATCH_CHAIN", 0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    dS, dA1, dI0 = t(S), t(al1), t(I0)
    np_ = plan.native_plane_count(nlam)
    assert np_ == (nlam + 1) // 2 * 2 * n
    S_up, S_dn = (torch.full((np_,), 7.0, dtype=torch.float64, device=dev) for _ in range(2))
    plan.to_native_dev(nlam, nlam, dS.data_ptr(), S_up.data_ptr(), S_dn.data_ptr(), stream=st)
    # the layout: element (l, pos) at ((l / 2) n + pos) 2 + l % 2, pos along the direction's storage order
    for d, buf in ((1, S_up), (-1, S_dn)):
        order = hs.storage_order(d) - 1
        planes = buf.cpu().numpy().reshape((nlam + 1) // 2, n, 2)
        for l in (0, nlam - 1):
            assert np.array_equal(planes[l // 2, :, l % 2], S[order, l])
        back = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
        plan.from_native_dev(d, nlam, nlam, buf.data_ptr(), back.data_ptr(), stream=st)
        assert torch.equal(back, dS)
    for mode in ("site", "site_lam", "native"):
        dal_caller = None
        if mode == "site":
            dal, am = dA1, _lib.ALPHA_SITE
            al_ref = np.repeat(al1[:, None], nlam, axis=1)
        elif mode == "site_lam":
            # alpha per (site, wavelength) -- the continuum's, lambda_continuum.jl:27-56 -- laid out ONCE in sweep order
            # (both directions' plane sets one behind the other) against the caller-layout array of vrt_plan_execute_dev
            al_ref = al1[:, None] * (1 + 0.02 * np.arange(nlam)[None, :])
            dal_caller = t(al_ref)
            dal = torch.full((2 * np_,), 5.0, dtype=torch.float64, device=dev)
            plan.to_native_dev(nlam, nlam, dal_caller.data_ptr(), dal.data_ptr(), dal.data_ptr() + 8 * np_, stream=st)
            am = _lib.ALPHA_SITE_LAM_NATIVE
        else:
            al3 = np.stack([np.repeat(al1[:, None], nlam, axis=1) * (1 + 0.03 * i + 0.01 * np.arange(nlam)[None, :]) for i in range(nq)])
            d3 = t(al3)
            dal = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
            plan.alpha_to_native_dev(nlam, nlam, d3.data_ptr(), dal.data_ptr(), stream=st)
            am = _lib.ALPHA_ANGLE_NATIVE
            al_ref = al3
        J = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
        if dal_caller is not None:
            plan.execute_dev(nlam, nlam, dS.data_ptr(), dal_caller.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
            with pytest.raises(vrt.VrtError):       # the sweep-order alpha goes with sweep-order S and J only
                plan.execute_dev(nlam, nlam, dS.data_ptr(), dal.data_ptr(), am, w, dJ=J.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
        else:
            plan.execute_dev(nlam, nlam, dS.data_ptr(), dal.data_ptr(), am, w, dJ=J.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
        path_a = plan.last_path
        J_up, J_dn = (torch.full((np_,), -3.0, dtype=torch.float64, device=dev) for _ in range(2))
        plan.execute_native_dev(nlam, S_up.data_ptr(), S_dn.data_ptr(), dal.data_ptr(), am, w, dJ_up=J_up.data_ptr(),
                                dJ_down=J_dn.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
        assert plan.last_path == path_a == ("steps" if pathopt == "steps" else "patches")
        Jn = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
        plan.J_from_native_dev(nlam, nlam, J_up.data_ptr(), J_dn.data_ptr(), Jn.data_ptr(), stream=st)
        torch.cuda.synchronize()
        plan.check()
        assert torch.equal(J, Jn)
        if mode != "native" or nlam <= 7:
            ref = orc.J_voronoi(w, th, ph, S, al_ref, so, I0_up=I0, nthreads=8)
            assert _rel(Jn.cpu().numpy(), ref) < RTOL
    # the caller-layout alphas carry the caller's leading dimension: refused with sweep-order S
    with pytest.raises(vrt.VrtError):
        plan.execute_native_dev(nlam, S_up.data_ptr(), S_dn.data_ptr(), dS.data_ptr(), _lib.ALPHA_SITE_LAM, w,
                                dJ_up=J_up.data_ptr(), dJ_down=J_dn.data_ptr(), stream=st)
    plan.close()
    # a plan with up rays only: J_down comes back as zeros, S_down is not needed
    up_only = [i for i in range(nq) if th[i] > 90]
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th[up_only], ph[up_only]), 3)
    J = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
    plan.execute_dev(nlam, nlam, dS.data_ptr(), dA1.data_ptr(), _lib.ALPHA_SITE, w[up_only], dJ=J.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
    J_up, J_dn = (torch.full((np_,), -3.0, dtype=torch.float64, device=dev) for _ in range(2))
    plan.execute_native_dev(nlam, S_up.data_ptr(), 0, dA1.data_ptr(), _lib.ALPHA_SITE, w[up_only], dJ_up=J_up.data_ptr(),
                            dJ_down=J_dn.data_ptr(), dI0_up=dI0.data_ptr(), stream=st)
    Jn = torch.zeros((n, nlam), dtype=torch.float64, device=dev)
    plan.J_from_native_dev(nlam, nlam, J_up.data_ptr(), J_dn.data_ptr(), Jn.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(J, Jn) and float(J_dn.abs().max()) == 0.0
    plan.close()
