"""GPU parity tests (-m gpu) of the fused patch kernel (voronoirt_amd/csrc/vrt_patch.hip, VRT_PATH=patches):
layers cut into patches, each solved with the halo of its in-layer dependency cone by one workgroup.
Against the oracle (1e-10, fp64), against the other device paths, over the kernel's launch shapes
and with patches small enough that every layer of the test grids is split."""
import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import _lib, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-10


from oracle.parity import rel as _rel     # element-wise: |a - b| < tol (|b| + smallest non-zero |b|) for EVERY element


@pytest.fixture(scope="module")
def grids(bcc_small, voro_small):
    out = {}
    for name, (pos, nbr, bounds) in (("bcc", bcc_small), ("voronoi", voro_small)):
        out[name] = (vrt.VoronoiSites(pos, nbr, bounds, device=0), orc.make_sites(pos, nbr, bounds))
    yield out
    for hs, _ in out.values():
        hs.close()


def _case(so, nlam, seed, per_angle=0):
    rng = np.random.default_rng(seed)
    n = so.n
    S = 1 + rng.random((n, nlam))
    al = 10 ** rng.uniform(-3, 3, (n, 1)) * (1 + rng.random((n, nlam))) * 10 / (so.bounds[3] - so.bounds[2])
    if per_angle:
        al = np.stack([al * (1 + 0.1 * rng.random((n, nlam))) for _ in range(per_angle)])
    I0u = rng.random((so.layers_up[1] - 1, nlam))
    I0d = rng.random((so.layers_down[1] - 1, nlam))
    return S, al, I0u, I0d


@pytest.mark.parametrize("name", ["bcc", "voronoi"])
@pytest.mark.parametrize("own", [0, 40, 150])
@pytest.mark.parametrize("nlam", [1, 4, 7])
def test_patches_J_matches_oracle_and_steps(grids, name, own, nlam, monkeypatch):
    """J of ul7n12 x nlam wavelengths: default patch size (one or two patches per layer here) and patches
    of ~40 / ~150 sites (every layer split, halos everywhere); shared and per-angle alpha; up AND down
    boundary intensities.  Equal to the oracle at 1e-10 and to the layer-step path at rounding level."""
    hs, so = grids[name]
    monkeypatch.delenv("VRT_PATH", raising=False)
    if own:
        monkeypatch.setenv("VRT_PATCH_OWN", str(own))          # plan creation reads it
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    for per_angle in (0, nq):
        S, al, I0u, I0d = _case(so, nlam, 3 + nlam, per_angle)
        plan.set_option("VRT_PATH", "patches")
        J, I = plan.execute(S, al, weights=w, I0_up=I0u, I0_down=I0d, want_I=True)
        assert plan.last_path == "patches"
        ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0u, I0_down=I0d, nthreads=4)
        assert _rel(J, ref) < RTOL
        plan.set_option("VRT_PATH", "steps")
        J2, I2 = plan.execute(S, al, weights=w, I0_up=I0u, I0_down=I0d, want_I=True)
        assert plan.last_path == "steps"
        # (element-wise; where an optical depth sits within rounding of 5e-4 the two paths may take different
        # branches of linear_weights, which differ by Δτ³/6 = 2e-11 there: functions.jl:484-500)
        assert _rel(I, I2) < 5e-11 and _rel(J, J2) < 5e-11
        # the never-visited last site of each direction keeps I = 0 (voronoi_utils.jl:266)
        for a_i in range(nq):
            last = (so.perm_up if th[a_i] > 90 else so.perm_down)[-1] - 1
            assert (I[a_i, last] == 0.0).all()
    plan.close()


@pytest.mark.parametrize("shape", [(1, 1, 256), (1, 1, 512), (1, 1, 1024), (2, 1, 256), (2, 1, 512), (1, 2, 256),
                                   (1, 2, 512), (1, 2, 1024), (2, 2, 512)])
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("variant", ["plain", "lean", "chain", "chain-df", "block8", "block8-lean"])
def test_every_patch_kernel_instantiation(grids, shape, f32, variant, monkeypatch):
    """Every (entries per thread, wavelength pairs, threads) instantiation of k_patch_solve, fp64 and fp32
    storage: same results (fp64: to rounding; fp32: 5e-6 against the fp64 oracle on the rounded inputs).
    Variants: the 64-register kernel (VRT_PATCH_LEAN, the default of the (1, 1, NT) shapes; "plain" = the 72-register
    one) with one launch per layer, the chained launch (VRT_PATCH_CHAIN, the default: k_patch_chain; floats with an odd pair
    count as here run the pair kernel inside it; "chain-df": its hand-off by the intensities themselves, VRT_CHAIN_DATAFLAG,
    fp64 only) and the storage layout with 8 wavelength pairs of a site side by side (VRT_PAIR_BLOCK; 9 wavelengths =
    5 pairs: blocks 4 + 1)."""
    import torch
    hs, so = grids["voronoi"]
    K, Q, NT = shape
    chain = variant.startswith("chain")
    if ("lean" in variant or chain) and (K, Q) != (1, 1):
        pytest.skip("the 64-register kernel replaces the (1, 1, NT) shapes")
    if chain and NT != 512:
        pytest.skip("the chained launch exists for 512-thread workgroups")
    if variant == "chain-df" and f32:
        pytest.skip("the data-as-flag hand-off exists for fp64 storage")
    monkeypatch.setenv("VRT_PATCH_LEAN", "1" if ("lean" in variant or chain) else "0")
    monkeypatch.setenv("VRT_PATCH_CHAIN", "1" if chain else "0")
    monkeypatch.setenv("VRT_CHAIN_DATAFLAG", "1" if variant == "chain-df" else "0")
    if chain and f32:          # 5 pairs: floats in blocks of two pairs would need sibling workgroups (per-layer launches)
        monkeypatch.setenv("VRT_PATCH_QUAD", "0")
    monkeypatch.setenv("VRT_PAIR_BLOCK", "8" if "block8" in variant else "4" if "block4" in variant else "1")
    monkeypatch.setenv("VRT_PATCH_K", str(K))
    monkeypatch.setenv("VRT_PATCH_Q", str(Q))
    monkeypatch.setenv("VRT_PATCH_NT", str(NT))
    monkeypatch.setenv("VRT_PATCH_OWN", "90")
    monkeypatch.setenv("VRT_PATCH_TARGET", str(64 if K == 1 else 4096))      # several pairs per workgroup / one
    monkeypatch.setenv("VRT_PATH", "patches")
    n, nlam = so.n, 9
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
    S, al, I0u, I0d = _case(so, nlam, 41)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    dt, npdt = (torch.float32, np.float32) if f32 else (torch.float64, np.float64)
    Sd, Ad, Ud, Dd = (torch.from_numpy(x.astype(npdt)).to(dev).contiguous() for x in (S, al, I0u, I0d))
    Jd = torch.zeros((n, nlam), dtype=dt, device=dev)
    plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jd.data_ptr(),
                     dI0_up=Ud.data_ptr(), dI0_down=Dd.data_ptr(), stream=st, f32=f32)
    torch.cuda.synchronize()
    assert plan.last_path == "patches"
    assert (plan.last_launches == 1) == chain
    r = lambda x: x.astype(npdt).astype(np.float64)
    ref = orc.J_voronoi(w, th, ph, r(S), r(al), so, I0_up=r(I0u), I0_down=r(I0d), nthreads=4)
    assert _rel(Jd.cpu().numpy().astype(np.float64), ref) < (5e-6 if f32 else RTOL)
    plan.close()


@pytest.mark.parametrize("f32", [False, True])
def test_kernel_variants_agree_bit_for_bit(grids, f32, monkeypatch):
    """The chained launch (both hand-offs: progress words, and the intensities as their own flags), the 64-register
    kernel launched per layer and the 72-register one evaluate the same expressions in the same order on the same values: J and the per-angle intensities are bitwise
    equal (fp32 storage, 16 wavelengths: the chained launch runs k_patch_quad's pair loop with VRT_PATCH_QUAD=1)."""
    import torch
    hs, so = grids["bcc"]
    n, nlam = so.n, 16                  # 8 pairs: an even count, so that fp32 storage can take them four wavelengths at a time
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    monkeypatch.setenv("VRT_PATH", "patches")
    monkeypatch.setenv("VRT_PATCH_OWN", "200")
    monkeypatch.setenv("VRT_PATCH_QUAD", "0")
    S, al, I0u, I0d = _case(so, nlam, 23, per_angle=nq)
    dt = torch.float32 if f32 else torch.float64
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev).to(dt).contiguous()
    Sd, Ad, Ud, Dd = f(S), f(al), f(I0u), f(I0d)
    got = {}
    for name, env in (("lean", {"VRT_PATCH_LEAN": "1"}), ("plain", {"VRT_PATCH_LEAN": "0"}),
                      ("chain", {"VRT_PATCH_LEAN": "1", "VRT_PATCH_CHAIN": "1"}),
                      ("chain-df", {"VRT_PATCH_LEAN": "1", "VRT_PATCH_CHAIN": "1", "VRT_CHAIN_DATAFLAG": "1"}),
                      ("chain-quad", {"VRT_PATCH_LEAN": "1", "VRT_PATCH_CHAIN": "1", "VRT_PATCH_QUAD": "1"}),
                      ("lean-quad", {"VRT_PATCH_LEAN": "1", "VRT_PATCH_QUAD": "1"})):
        if ("quad" in name and not f32) or (name == "chain-df" and f32):
            continue
        for k in ("VRT_PATCH_LEAN", "VRT_PATCH_CHAIN", "VRT_PATCH_QUAD", "VRT_CHAIN_DATAFLAG"):
            monkeypatch.setenv(k, env.get(k, "0"))
        plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
        Jd = torch.full((n, nlam), float("nan"), dtype=dt, device=dev)
        Id = torch.full((nq, n, nlam), float("nan"), dtype=dt, device=dev)
        plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), _lib.ALPHA_ANGLE_SITE_LAM, w, dJ=Jd.data_ptr(),
                         dI0_up=Ud.data_ptr(), dI0_down=Dd.data_ptr(), dI_out=Id.data_ptr(), stream=st, f32=f32)
        torch.cuda.synchronize()
        assert plan.last_path == "patches"
        assert (plan.last_launches == 1) == ("chain" in name)
        got[name] = (Jd.cpu().numpy(), Id.cpu().numpy())
        plan.close()
    for name in got:
        assert np.array_equal(got[name][0], got["lean"][0]), name
        assert np.array_equal(got[name][1], got["lean"][1]), name


@pytest.mark.parametrize("pair_block", [1, 4])
@pytest.mark.parametrize("mode", ["site", "site_lam", "angle"])
def test_fp32_quad_kernel_equals_pair_kernel(grids, pair_block, mode, monkeypatch):
    """fp32 storage: k_patch_quad (two neighbouring wavelength pairs per lane, 16-byte accesses, option
    VRT_PATCH_QUAD) gives the pair kernel's J bit for bit -- 12 wavelengths = 6 pairs, in blocks of 2, 2, 2
    (VRT_PAIR_BLOCK=1 -> 2 for floats) and 4, 2 -- and both stay within 5e-6 of the fp64 oracle; an odd pair count
    (10 wavelengths) runs on the pair kernel."""
    import torch
    hs, so = grids["voronoi"]
    n = so.n
    w, th, ph, nq = vrt.read_quadrature("ul2n3.dat")
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    monkeypatch.setenv("VRT_PATH", "patches")
    monkeypatch.setenv("VRT_PAIR_BLOCK", str(pair_block))
    monkeypatch.setenv("VRT_PATCH_OWN", "120")
    for nlam in (12, 10):
        S, al, I0u, I0d = _case(so, nlam, 17 + nlam)
        if mode == "site":
            al, amode = al[:, 0].copy(), _lib.ALPHA_SITE
        elif mode == "angle":
            rng = np.random.default_rng(3)
            al, amode = al[None, :, :] * rng.uniform(0.5, 2.0, (nq, 1, nlam)), _lib.ALPHA_ANGLE_SITE_LAM
        else:
            amode = _lib.ALPHA_SITE_LAM
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        Sd, Ad, Ud, Dd = f(S), f(al), f(I0u), f(I0d)
        got = {}
        for quad in (1, 0):
            monkeypatch.setenv("VRT_PATCH_QUAD", str(quad))
            plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3, dirs=[1 if t > 90 else -1 for t in th])
            assert plan.native_pair_block_f32 == (max(pair_block, 2) if quad else pair_block)
            Jd = torch.full((n, nlam), float("nan"), dtype=torch.float32, device=dev)
            plan.execute_dev(nlam, nlam, Sd.data_ptr(), Ad.data_ptr(), amode, w, dJ=Jd.data_ptr(),
                             dI0_up=Ud.data_ptr(), dI0_down=Dd.data_ptr(), stream=st, f32=True)
            torch.cuda.synchronize()
            assert plan.last_path == "patches"
            got[quad] = Jd.cpu().numpy()
            plan.close()
        assert np.array_equal(got[0], got[1])
        r = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
        ref = orc.J_voronoi(w, th, ph, r(S), r(al), so, I0_up=r(I0u), I0_down=r(I0d), nthreads=4)
        assert _rel(got[1].astype(np.float64), ref) < 5e-6


def test_patches_single_solves_and_sweep_counts(grids, monkeypatch):
    """Delaunay_upII / Delaunay_downII (one problem, one wavelength) and n_sweeps 1, 2, 4 on the patch path."""
    monkeypatch.setenv("VRT_PATH", "patches")      # read when the grid creates the plans of the single solves
    monkeypatch.setenv("VRT_PATCH_OWN", "64")
    hs, so = grids["bcc"]
    n = so.n
    rng = np.random.default_rng(5)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-3, 3, n) / (so.bounds[3] - so.bounds[2]) * 10
    for n_sweeps in (1, 2, 3, 4):
        for t, p in ((109.7, 193.6), (70.3, 346.4), (152.7, 315.5)):
            k = vrt.direction(t, p)
            if t > 90:
                I0 = rng.random(so.layers_up[1] - 1)
                got = vrt.Delaunay_upII(k, S, I0, alpha, hs, n_sweeps)
                ref = orc.Delaunay_upII(k, S, I0, alpha, so, n_sweeps)
            else:
                I0 = rng.random(so.layers_down[1] - 1)
                got = vrt.Delaunay_downII(k, S, I0, alpha, hs, n_sweeps)
                ref = orc.Delaunay_downII(k, S, I0, alpha, so, n_sweeps)
            assert _rel(got, ref) < RTOL, (n_sweeps, t)


def test_patches_on_a_tessellated_stratified_grid(monkeypatch):
    """vrt.voro (the in-process tessellation, rt_preprocessing/output_sites.cc:35-49) -> VoronoiSites ->
    J on the patch path: 20 000 density-stratified sites (src/sample_grids.jl:223-230)."""
    monkeypatch.setenv("VRT_PATH", "patches")
    monkeypatch.setenv("VRT_PATCH_OWN", "300")
    n = 20000
    bounds = (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6)
    rng = np.random.default_rng(4)
    u = rng.random(n)
    Lz, H = bounds[1] - bounds[0], 2.0e6
    pos = np.stack([bounds[0] - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H))),
                    bounds[2] + rng.random(n) * (bounds[3] - bounds[2]),
                    bounds[4] + rng.random(n) * (bounds[5] - bounds[4])], axis=1)
    nbr = vrt.voro(pos, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    so = orc.make_sites(pos, nbr, bounds)
    nlam = 6
    S, al = synth.synthetic_fields(pos, bounds, nlam, seed=9)
    I0 = S[so.perm_up[: so.layers_up[1] - 1] - 1]
    J = vrt.J_lambda_voronoi(S, al, hs, "ul7n12.dat", I0_up=I0)
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    ref = orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0, nthreads=8)
    assert _rel(J, ref) < RTOL
    assert list(hs._plans.values())[0].last_path == "patches"
    hs.close()
