"""Regular-grid short characteristics: the oracle against the REFERENCE'S OWN
committed outputs, and the HIP kernel against the oracle.

tests/golden/I_160_45_regular.npy and I_20_15_regular.npy are data files the reference holds
(data/searchlight_data/, loaded by its python/plot_searchlight.py): the top / bottom plane of
`searchlight_regular` (src/compare_searchlight.jl:154-225: 51^3 unit cube, alpha = S = 0, a disk of
radius 0.1 lit on the boundary plane with xi = i/nx, ghost border stripped) for (θ, ϕ) = (160°, 45°)
up and (20°, 15°) down.  They were produced by an earlier revision of the reference (those angles
are not in today's quadrature file); its down-ray direction had k_x, k_y negated relative to
today's k = [cos θ, cos ϕ sin θ, sin ϕ sin θ], i.e. the second file is reproduced with ϕ = 195°.
Both are outputs of the reference itself, so this pins the oracle's regular solver (bilinear,
linear_weights, trapezoidal, plane marching, ghost zones) to reference arithmetic."""
import os

import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _searchlight(n=51):
    z = np.linspace(0, 1, n)
    x = np.linspace(0, 1, n)
    y = np.linspace(0, 1, n)
    S = np.zeros((n, n, n))
    al = np.zeros((n, n, n))
    I0 = np.zeros((n, n))                       # (ny, nx)
    for i in range(1, n + 1):                   # compare_searchlight.jl:180-190
        for j in range(1, n + 1):
            if np.sqrt((i / n - 0.5) ** 2 + (j / n - 0.5) ** 2) < 0.1:
                I0[j - 1, i - 1] = 1.0
    return z, x, y, S, al, I0


def _exit_plane(I, up):
    plane = I[:, :, -1] if up else I[:, :, 0]   # I[end, :, :] / I[1, :, :]
    return plane[1:-1, 1:-1].T                  # [:, 2:end-1, 2:end-1], indexed [ix, iy] like the .npy


@pytest.mark.parametrize("fname,theta,phi,up", [("I_160_45_regular.npy", 160.0, 45.0, True),
                                                  ("I_20_15_regular.npy", 20.0, 195.0, False)])
def test_oracle_reproduces_reference_searchlight_outputs(fname, theta, phi, up):
    z, x, y, S, al, I0 = _searchlight()
    assert I0.sum() == 80.0                     # "Bottom: 80" (compare_searchlight.jl:209)
    ref = np.load(os.path.join(GOLDEN, fname))
    k = orc.direction(theta, phi)
    f = orc.short_characteristics_up if up else orc.short_characteristics_down
    I, kinds = f(k, S, I0, al, z, x, y, 3, return_planes=True)
    assert set(kinds.tolist()) == {0, 1}        # these rays cut the xy planes only
    got = _exit_plane(I, up)
    assert got.shape == ref.shape == (49, 49)
    assert np.abs(got - ref).max() < 1e-15      # a few ulp of the reference's own output
    assert abs(got.sum() - ref.sum()) < 1e-13 and abs(ref.sum() - 80.0) < 1e-12   # conservation


def _random_problem(nz, nx, ny, seed):
    rng = np.random.default_rng(seed)
    z = np.cumsum(rng.uniform(0.5, 1.5, nz)) / nz        # non-uniform z like a real atmosphere
    x = np.linspace(0, 1, nx)
    y = np.linspace(0, 1.3, ny)
    S = 1 + rng.random((ny, nx, nz))
    al = 10 ** rng.uniform(-3, 3, (ny, nx, nz)) * 5
    I0 = rng.random((ny, nx))
    return z, x, y, S, al, I0


def test_oracle_plane_kinds_cover_all_six_kernels():
    z, x, y, S, al, I0 = _random_problem(9, 12, 10, 1)
    seen = set()
    for theta, phi in ((170.0, 30.0), (100.0, 10.0), (100.0, 80.0), (10.0, 200.0), (80.0, 190.0), (80.0, 100.0)):
        k = orc.direction(theta, phi)
        f = orc.short_characteristics_up if theta > 90 else orc.short_characteristics_down
        I, kinds = f(k, S, I0, al, z, x, y, 3, return_planes=True)
        assert np.isfinite(I).all()
        seen |= {(theta > 90, int(c)) for c in kinds if c}
    assert seen == {(True, 1), (True, 2), (True, 3), (False, 1), (False, 2), (False, 3)}


@pytest.mark.gpu
@pytest.mark.parametrize("fname,theta,phi,up", [("I_160_45_regular.npy", 160.0, 45.0, True),
                                                  ("I_20_15_regular.npy", 20.0, 195.0, False)])
def test_gpu_reproduces_reference_searchlight_outputs(fname, theta, phi, up):
    z, x, y, S, al, I0 = _searchlight()
    ref = np.load(os.path.join(GOLDEN, fname))
    k = vrt.direction(theta, phi)
    f = vrt.short_characteristics_up if up else vrt.short_characteristics_down
    got = _exit_plane(f(k, S, I0, al, z, x, y, 3), up)
    assert np.abs(got - ref).max() < 1e-15
    assert abs(got.sum() - 80.0) < 1e-12


@pytest.mark.gpu
def test_gpu_regular_matches_oracle_all_plane_kinds():
    """Random S, α (all three linear_weights branches), non-uniform z, nx != ny, every one of the
    six per-plane kernels; batched call with shared and per-solve fields."""
    z, x, y, S, al, I0 = _random_problem(11, 14, 12, 2)
    angles = [(170.0, 30.0), (100.0, 10.0), (100.0, 80.0), (10.0, 200.0), (80.0, 190.0), (80.0, 100.0),
              (125.0, 300.0), (55.0, 250.0), (180.0, 0.0)]
    ks = np.stack([vrt.direction(t, p) for t, p in angles])
    ups = [t > 90 for t, _ in angles]
    rng = np.random.default_rng(3)
    I0s = rng.random((len(angles),) + I0.shape)
    got = vrt.short_characteristics_batch(ks, ups, S, I0s, al, z, x, y, 3)
    for j, (t, p) in enumerate(angles):
        f = orc.short_characteristics_up if ups[j] else orc.short_characteristics_down
        ref = f(ks[j], S, I0s[j], al, z, x, y, 3)
        assert np.abs(got[j] - ref).max() / np.abs(ref).max() < 1e-12, (t, p)
    # per-solve S and alpha, other sweep counts
    Ss = np.stack([S * (1 + 0.1 * j) for j in range(3)])
    als = np.stack([al * (1 + 0.3 * j) for j in range(3)])
    for n_sweeps in (1, 2):
        got = vrt.short_characteristics_batch(ks[1:4], ups[1:4], Ss, I0s[1:4], als, z, x, y, n_sweeps)
        for j in range(3):
            f = orc.short_characteristics_up if ups[1 + j] else orc.short_characteristics_down
            ref = f(ks[1 + j], Ss[j], I0s[1 + j], als[j], z, x, y, n_sweeps)
            assert np.abs(got[j] - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.gpu
def test_gpu_regular_errors():
    z, x, y, S, al, I0 = _random_problem(5, 6, 6, 4)
    with pytest.raises(vrt.VrtError):
        vrt.short_characteristics_up([0.0, 1.0, 0.0], S, I0, al, z, x, y)      # horizontal ray
    with pytest.raises(vrt.VrtError):
        vrt.short_characteristics_up([0.5, 0.5, 0.5], S, I0, al, z, x, y)      # not a unit vector
    with pytest.raises(ValueError):
        vrt.short_characteristics_up(vrt.direction(150, 20), S[:-1], I0, al, z, x, y)


@pytest.mark.gpu
def test_gpu_regular_device_resident_handle():
    """vrt_regular_*: device pointers in, device pointers out, handle reused across calls with a
    growing batch (grow-only workspaces), same numbers as the host-pointer entry and the oracle."""
    import torch
    z, x, y, S, al, I0 = _random_problem(10, 13, 11, 7)
    angles = [(170.0, 30.0), (100.0, 10.0), (80.0, 100.0), (10.0, 200.0)]
    ks = np.stack([vrt.direction(t, p) for t, p in angles])
    ups = [t > 90 for t, _ in angles]
    rng = np.random.default_rng(8)
    I0s = rng.random((len(angles),) + I0.shape)
    dev = torch.device("cuda", 0)
    dS, dA = torch.as_tensor(S, device=dev), torch.as_tensor(al, device=dev)
    solver = vrt.RegularSolver(z, x, y, device=0)
    for ns in (1, 4, 2):
        dI0 = torch.as_tensor(I0s[:ns].copy(), device=dev)
        dI = torch.full((ns,) + S.shape, np.nan, device=dev, dtype=torch.float64)
        solver.execute_dev(ks[:ns], ups[:ns], dS.data_ptr(), 0, dA.data_ptr(), 0, dI0.data_ptr(), dI.data_ptr(),
                           3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert solver.last_solve_ms() > 0
        got = dI.cpu().numpy()
        for j in range(ns):
            f = orc.short_characteristics_up if ups[j] else orc.short_characteristics_down
            ref = f(ks[j], S, I0s[j], al, z, x, y, 3)
            assert np.abs(got[j] - ref).max() / np.abs(ref).max() < 1e-12
    # field_period: solves ordered direction-major share one (S, alpha) pair per wavelength
    Ss = np.stack([S * (1 + 0.1 * j) for j in range(2)])
    als = np.stack([al * (1 + 0.3 * j) for j in range(2)])
    dSs, dAs = torch.as_tensor(Ss, device=dev), torch.as_tensor(als, device=dev)
    k4 = np.repeat(ks[:2], 2, axis=0)
    up4 = [ups[0], ups[0], ups[1], ups[1]]
    dI0 = torch.as_tensor(I0s[:4].copy(), device=dev)
    dI = torch.full((4,) + S.shape, np.nan, device=dev, dtype=torch.float64)
    vol = S.size
    solver.execute_dev(k4, up4, dSs.data_ptr(), vol, dAs.data_ptr(), vol, dI0.data_ptr(), dI.data_ptr(), 3,
                       torch.cuda.current_stream().cuda_stream, field_period=2)
    torch.cuda.synchronize()
    got = dI.cpu().numpy()
    for j in range(4):
        f = orc.short_characteristics_up if up4[j] else orc.short_characteristics_down
        ref = f(k4[j], Ss[j % 2], I0s[j], als[j % 2], z, x, y, 3)
        assert np.abs(got[j] - ref).max() / np.abs(ref).max() < 1e-12
    solver.close()


@pytest.mark.gpu
def test_gpu_regular_rows_longer_than_the_workgroup(monkeypatch):
    """Rows with more points than threads take the strided (non-pipelined) row loop of the
    yz/xz kernels; VRT_REG_THREADS=64 forces that on a 70 x 75 plane."""
    monkeypatch.setenv("VRT_REG_THREADS", "64")
    z, x, y, S, al, I0 = _random_problem(6, 72, 77, 11)
    angles = [(100.0, 10.0), (100.0, 80.0), (80.0, 190.0), (80.0, 100.0), (170.0, 30.0)]
    ks = np.stack([vrt.direction(t, p) for t, p in angles])
    ups = [t > 90 for t, _ in angles]
    rng = np.random.default_rng(12)
    I0s = rng.random((len(angles),) + I0.shape)
    got = vrt.short_characteristics_batch(ks, ups, S, I0s, al, z, x, y, 3)
    for j in range(len(angles)):
        f = orc.short_characteristics_up if ups[j] else orc.short_characteristics_down
        ref = f(ks[j], S, I0s[j], al, z, x, y, 3)
        assert np.abs(got[j] - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(7, 62, 62), (5, 40, 23), (4, 72, 77), (3, 3, 3), (4, 150, 140)])
def test_gpu_regular_xy_batches_split_form(monkeypatch, shape):
    """Batches whose planes are all of the xy kind (steep rays) run as k_reg_xy_coefs over the whole chip + one light
    march per solve (upwind plane in LDS, or read back from memory when two planes do not fit): the same expressions
    in the same order as the plane loop of k_regular_solve (xy_up_ray :191-278, xy_down_ray :288-372), so all three
    forms agree bit for bit -- and with the oracle to 1e-12."""
    nz, nx, ny = shape
    z, x, y, S, al, I0 = _random_problem(nz, nx, ny, 21)
    z = np.cumsum(0.2 + 0.05 * np.arange(nz)) * (x[1] - x[0])            # thin planes: every steep ray cuts xy first
    angles = [(175.0, 30.0), (6.0, 200.0), (172.0, 300.0), (180.0, 0.0), (9.0, 100.0)]
    ks = np.stack([vrt.direction(t, p) for t, p in angles])
    ups = [t > 90 for t, _ in angles]
    rng = np.random.default_rng(5)
    I0s = rng.random((len(angles),) + I0.shape)
    got = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("VRT_REG_XY", mode)
        got[mode] = vrt.short_characteristics_batch(ks, ups, S, I0s, al, z, x, y, 3)
    assert np.array_equal(got["0"], got["1"]) and np.array_equal(got["0"], got["2"])
    for j in range(len(angles)):
        f = orc.short_characteristics_up if ups[j] else orc.short_characteristics_down
        ref, kinds = f(ks[j], S, I0s[j], al, z, x, y, 3, return_planes=True)
        assert {int(c) for c in kinds if c} == {1}
        assert np.abs(got["1"][j] - ref).max() / np.abs(ref).max() < 1e-12


# ---- analytic known answers that follow from the code (no reference run needed) ---------------------
_KAT_ANGLES = ((170.0, 30.0), (100.0, 10.0), (100.0, 80.0), (10.0, 200.0), (80.0, 190.0), (80.0, 100.0))


def _uniform_field_cases():
    """(label, alpha) for a uniform source function S = I_0 = B on the random non-uniform grid:
    * steep rays (xy planes): I = e I_u + a S_u + b S_c with a + b + e = 1 in every branch of
      linear_weights (functions.jl:484-500) and bilinear weights that sum to 1 -> I == B for ANY α;
    * every plane kind in the thick limit Δτ > 50: e = 0, a + b = 1 -> I == B whatever the carried
      row of the yz / xz kernels holds."""
    z, x, y, S, al, I0 = _random_problem(9, 12, 10, 4)
    B = 2.75
    S[:], I0[:] = B, B
    thick = np.full_like(al, 5e4)                      # Δτ = r (α_c + α_u)/2 >> 50 on this grid
    return z, x, y, S, I0, B, (("any", al), ("thick", thick))


def _check_uniform(solve_up, solve_down, direction):
    z, x, y, S, I0, B, cases = _uniform_field_cases()
    for label, al in cases:
        for theta, phi in _KAT_ANGLES:
            if label == "any" and not (theta > 150 or theta < 30):
                continue                               # inclined rays: only the thick limit is exact
            f = solve_up if theta > 90 else solve_down
            I = f(direction(theta, phi), S, I0, al, z, x, y, 3)
            assert np.abs(I / B - 1).max() < 4e-15, (label, theta, phi)


def test_oracle_uniform_field_stays_uniform():
    _check_uniform(orc.short_characteristics_up, orc.short_characteristics_down, orc.direction)


@pytest.mark.gpu
def test_gpu_uniform_field_stays_uniform():
    _check_uniform(vrt.short_characteristics_up, vrt.short_characteristics_down, vrt.direction)
