"""The Voronoi path against the reference's own searchlight OUTPUT (qualitatively).

data/searchlight_data/I_160_45_voronoi.npy / I_20_15_voronoi.npy are rasters the reference produced
with Delaunay_upII / Delaunay_downII on 51^3 random sites (src/compare_searchlight.jl:10-152).  Its
sites were unseeded, so the images cannot be reproduced pixel by pixel; their statistics
(tests/golden/voronoi_searchlight_reference_stats.json, derived by make_searchlight_stats.py) still
pin what a restatement can get wrong silently: the direction convention (k points from a site to
its UPWIND side, the beam leaves at 0.5 - k_xy/|k_z|), the weighting of the two upwind neighbours
and the numerical diffusion of the scheme.  The second file predates today's sign convention for
down rays (ϕ = 195° reproduces it; see tests/test_regular.py)."""
import json
import os

import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("I_160_45_voronoi", 160.0, 45.0, True), ("I_20_15_voronoi", 20.0, 195.0, False)]


def _stats(a):
    n = a.shape[0]
    ang = 2 * np.pi * (np.arange(n) + 0.5) / n
    w0, w1 = a.sum(axis=1), a.sum(axis=0)
    c0 = np.angle((w0 * np.exp(1j * ang)).sum()) / (2 * np.pi) % 1
    c1 = np.angle((w1 * np.exp(1j * ang)).sum()) / (2 * np.pi) % 1
    r0 = abs((w0 * np.exp(1j * ang)).sum()) / w0.sum()
    r1 = abs((w1 * np.exp(1j * ang)).sum()) / w1.sum()
    return (c0, c1), (r0, r1), a.mean()


def _radial_profile(a, c0, c1, edges):
    n = a.shape[0]
    g = (np.arange(n) + 0.5) / n
    d0 = np.abs(g - c0) % 1.0
    d0 = np.minimum(d0, 1.0 - d0)
    d1 = np.abs(g - c1) % 1.0
    d1 = np.minimum(d1, 1.0 - d1)
    r = np.sqrt(d0[:, None] ** 2 + d1[None, :] ** 2)
    return np.array([a[(r >= lo) & (r < hi)].mean() for lo, hi in zip(edges[:-1], edges[1:])])


def _circ_dist(a, b):
    d = abs(a - b) % 1.0
    return min(d, 1.0 - d)


@pytest.fixture(scope="module")
def searchlight_grid():
    # the reference's own resolution: 51^3 uniformly random sites in the unit cube
    # (compare_searchlight.jl:11-28); ~20 s of Qhull, seeded
    pos, nbr, bounds = synth.voronoi_grid(51 ** 3, seed=51, margin=0.2)
    return pos, nbr, bounds, orc.make_sites(pos, nbr, bounds)


def _raster(I, pos, z_plane, res=170):
    """Nearest-neighbour raster of the exit plane, compare_searchlight.jl:116-124."""
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    g = np.linspace(0, 1, res)
    X, Y = np.meshgrid(g, g, indexing="ij")
    q = np.stack([np.full(X.size, z_plane), X.ravel(), Y.ravel()], axis=1)
    _, idx = tree.query(q)
    return I[idx].reshape(res, res)


def _run(solver_up, solver_down, sites, pos, theta, phi, up):
    perm, lay = (sites.perm_up, sites.layers_up) if up else (sites.perm_down, sites.layers_down)
    idx = perm[: lay[1] - 1] - 1
    lit = np.sqrt((pos[idx, 1] - 0.5) ** 2 + (pos[idx, 2] - 0.5) ** 2) < 0.1     # :77-82
    n = pos.shape[0]
    k = orc.direction(theta, phi)
    I = (solver_up if up else solver_down)(k, np.zeros(n), lit.astype(float), np.zeros(n), sites, 3)
    return _raster(I, pos, 1.0 if up else 0.0), k


def _check(img, k, ref):
    (c0, c1), (r0, r1), mean = _stats(img)
    # geometric exit point of the beam axis, and the reference image's own centroid
    t = 1.0 / abs(k[0])
    assert _circ_dist(ref["centroid"][0], (0.5 - t * k[1]) % 1) < 0.02
    assert _circ_dist(ref["centroid"][1], (0.5 - t * k[2]) % 1) < 0.02
    # (another realisation of the random sites: measured differences are 0.000-0.008 in the centroid,
    #  0.004-0.02 in the resultant length, 0.02-0.06 in the peak)
    assert _circ_dist(c0, ref["centroid"][0]) < 0.02 and _circ_dist(c1, ref["centroid"][1]) < 0.02
    # beam width (circular resultant length), peak and surviving flux: same diffusion as the reference
    assert abs(r0 - ref["resultant"][0]) < 0.04 and abs(r1 - ref["resultant"][1]) < 0.04
    assert abs(img.max() - ref["max"]) < 0.12      # single-pixel peak: the noisiest statistic (0.73-0.81 seen)
    assert 0.65 < mean / ref["mean"] < 1.35
    assert img.min() >= 0.0 and img.max() <= 1.0 + 1e-12
    # lit area and the radial beam profile around the centroid (mean intensity per annulus): the
    # scheme's numerical diffusion, bin by bin, against what the reference itself produced
    frac = float((img > 0.05).mean())
    prof = _radial_profile(img, c0, c1, ref["radial_edges"])
    print("frac_above_0.05", frac, ref["frac_above_0.05"], "radial", np.round(prof, 4).tolist(),
          np.round(ref["radial_profile"], 4).tolist())
    # measured on this realisation: lit fraction 0.088 / 0.095 vs the reference's 0.102 / 0.095; the
    # (20, 15) profile agrees to 0.05 in every bin, the (160, 45) one to 0.02 in the core (r < 0.06)
    # and the tail with the reference's shoulder (0.06 <= r < 0.12) 0.09-0.12 higher (its sites were
    # unseeded, and the files come from an earlier revision of the reference)
    refp = np.array(ref["radial_profile"])
    assert abs(frac - ref["frac_above_0.05"]) < 0.02
    assert abs(prof[0] - refp[0]) < 0.05 and abs(prof[1] - refp[1]) < 0.05
    assert np.abs(prof - refp).max() < 0.15
    assert (np.diff(prof) < 0).all() and (np.diff(refp) < 0).all()      # monotone falling beam


@pytest.mark.parametrize("name,theta,phi,up", CASES)
def test_oracle_searchlight_matches_reference_image_statistics(searchlight_grid, name, theta, phi, up):
    pos, nbr, bounds, so = searchlight_grid
    ref = json.load(open(os.path.join(GOLDEN, "voronoi_searchlight_reference_stats.json")))[name]
    img, k = _run(orc.Delaunay_upII, orc.Delaunay_downII, so, pos, theta, phi, up)
    _check(img, k, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name,theta,phi,up", CASES)
def test_gpu_searchlight_matches_reference_image_statistics(searchlight_grid, name, theta, phi, up):
    pos, nbr, bounds, so = searchlight_grid
    ref = json.load(open(os.path.join(GOLDEN, "voronoi_searchlight_reference_stats.json")))[name]
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    img, k = _run(vrt.Delaunay_upII, vrt.Delaunay_downII, hs, pos, theta, phi, up)
    _check(img, k, ref)
    hs.close()


@pytest.mark.gpu
def test_gpu_J_on_true_voronoi_grid_at_reference_resolution(searchlight_grid):
    """J_λ_voronoi on the 51^3-site TRUE Voronoi tessellation (irregular layer sizes, 5-30
    neighbours per cell, upwind neighbours in later layers): 12 angles x 24 wavelengths and a single
    wavelength (the continuum caller) on the default path against the oracle."""
    pos, nbr, bounds, so = searchlight_grid
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    n = so.n
    rng = np.random.default_rng(5)
    w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
    plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
    for a in (0, 5, 8):
        up, dots, wt, r, st = orc.upwind_table(so, vrt.direction(th[a], ph[a]))
        gup, gd, gw, gr = plan.upwind(a)
        ok = st == 0
        assert np.array_equal(gup[ok], up[ok]) and np.array_equal(gd[ok], dots[ok])
    for nlam, expect in ((24, "patches"), (1, "patches")):
        S = 1 + rng.random((n, nlam))
        al = 30 * 10 ** rng.uniform(-3, 2, (n, 1)) * (1 + rng.random((n, nlam)))
        I0 = rng.random((so.layers_up[1] - 1, nlam))
        J, _ = plan.execute(S, al, weights=w, I0_up=I0)
        assert plan.last_path == expect
        lam_check = [0, nlam - 1] if nlam > 1 else [0]
        ref = orc.J_voronoi(w, th, ph, S[:, lam_check], al[:, lam_check], so, I0_up=I0[:, lam_check], nthreads=4)
        from oracle.parity import rel
        err = rel(J[:, lam_check], ref)
        assert err < 1e-10, (nlam, err)
    plan.close()
    hs.close()
