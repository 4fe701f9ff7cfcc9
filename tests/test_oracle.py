"""CPU tests of the oracle (oracle/vrt_oracle.c): known answers that follow from the reference's
code, hand-built cases for each quirk, and the committed golden vectors.

The reference has no numeric fixture for the Voronoi path and cannot be run here, so these pin
the restatement by construction ("parity unpinned", see the oracle header)."""
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc
from voronoirt_amd import synth


# ---- linear_weights, src/functions.jl:484-500 ------------------------------------------------
def test_linear_weights_branches():
    a, b, e = orc.linear_weights(0.0)
    assert (a, b, e) == (0.0, 0.0, 1.0)
    dt = 1e-4                                  # Taylor branch
    a, b, e = orc.linear_weights(dt)
    assert e == 1 - dt + 0.5 * dt * dt
    assert a == dt * (0.5 - dt / 3) and b == dt * (0.5 - dt / 6)
    dt = 1.0                                   # exp branch
    a, b, e = orc.linear_weights(dt)
    assert e == math.exp(-1.0)
    assert a == (1 - e) / dt - e and b == 1 - a - e
    dt = 100.0                                 # optically thick branch
    a, b, e = orc.linear_weights(dt)
    assert (a, b, e) == (0.01, 0.99, 0.0)
    # thresholds are strict: 5e-4 and 50 themselves take the exp branch
    for dt in (5e-4, 50.0):
        a, b, e = orc.linear_weights(dt)
        assert e == math.exp(-dt)
    # weights always sum to 1 - e up to rounding (a + b + e = 1: S = I = const is preserved)
    for dt in (1e-7, 3e-4, 0.02, 7.0, 49.0, 51.0, 1e4):
        a, b, e = orc.linear_weights(dt)
        assert abs(a + b + e - 1.0) < 1e-6


# ---- reduce_layers, src/voronoi_utils.jl:253-269 ---------------------------------------------
def test_reduce_layers_example_and_quirk():
    r = orc.reduce_layers(np.array([1, 1, 2, 2, 3, 3]))
    assert r.tolist() == [1, 3, 5, 6]          # r[end] = n, not n + 1


def test_sortperm_is_stable():
    layers = np.array([2, 1, 3, 1, 2, 1, 3])
    assert orc.sortperm(layers).tolist() == [2, 4, 6, 1, 5, 3, 7]


# ---- direction, src/lambda_iteration.jl:87 ----------------------------------------------------
def test_direction_n1():
    k = orc.direction(180.0, 0.0)              # quadratures/n1.dat
    assert k[0] == -1.0 and k[2] == 0.0 and abs(k[1] - 1.2246467991473532e-16) < 1e-30
    k = orc.direction(70.292581108446825, 346.412955051617416)
    assert abs(np.linalg.norm(k) - 1) < 1e-15 and k[0] > 0   # θ < 90: k_z > 0, ray travels down


# ---- smallest_angle, src/voronoi_utils.jl:360-396 ----------------------------------------------
def _star_sites(dirs, ids=None):
    """One centre site (id 1) whose neighbours sit at the given unit directions (z, x, y)."""
    dirs = np.asarray(dirs, dtype=float)
    m = dirs.shape[0]
    pos = np.zeros((m + 1, 3))
    pos[0] = (0.5, 0.5, 0.5)
    pos[1:] = pos[0] + 0.1 * dirs
    nbr = np.zeros((m + 1, m + 1), dtype=np.int64)
    nbr[0, 0] = m
    nbr[1:, 0] = np.arange(2, m + 2) if ids is None else ids
    for j in range(1, m + 1):
        nbr[0, j] = 1
        nbr[1, j] = 1
    lines = orc.delaunay_lines(pos, nbr, 0.0, 1.0, 0.0, 1.0)
    return orc.OracleSites(pos, nbr, lines, None, None, None, None, (0, 1, 0, 1, 0, 1), m + 1)


def test_smallest_angle_discards_old_best():
    # dots along k = -z: 0.5, 0.9, 0.7  -> slot 1 = 0.9 (3rd id), old best 0.5 is DISCARDED,
    # slot 2 = 0.7 (a plain top-2 would also give 0.7)
    def unit(c):
        return [-c, math.sqrt(1 - c * c), 0.0]
    s = _star_sites([unit(0.5), unit(0.9), unit(0.7)])
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert rc == 0 and idx.tolist() == [3, 4]
    # increasing sequence 0.5, 0.7, 0.9: every element is a new best, nothing is ever demoted,
    # slot 2 stays empty -> duplicated slot 1 with dot 0 (a plain top-2 would give 0.7)
    s = _star_sites([unit(0.5), unit(0.7), unit(0.9)])
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert idx.tolist() == [4, 4] and dots[1] == 0.0 and abs(dots[0] - 0.9) < 1e-15
    # decreasing sequence 0.9, 0.7, 0.5: slot 2 = first runner-up
    s = _star_sites([unit(0.9), unit(0.7), unit(0.5)])
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert idx.tolist() == [2, 3]
    # negative runner-up is replaced by slot 1 with weight 0 (:390-393)
    s = _star_sites([unit(0.9), unit(-0.2)])
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert idx.tolist() == [2, 2] and dots[1] == 0.0


def test_smallest_angle_skips_walls_and_ties_keep_first():
    # exact tie (binary-exact offsets): the first neighbour stays in slot 1, the equal second one
    # is not a strict record and lands in slot 2
    pos = np.array([[0.5, 0.5, 0.5], [0.375, 0.75, 0.5], [0.375, 0.25, 0.5], [0.375, 0.5, 0.5]])
    nbr = np.zeros((4, 4), dtype=np.int64)
    nbr[:, 0] = [2, 2, 3, 0]
    nbr[0, 1:] = 1
    nbr[1, 1:] = 1
    lines = orc.delaunay_lines(pos, nbr, 0.0, 1.0, 0.0, 1.0)
    s = orc.OracleSites(pos, nbr, lines, None, None, None, None, (0, 1, 0, 1, 0, 1), 4)
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert idx.tolist() == [2, 3] and dots[0] == dots[1]
    # wall entries (<= 0) are skipped: row = [-5, 2, 4]; site 4 is straight below (dot 1)
    nbr[:, 0] = [3, -5, 2, 4]
    lines = orc.delaunay_lines(pos, nbr, 0.0, 1.0, 0.0, 1.0)
    s = orc.OracleSites(pos, nbr, lines, None, None, None, None, (0, 1, 0, 1, 0, 1), 4)
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert rc == 0 and idx.tolist() == [4, 4] and dots.tolist() == [1.0, 0.0]
    # a site whose only entries are walls has no upwind neighbour at all
    nbr[:, 0] = [2, -5, -6, 0]
    s = orc.OracleSites(pos, nbr, lines, None, None, None, None, (0, 1, 0, 1, 0, 1), 4)
    rc, dots, idx = orc.smallest_angle(0, s, [-1.0, 0.0, 0.0])
    assert rc == -1


# ---- calc_Delaunay_lines, src/voronoi_utils.jl:186-245 ------------------------------------------
def test_delaunay_lines_periodic_shift_and_mirror():
    # site A near the right x edge, B near the left edge: from A, B is shifted by +Lx (correct
    # image); from B, A is MIRRORED about the box centre (reference quirk, :221-222)
    pos = np.array([[0.5, 0.95, 0.5], [0.5, 0.05, 0.5]])
    nbr = np.array([[1, 1], [2, 1]], dtype=np.int64)
    lines = orc.delaunay_lines(pos, nbr, 0.0, 1.0, 0.0, 1.0)
    assert np.allclose(lines[0, 0], [0.0, 1.0, 0.0])      # A -> B image at x = 1.05
    # B -> A: the mirror image x_min + x_max - 0.95 lands (up to rounding) ON B itself, the
    # degenerate outcome of the quirk; the second case below has a finite offset
    pos = np.array([[0.5, 0.9, 0.5], [0.6, 0.05, 0.5]])
    lines = orc.delaunay_lines(pos, nbr, 0.0, 1.0, 0.0, 1.0)
    d = np.array([0.1, (1.0 + 0.05) - 0.9, 0.0])
    assert np.allclose(lines[0, 0], d / np.linalg.norm(d))
    d = np.array([-0.1, (0.0 + 1.0 - 0.9) - 0.05, 0.0])    # mirror: 0.1 - 0.05
    assert np.allclose(lines[1, 0], d / np.linalg.norm(d))


# ---- layering + file parsing, src/voronoi_utils.jl:36-174 ---------------------------------------
def test_layers_simple_column():
    pos, nbr, bounds = synth.regular_lattice_grid(3, 3, 5)
    s = orc.make_sites(pos, nbr, bounds)
    assert s.layers_up.tolist() == [1, 10, 19, 28, 37, 45]
    assert s.layers_down.tolist() == [1, 10, 19, 28, 37, 45]
    # layer 1 (up) = the 9 sites of the bottom plane in ascending id order
    z = pos[s.perm_up[:9] - 1, 0]
    assert np.allclose(z, 0.1) and (np.diff(s.perm_up[:9]) > 0).all()


def test_read_neighbours_roundtrip(tmp_path, voro_small):
    pos, nbr, bounds = voro_small
    f = tmp_path / "nb.txt"
    synth.write_neighbours_file(str(f), nbr, seed=3)
    M = orc.read_neighbours(str(f), pos.shape[0])
    assert np.array_equal(M, nbr)


# ---- Delaunay_upII / downII known answers --------------------------------------------
def test_kat_transparent_uniform_boundary(bcc_small):
    """α = S = 0: linear_weights(0) = (0, 0, 1), so every visit forms a convex combination of
    upwind values.  A uniformly lit boundary therefore gives 0 <= I <= 1, the never-visited last
    site keeps I = 0 (voronoi_utils.jl:266), and I = 1 wherever the three Gauss-Seidel sweeps
    have propagated the boundary value through the in-layer neighbours (almost everywhere)."""
    pos, nbr, bounds = bcc_small
    s = orc.make_sites(pos, nbr, bounds)
    n = s.n

    def check(I, perm, steep):
        assert I[perm[-1] - 1] == 0.0
        assert I.min() >= 0.0 and I.max() <= 1.0 + 1e-14
        if steep:   # in-layer upwind weights are tiny: three sweeps carry the boundary value up
            assert np.median(I) > 1 - 1e-14 and (np.abs(I - 1) < 1e-3).mean() > 0.95

    for theta, phi, steep in ((152.666292044518485, 315.475247829748128, True),
                              (109.707418891553175, 193.587, False)):
        k = orc.direction(theta, phi)
        I = orc.Delaunay_upII(k, np.zeros(n), np.ones(s.layers_up[1] - 1), np.zeros(n), s, 3)
        check(I, s.perm_up, steep)
    k = orc.direction(27.3, 135.0)
    I = orc.Delaunay_downII(k, np.zeros(n), np.ones(s.layers_down[1] - 1), np.zeros(n), s, 3)
    check(I, s.perm_down, True)


def _one_based(a):
    return [0] + list(a)


@pytest.mark.parametrize("kind", ["bcc", "voronoi"])
def test_c_oracle_matches_pure_python_restatement(kind):
    """oracle/pyref.py is an independent loop-for-loop transcription of the Julia source; the C
    oracle must agree with it exactly on layers/perm/offsets/upwind ids and to rounding on I."""
    from oracle import pyref
    if kind == "bcc":
        pos, nbr, bounds = synth.bcc_grid(4, 5, seed=9)
    else:
        pos, nbr, bounds = synth.voronoi_grid(220, seed=4, bounds=(0.0, 1.5, 0.0, 1.0, 0.0, 1.0))
    n = pos.shape[0]
    s = orc.make_sites(pos, nbr, bounds)
    P = [None] + [[0.0] + list(map(float, pos[i])) for i in range(n)]
    N = [None] + [[0] + [int(v) for v in nbr[:, i]] for i in range(n)]
    z_min, z_max, x_min, x_max, y_min, y_max = bounds
    lines = pyref.calc_delaunay_lines(P, N, n, x_min, x_max, y_min, y_max)
    rng = np.random.default_rng(3)
    S = 1 + rng.random(n)
    al = 10 ** rng.uniform(-3, 3, n) / (x_max - x_min) * 5
    for up, wall, layers_c, perm_c in ((True, -5, s.layers_up, s.perm_up),
                                       (False, -6, s.layers_down, s.perm_down)):
        lay = pyref.sort_by_layer(N, n, wall)
        perm = pyref.sortperm(lay, n)
        red = pyref.reduce_layers([0] + [lay[i] for i in perm[1:]])
        assert perm[1:] == perm_c.tolist()
        assert red[1:] == layers_c.tolist()
        I0 = rng.random(red[2] - 1)
        for theta, phi in ((109.7, 193.6), (152.7, 315.5)) if up else ((70.3, 346.4), (27.3, 135.5)):
            k = orc.direction(theta, phi)
            I_py = pyref.delaunay(up, _one_based(k), _one_based(S), _one_based(I0), _one_based(al),
                                  P, N, lines, red, perm, 3)
            I_c = (orc.Delaunay_upII if up else orc.Delaunay_downII)(k, S, I0, al, s, 3)
            assert np.allclose(I_c, np.array(I_py[1:]), rtol=1e-13, atol=1e-300)


def test_kat_optically_thick_gives_source_function(bcc_small):
    """uniform S = B and Δτ > 50 everywhere: e = 0, so I = a·S + b·S = S at every visited site,
    whatever the upwind intensity is."""
    pos, nbr, bounds = bcc_small
    s = orc.make_sites(pos, nbr, bounds)
    n = s.n
    k = orc.direction(147.2, 135.7)
    I = orc.Delaunay_upII(k, np.full(n, 3.5), np.full(s.layers_up[1] - 1, 3.5), np.full(n, 1.0), s, 3)
    skipped = s.perm_up[-1] - 1
    assert np.allclose(np.delete(I, skipped), 3.5, rtol=1e-14) and I[skipped] == 0.0


def test_later_layer_upwind_reads_zero(voro_small):
    """On a true Voronoi grid some upwind neighbours sit in a LATER layer and are read as 0
    (SURVEY appendix A.4), so the transparent solution dips below 1 but stays in [0, 1]."""
    pos, nbr, bounds = voro_small
    s = orc.make_sites(pos, nbr, bounds)
    n = s.n
    k = orc.direction(109.707418891553175, 193.587044948382584)
    I = orc.Delaunay_upII(k, np.zeros(n), np.ones(s.layers_up[1] - 1), np.zeros(n), s, 3)
    assert I.min() >= 0.0 and I.max() <= 1.0 + 1e-14
    assert (I < 1 - 1e-12).sum() > 1


def test_wrong_I0_length_is_an_error(bcc_small):
    pos, nbr, bounds = bcc_small
    s = orc.make_sites(pos, nbr, bounds)
    with pytest.raises(ValueError):
        orc.Delaunay_upII(orc.direction(150, 0), np.zeros(s.n), np.ones(3), np.zeros(s.n), s, 3)


def test_J_is_weighted_sum_of_solves(voro_small):
    pos, nbr, bounds = voro_small
    s = orc.make_sites(pos, nbr, bounds)
    n = s.n
    rng = np.random.default_rng(1)
    nlam = 2
    S = 1 + rng.random((n, nlam))
    al = 10 ** rng.uniform(-2, 2, (n, nlam))
    I0 = rng.random((s.layers_up[1] - 1, nlam))
    from voronoirt_amd.api import read_quadrature
    w, th, ph, _ = read_quadrature("ul7n12.dat")
    J = orc.J_voronoi(w, th, ph, S, al, s, I0_up=I0, nthreads=2)
    Jm = np.zeros_like(J)
    for a in range(12):
        k = orc.direction(th[a], ph[a])
        for l in range(nlam):
            if th[a] > 90:
                I = orc.Delaunay_upII(k, S[:, l], I0[:, l], al[:, l], s, 3)
            else:
                I = orc.Delaunay_downII(k, S[:, l], np.zeros(s.layers_down[1] - 1), al[:, l], s, 3)
            Jm[:, l] += w[a] * I
    assert np.array_equal(J, Jm)


# ---- golden vectors ---------------------------------------------------------------------------
def test_oracle_matches_golden(golden):
    g = golden
    exp = g["exp"]
    n = g["meta"]["n"]
    s = orc.read_cell(g["nbr_file"], n, g["pos"], g["bounds"])
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(s, key), exp[key]), key
    from voronoirt_amd.api import read_quadrature
    w, th, ph, _ = read_quadrature(g["meta"]["quadrature"])
    S, al = exp["S"], exp["alpha"]
    for a in g["meta"]["angles"]:
        k = orc.direction(th[a], ph[a])
        up, dots, wt, r, st = orc.upwind_table(s, k)
        assert np.array_equal(up, exp[f"up_{a}"])
        if th[a] > 90:
            I = orc.Delaunay_upII(k, S[:, 0], exp["I0_up"][:, 0], al[:, 0], s, 3)
        else:
            I = orc.Delaunay_downII(k, S[:, 0], exp["I0_down"][:, 0], al[:, 0], s, 3)
        assert np.allclose(I, exp[f"I_{a}"], rtol=1e-13, atol=0)
    J = orc.J_voronoi(w, th, ph, S, al, s, I0_up=exp["I0_up"], I0_down=exp["I0_down"])
    assert np.allclose(J, exp["J"], rtol=1e-13, atol=0)
