"""CPU tests (-m "not gpu") of the product's host side: the C-ABI library loads and exports every
symbol include/voronoirt.h declares, the host grid preparation (layers, stable permutation,
reduced offsets, neighbour-file parsing) equals the oracle's literal restatement, the
dependency schedule reproduces the serial Gauss-Seidel sweep exactly, error behaviour, and the
multi-rank sharding logic under gloo.  No compute call is made: there is no GPU here and the
library has no CPU fallback (that is asserted too)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import voronoirt_amd as vrt
from oracle import oracle as orc
from voronoirt_amd import _lib, distributed, synth
from voronoirt_amd.api import build_patch_schedule, build_layer_schedule, build_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "voronoirt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vrt_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in voronoirt.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.load().vrt_version() >= 100


def test_default_patch_kernels_hold_their_register_budget(tmp_path):
    """The default sweep kernel owes its speed to FOUR 512-thread workgroups per CU: k_patch_lean must fit 64
    VGPRs without scratch (8 bytes of scratch cost 3.5 %, 56 bytes 80 %: DESIGN.md section 5), the 72-register
    kernel and k_patch_quad at most 80 (three workgroups).  Read from hipcc's own resource report."""
    import subprocess
    from voronoirt_amd import build
    src = os.path.join(ROOT, "voronoirt_amd", "csrc", "vrt_patch.hip")
    cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "voronoirt_amd", "csrc"), "-c", src,
           "-o", str(tmp_path / "p.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    usage = {}
    name = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                usage[name][key] = int(m.group(1))
    lean = {k: v for k, v in usage.items() if "k_patch_lean" in k and "Li512E" in k}
    assert len(lean) == 6                                   # double / float x three alpha modes
    for k, v in lean.items():
        assert v["VGPRs"] <= 64 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
    plain = {k: v for k, v in usage.items() if "k_patch_solveI" in k and "Li1ELi1ELi512E" in k}
    assert len(plain) == 6
    for k, v in plain.items():
        assert v["VGPRs"] <= 80 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
    quad = {k: v for k, v in usage.items() if "k_patch_quad" in k and "Li512E" in k}
    assert len(quad) == 3
    for k, v in quad.items():
        # (12 bytes used: the thread index parked across the kernel's two roles, one word reloaded at the tail of the pair loop --
        # once per four wavelengths; nothing inside the gather / weights / level phases)
        assert v["VGPRs"] <= 80 and v["ScratchSize [bytes/lane]"] <= 16, (k, v)
    # the chained launch of the default (fp64) storage: the same budget as k_patch_lean, at eight waves per SIMD
    # (a ninth scalar register over 80 would be spilled into a vector register's lanes, and the solver has none to spare)
    chain = {k: v for k, v in usage.items() if "k_patch_chainId" in k and "Li512E" in k}
    assert len(chain) == 3
    for k, v in chain.items():
        assert v["VGPRs"] <= 64 and v["ScratchSize [bytes/lane]"] == 0 and v["Occupancy [waves/SIMD]"] == 8, (k, v)
    # its data-as-flag form (one or two wavelength pairs: the chip is far from full) trades two waves per SIMD for no scratch
    df = {k: v for k, v in usage.items() if "k_patch_chain_dfId" in k and "Li512E" in k}
    assert len(df) == 3
    for k, v in df.items():
        assert v["VGPRs"] <= 80 and v["ScratchSize [bytes/lane]"] == 0, (k, v)


def test_no_cpu_fallback(bcc_small):
    """Without a HIP device every compute entry point fails loudly."""
    L = _lib.load()
    if L.vrt_device_count() > 0:
        pytest.skip("a GPU is present")
    pos, nbr, bounds = bcc_small
    with pytest.raises(vrt.VrtError) as e:
        vrt.VoronoiSites(pos, nbr, bounds, device=0)
    assert e.value.code == _lib.VRT_ENODEVICE
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)       # host-only handle
    with pytest.raises(vrt.VrtError) as e:
        vrt.FormalPlan(hs, [[-1.0, 0.0, 0.0]])
    assert e.value.code == _lib.VRT_ENODEVICE
    with pytest.raises(vrt.VrtError) as e:
        vrt.Delaunay_upII([-1.0, 0, 0], np.zeros(hs.n), np.zeros(hs.layers_up[1] - 1),
                          np.zeros(hs.n), hs)
    assert e.value.code == _lib.VRT_ENODEVICE
    with pytest.raises(vrt.VrtError):
        hs.Delaunay_lines


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under voronoirt_amd/ may reference it."""
    pkg = os.path.join(ROOT, "voronoirt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libvrt_oracle" not in txt and "orc_" not in txt, f


# ---- grid preparation == oracle -----------------------------------------------------------
def _check_grid(hs, so):
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), getattr(so, key)), key
    assert hs.max_neighbours == so.D


def test_layers_perm_match_oracle_bcc(bcc_small):
    pos, nbr, bounds = bcc_small
    _check_grid(vrt.VoronoiSites(pos, nbr, bounds, device=-1), orc.make_sites(pos, nbr, bounds))


def test_layers_perm_match_oracle_voronoi(voro_small):
    pos, nbr, bounds = voro_small
    _check_grid(vrt.VoronoiSites(pos, nbr, bounds, device=-1), orc.make_sites(pos, nbr, bounds))


def test_layers_match_oracle_asymmetric_lists(voro_small):
    """The reference layers a cell by ITS OWN list (voronoi_utils.jl:113-114); drop one direction
    of some edges and the product's transposed-graph BFS must still agree with the literal scan."""
    pos, nbr, bounds = voro_small
    nbr = nbr.copy()
    rng = np.random.default_rng(0)
    n = nbr.shape[1]
    for i in rng.choice(n, 150, replace=False):
        c = nbr[0, i]
        ids = nbr[1:c + 1, i]
        keep = np.ones(c, dtype=bool)
        cand = np.nonzero(ids > 0)[0]
        if cand.size > 4:
            keep[rng.choice(cand)] = False
        new = ids[keep]
        nbr[1:, i] = 0
        nbr[1:new.size + 1, i] = new
        nbr[0, i] = new.size
    _check_grid(vrt.VoronoiSites(pos, nbr, bounds, device=-1), orc.make_sites(pos, nbr, bounds))


def test_read_cell_from_file_matches_oracle(golden):
    g = golden
    n = g["meta"]["n"]
    hs = vrt.read_cell(g["nbr_file"], n, g["pos"], g["bounds"], device=-1)
    exp = g["exp"]
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(hs, key), exp[key]), key
    so = orc.read_cell(g["nbr_file"], n, g["pos"], g["bounds"])
    assert hs.max_neighbours == so.D


def test_single_layer_grid():
    """Every cell touches the bottom wall: one layer, reduced offsets [1, n] (reduce_layers with
    max = 1), n1 = n - 1 and no sweep at all."""
    pos, nbr, bounds = synth.regular_lattice_grid(3, 3, 1)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    so = orc.make_sites(pos, nbr, bounds)
    assert hs.layers_up.tolist() == [1, 9] == so.layers_up.tolist()


def test_grid_errors(bcc_small, tmp_path):
    pos, nbr, bounds = bcc_small
    n = pos.shape[0]
    bad = nbr.copy()
    bad[1, 0] = n + 5                                   # id out of range
    with pytest.raises(vrt.VrtError) as e:
        vrt.VoronoiSites(pos, bad, bounds, device=-1)
    assert e.value.code == _lib.VRT_EGRID
    bad = nbr.copy()
    bad[bad == synth.TOP_WALL] = -1                     # no cell touches the top wall
    with pytest.raises(vrt.VrtError) as e:
        vrt.VoronoiSites(pos, bad, bounds, device=-1)
    assert e.value.code == _lib.VRT_EGRID and "not connected" in e.value.message
    bad = nbr.copy()
    bad[0, 3] = 99                                      # count beyond the matrix
    with pytest.raises(vrt.VrtError):
        vrt.VoronoiSites(pos, bad, bounds, device=-1)
    with pytest.raises(vrt.VrtError) as e:
        vrt.read_cell(str(tmp_path / "missing.txt"), n, pos, bounds, device=-1)
    assert e.value.code == _lib.VRT_EIO
    f = tmp_path / "garbage.txt"
    f.write_text("1 2 x 3\n")
    with pytest.raises(vrt.VrtError) as e:
        vrt.read_cell(str(f), n, pos, bounds, device=-1)
    assert e.value.code == _lib.VRT_EIO
    with pytest.raises(ValueError):
        vrt.VoronoiSites(pos[:, :2], nbr, bounds, device=-1)


def test_read_quadrature():
    w, th, ph, n = vrt.read_quadrature("ul7n12.dat")
    assert n == 12 and abs(w.sum() - 1.0) < 1e-12
    assert (th > 90).sum() == 6 and (th < 90).sum() == 6
    w, th, ph, n = vrt.read_quadrature(os.path.join(vrt.QUADRATURE_DIR, "ul9n20.dat"))
    assert n == 20 and abs(w.sum() - 1.0) < 1e-12
    w, th, ph, n = vrt.read_quadrature("n1.dat")
    assert (w.tolist(), th.tolist(), ph.tolist()) == ([1.0], [180.0], [0.0])
    assert np.array_equal(vrt.direction(180.0, 0.0), orc.direction(180.0, 0.0))


# ---- the schedule reproduces the serial sweep ----------------------------------------------------
def _lw(dt):
    safe = np.where(dt == 0, 1.0, dt)
    e = np.where(dt < 5e-4, 1 - dt + 0.5 * (dt * dt), np.where(dt > 50, 0.0, np.exp(-dt)))
    a = np.where(dt < 5e-4, dt * (0.5 - dt / 3), np.where(dt > 50, 1 / safe, (1 - e) / safe - e))
    b = np.where(dt < 5e-4, dt * (0.5 - dt / 6), np.where(dt > 50, 1 - a, 1 - a - e))
    return a, b, e


def _run_schedule(so, hs, k, S, I0, alpha, dirn, n_sweeps):
    """Executes the product's level schedule with numpy: within a level all nodes are evaluated
    from the state before the level (they must be independent), levels in order."""
    up, dots, w, r, st = orc.upwind_table(so, k)
    site, z, off = build_schedule(hs, dirn, up, n_sweeps)
    perm = so.perm_up if dirn > 0 else so.perm_down
    lay = so.layers_up if dirn > 0 else so.layers_down
    I = np.full(so.n, np.nan)                    # NaN = never written: reading it is a bug
    I[perm[: lay[1] - 1] - 1] = I0
    I[perm[-1] - 1] = 0.0
    for t in range(len(off) - 1):
        s = site[off[t]:off[t + 1]] - 1
        zz = z[off[t]:off[t + 1]]
        assert np.unique(s).size == s.size       # one write per site per level
        u1, u2 = up[s, 0] - 1, up[s, 1] - 1
        I1 = np.where(zz & 1, 0.0, I[u1])
        I2 = np.where(zz & 2, 0.0, I[u2])
        assert not np.isnan(I1).any() and not np.isnan(I2).any()
        # no node of this level may read a site another node of this level writes
        written = np.zeros(so.n, dtype=bool)
        written[s] = True
        assert not (written[u1] & ~(zz & 1).astype(bool)).any()
        assert not (written[u2] & ~(zz & 2).astype(bool)).any()
        a1, b1, e1 = _lw(r[s, 0] * (alpha[s] + alpha[u1]) / 2)
        a2, b2, e2 = _lw(r[s, 1] * (alpha[s] + alpha[u2]) / 2)
        t1 = ((e1 * I1 + a1 * S[u1]) + b1 * S[s]) * w[s, 0]
        t2 = ((e2 * I2 + a2 * S[u2]) + b2 * S[s]) * w[s, 1]
        I[s] = (0.0 + t1) + t2
    assert not np.isnan(I).any()
    return I, site.size, len(off) - 1


@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
@pytest.mark.parametrize("n_sweeps", [1, 2, 3, 5])
def test_schedule_equals_serial_gauss_seidel(grid, n_sweeps, bcc_small, voro_small):
    pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    n = so.n
    rng = np.random.default_rng(7)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-3, 3, n) / (bounds[3] - bounds[2]) * 10
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for t, p in list(zip(th, ph))[:: 1 if n_sweeps == 3 else 4]:
        k = orc.direction(t, p)
        dirn = 1 if t > 90 else -1
        lay = so.layers_up if dirn > 0 else so.layers_down
        I0 = rng.random(lay[1] - 1)
        ref = (orc.Delaunay_upII if dirn > 0 else orc.Delaunay_downII)(k, S, I0, alpha, so, n_sweeps)
        got, n_nodes, n_levels = _run_schedule(so, hs, k, S, I0, alpha, dirn, n_sweeps)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-300)
        assert n_nodes <= n_sweeps * n          # never more visits than the reference makes


def test_schedule_lattice_ordered_ids_deep_chains():
    """Sites numbered in lattice order give long in-sweep dependency chains (worst case for the
    level count); the schedule must still be exact."""
    pos, nbr, bounds = synth.bcc_grid(6, 6, seed=3, permute_ids=False)
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    rng = np.random.default_rng(1)
    S = 1 + rng.random(so.n)
    alpha = 10 ** rng.uniform(-3, 2, so.n) / 6e6 * 10
    for t, p, dirn in ((109.7, 193.6, 1), (70.3, 346.4, -1)):
        k = orc.direction(t, p)
        lay = so.layers_up if dirn > 0 else so.layers_down
        I0 = rng.random(lay[1] - 1)
        ref = (orc.Delaunay_upII if dirn > 0 else orc.Delaunay_downII)(k, S, I0, alpha, so, 3)
        got, _, _ = _run_schedule(so, hs, k, S, I0, alpha, dirn, 3)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-300)


def _run_layer_tiles(so, hs, k, S, I0, alpha, dirn, n_sweeps):
    """numpy model of k_sweep_tiles: data in sweep order, per layer the coefficients (c, g1, g2)
    are formed once, then the layer's levels update a zero-initialised tile."""
    up, dots, w, r, st = orc.upwind_table(so, k)
    vis, nlev, nv = build_layer_schedule(hs, dirn, up, n_sweeps)
    perm = so.perm_up if dirn > 0 else so.perm_down
    lay = so.layers_up if dirn > 0 else so.layers_down
    n = so.n
    rank = np.empty(n, dtype=np.int64)
    rank[perm - 1] = np.arange(n)
    I = np.full(n, np.nan)
    I[: lay[1] - 1] = I0
    for layer in range(2, len(lay)):
        lo, hi = lay[layer - 1] - 1, lay[layer] - 1
        sites = perm[lo:hi] - 1
        s1, s2 = up[sites, 0] - 1, up[sites, 1] - 1
        u1, u2 = rank[s1], rank[s2]
        a1, b1, e1 = _lw(r[sites, 0] * (alpha[sites] + alpha[s1]) / 2)
        a2, b2, e2 = _lw(r[sites, 1] * (alpha[sites] + alpha[s2]) / 2)
        w1, w2 = w[sites, 0], w[sites, 1]
        early1, in1 = u1 < lo, (u1 >= lo) & (u1 < hi)
        early2, in2 = u2 < lo, (u2 >= lo) & (u2 < hi)
        I1 = np.where(early1, I[np.minimum(u1, max(lo - 1, 0))], 0.0)
        I2 = np.where(early2, I[np.minimum(u2, max(lo - 1, 0))], 0.0)
        assert not np.isnan(I1).any() and not np.isnan(I2).any()
        t1 = np.where(early1, ((e1 * I1 + a1 * S[s1]) + b1 * S[sites]) * w1, (a1 * S[s1] + b1 * S[sites]) * w1)
        t2 = np.where(early2, ((e2 * I2 + a2 * S[s2]) + b2 * S[sites]) * w2, (a2 * S[s2] + b2 * S[sites]) * w2)
        c = t1 + t2
        g1, g2 = np.where(in1, e1 * w1, 0.0), np.where(in2, e2 * w2, 0.0)
        loc1, loc2 = np.where(in1, u1 - lo, 0), np.where(in2, u2 - lo, 0)
        tile = np.zeros(hi - lo)
        v = vis[sites]
        for t in range(1, nlev[layer] + 1):
            hit = ((v & 0xFF) == t) | (((v >> 8) & 0xFF) == t) | (((v >> 16) & 0xFF) == t) | ((v >> 24) == t)
            # nodes of one level must not read a slot another node of the level writes
            written = np.zeros(hi - lo, dtype=bool)
            written[hit] = True
            assert not (written[loc1] & in1 & hit).any() and not (written[loc2] & in2 & hit).any()
            tile = np.where(hit, c + g1 * tile[loc1] + g2 * tile[loc2], tile)
        I[lo:hi] = tile
    I[n - 1] = 0.0
    out = np.empty(n)
    out[perm - 1] = I
    return out, nv


@pytest.mark.parametrize("grid", ["bcc", "voronoi", "lattice"])
def test_layer_tile_schedule_equals_serial_gauss_seidel(grid, bcc_small, voro_small):
    if grid == "lattice":
        pos, nbr, bounds = synth.bcc_grid(6, 6, seed=3, permute_ids=False)
    else:
        pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    n = so.n
    rng = np.random.default_rng(17)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-3, 3, n) / (bounds[3] - bounds[2]) * 10
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for n_sweeps in (1, 2, 3, 4):
        for t, p in list(zip(th, ph))[:: 1 if n_sweeps == 3 else 5]:
            k = orc.direction(t, p)
            dirn = 1 if t > 90 else -1
            lay = so.layers_up if dirn > 0 else so.layers_down
            I0 = rng.random(lay[1] - 1)
            ref = (orc.Delaunay_upII if dirn > 0 else orc.Delaunay_downII)(k, S, I0, alpha, so, n_sweeps)
            got, nv = _run_layer_tiles(so, hs, k, S, I0, alpha, dirn, n_sweeps)
            assert np.allclose(got, ref, rtol=1e-10, atol=1e-300)
            assert nv <= n_sweeps * n
    with pytest.raises(vrt.VrtError):     # 5 sweeps can need 5 visits: beyond the 4-slot encoding
        up, *_ = orc.upwind_table(so, orc.direction(109.7, 193.6))
        build_layer_schedule(hs, +1, up, 9)


def _run_patches(so, hs, k, S, I0, alpha, dirn, n_sweeps, own, cap):
    """numpy model of k_patch_solve (vrt_patch.hip): data in STORAGE order; every patch forms the
    coefficients of its entries (own sites + halo), runs its levels on a private zero-initialised tile
    and writes back the sites it owns.  Patches of a layer never see each other's results."""
    up, dots, w, r, st = orc.upwind_table(so, k)
    ps = build_patch_schedule(hs, dirn, up, n_sweeps, own, cap)
    store = hs.storage_order(dirn) - 1
    perm = so.perm_up if dirn > 0 else so.perm_down
    lay = so.layers_up if dirn > 0 else so.layers_down
    n = so.n
    srank = np.empty(n, dtype=np.int64)
    srank[store] = np.arange(n)
    I = np.full(n, np.nan)
    I[srank[perm[: lay[1] - 1] - 1]] = I0
    lpo, eo = ps["layer_patch_off"], ps["patch_ent_off"]
    assert ps["max_entries"] <= cap
    covered = np.zeros(n, dtype=np.int32)
    for layer in range(2, len(lay)):
        lo, hi = lay[layer - 1] - 1, lay[layer] - 1
        Inew = I.copy()
        for q in range(lpo[layer], lpo[layer + 1]):
            e0, e1 = eo[q], eo[q + 1]
            ne = e1 - e0
            pos = ps["entry_pos"][e0:e1].astype(np.int64)
            olo, ocnt = ps["patch_own_lo"][q], ps["patch_own_cnt"][q]
            assert np.array_equal(pos[:ocnt], np.arange(olo, olo + ocnt)) and lo <= olo and olo + ocnt <= hi
            assert ((pos[ocnt:] >= lo) & (pos[ocnt:] < hi)).all() and np.unique(pos).size == ne
            covered[olo:olo + ocnt] += 1
            sites = store[pos]
            s1, s2 = up[sites, 0] - 1, up[sites, 1] - 1
            u1, u2 = srank[s1], srank[s2]
            a1, b1, e1_ = _lw(r[sites, 0] * (alpha[sites] + alpha[s1]) / 2)
            a2, b2, e2_ = _lw(r[sites, 1] * (alpha[sites] + alpha[s2]) / 2)
            w1, w2 = w[sites, 0], w[sites, 1]
            early1, in1 = u1 < lo, (u1 >= lo) & (u1 < hi)
            early2, in2 = u2 < lo, (u2 >= lo) & (u2 < hi)
            I1 = np.where(early1, I[np.minimum(u1, max(lo - 1, 0))], 0.0)     # the layer BEFORE this launch
            I2 = np.where(early2, I[np.minimum(u2, max(lo - 1, 0))], 0.0)
            assert not np.isnan(I1).any() and not np.isnan(I2).any()
            t1 = np.where(early1, ((e1_ * I1 + a1 * S[s1]) + b1 * S[sites]) * w1, (a1 * S[s1] + b1 * S[sites]) * w1)
            t2 = np.where(early2, ((e2_ * I2 + a2 * S[s2]) + b2 * S[sites]) * w2, (a2 * S[s2] + b2 * S[sites]) * w2)
            c = t1 + t2
            g1, g2 = np.where(in1, e1_ * w1, 0.0), np.where(in2, e2_ * w2, 0.0)
            loc = ps["entry_loc"][e0:e1]
            l1, l2 = (loc & 0xFFFF).astype(np.int64), (loc >> 16).astype(np.int64)
            # a local slot names the upwind's entry; 0xFFFF only for upwinds outside the layer or the cone
            has1, has2 = l1 != 0xFFFF, l2 != 0xFFFF
            assert np.array_equal(pos[l1[has1]], u1[has1]) and np.array_equal(pos[l2[has2]], u2[has2])
            assert in1[has1].all() and in2[has2].all()
            l1, l2 = np.where(has1, l1, ne), np.where(has2, l2, ne)
            tile = np.zeros(ne + 1)
            v = ps["entry_vis"][e0:e1]
            assert (v[:ocnt] != 0).all()
            for t in range(1, ps["patch_nlev"][q] + 1):
                hit = ((v & 0xFF) == t) | (((v >> 8) & 0xFF) == t) | (((v >> 16) & 0xFF) == t) | ((v >> 24) == t)
                written = np.zeros(ne + 1, dtype=bool)
                written[:ne][hit] = True
                assert not (written[l1] & hit).any() and not (written[l2] & hit).any()
                tile[:ne] = np.where(hit, c + g1 * tile[l1] + g2 * tile[l2], tile[:ne])
            Inew[olo:olo + ocnt] = tile[:ocnt]
        I = Inew
    assert (covered[lay[1] - 1:n - 1] == 1).all() and (covered[:lay[1] - 1] == 0).all()
    I[n - 1] = 0.0
    out = np.empty(n)
    out[store] = I
    return out, ps


@pytest.mark.parametrize("grid", ["bcc", "voronoi", "lattice"])
def test_patch_schedule_equals_serial_gauss_seidel(grid, bcc_small, voro_small):
    """The fused patch path: layers cut into patches of a few dozen sites, each with the in-layer
    dependency cone of its sites as a halo; executing the patches independently reproduces the
    reference's serial sweep -- and the unsplit layer schedule bit for bit."""
    if grid == "lattice":
        pos, nbr, bounds = synth.bcc_grid(6, 6, seed=3, permute_ids=False)
    else:
        pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    n = so.n
    rng = np.random.default_rng(23)
    S = 1 + rng.random(n)
    alpha = 10 ** rng.uniform(-3, 3, n) / (bounds[3] - bounds[2]) * 10
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    halo = 0
    for n_sweeps, own, cap in ((3, 24, 128), (3, 7, 128), (1, 40, 48), (2, 16, 4096), (4, 120, 160), (3, 60000, 65535)):
        for t, p in list(zip(th, ph))[:: 1 if (n_sweeps, own) == (3, 24) else 5]:
            k = orc.direction(t, p)
            dirn = 1 if t > 90 else -1
            lay = so.layers_up if dirn > 0 else so.layers_down
            I0 = rng.random(lay[1] - 1)
            ref = (orc.Delaunay_upII if dirn > 0 else orc.Delaunay_downII)(k, S, I0, alpha, so, n_sweeps)
            got, ps = _run_patches(so, hs, k, S, I0, alpha, dirn, n_sweeps, own, min(cap, 65535))
            assert np.allclose(got, ref, rtol=1e-10, atol=1e-300)
            unsplit, nv = _run_layer_tiles(so, hs, k, S, I0, alpha, dirn, n_sweeps)
            assert np.array_equal(got, unsplit)                 # same arithmetic on the same values
            assert ps["live_visits"] == nv and ps["visits"] >= nv
            # the builder's by-product IS the angle's layer schedule (plan creation takes it from there)
            up_tab, *_ = orc.upwind_table(so, k)
            vis, nlev, nvis = build_layer_schedule(hs, dirn, up_tab, n_sweeps)
            assert np.array_equal(ps["layer_vis"], vis) and np.array_equal(ps["layer_nlev"], nlev) and ps["layer_visits"] == nvis
            if own >= 60000:                                    # one patch per layer: no halo at all
                assert ps["visits"] == nv and ps["entries"] == n - (lay[1] - 1) - 1
            halo += ps["entries"] - (n - (lay[1] - 1) - 1)
    assert halo > 0                                              # the small patches did need halos
    with pytest.raises(vrt.VrtError):     # the cone of a single site does not fit 4 entries
        up, *_ = orc.upwind_table(so, orc.direction(109.7, 193.6))
        build_patch_schedule(hs, +1, up, 3, 2, 4)


@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
def test_patch_dependency_lists_cover_every_gathered_intensity(grid, bcc_small, voro_small):
    """The chained launch orders patches by data only: patch q waits for dep_list[dep_off[q] .. dep_off[q+1]).
    Those lists must name the owner of EVERY earlier-layer intensity q gathers (irregular_ray_tracing.jl:75) and
    nothing of q's own or a later layer: then any order that respects them reads final values only."""
    pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    n = so.n
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    rng = np.random.default_rng(7)
    for own, cap in ((24, 128), (7, 128), (120, 512)):
        for t, p in list(zip(th, ph))[::3]:
            dirn = 1 if t > 90 else -1
            up, *_ = orc.upwind_table(so, orc.direction(t, p))
            ps = build_patch_schedule(hs, dirn, up, 3, own, cap)
            store = hs.storage_order(dirn) - 1
            lay = so.layers_up if dirn > 0 else so.layers_down
            srank = np.empty(n, dtype=np.int64)
            srank[store] = np.arange(n)
            P = ps["patches"]
            owner = np.full(n, -1, dtype=np.int64)              # storage position -> patch that stores it
            layer_of_patch = np.zeros(P, dtype=np.int64)
            for layer in range(2, len(lay)):
                for q in range(ps["layer_patch_off"][layer], ps["layer_patch_off"][layer + 1]):
                    owner[ps["patch_own_lo"][q]: ps["patch_own_lo"][q] + ps["patch_own_cnt"][q]] = q
                    layer_of_patch[q] = layer
            assert (owner[lay[1] - 1: n - 1] >= 0).all() and (owner[: lay[1] - 1] == -1).all() and owner[n - 1] == -1
            doff, dl = ps["dep_off"], ps["dep_list"]
            assert doff[0] == 0 and doff[-1] == dl.size and (np.diff(doff) >= 0).all()
            for q in range(P):
                lo = lay[layer_of_patch[q] - 1] - 1
                e0, e1 = ps["patch_ent_off"][q], ps["patch_ent_off"][q + 1]
                sites = store[ps["entry_pos"][e0:e1].astype(np.int64)]
                u = srank[np.concatenate([up[sites, 0], up[sites, 1]]) - 1]
                need = np.unique(owner[u[u < lo]])
                need = need[need >= 0]
                got = dl[doff[q]: doff[q + 1]]
                assert np.array_equal(got, need), (q, got, need)
                assert (layer_of_patch[got] < layer_of_patch[q]).all()
            # an execution order that respects nothing but the lists: the patches in a random order, each deferred
            # until its dependencies have run -- never reads an intensity that is not final
            done = np.zeros(P, dtype=bool)
            final = np.zeros(n, dtype=bool)
            final[: lay[1] - 1] = True                          # the boundary layer
            final[n - 1] = True                                 # the never-visited site (I = 0)
            pending = list(rng.permutation(P))
            while pending:
                rest = []
                for q in pending:
                    if done[dl[doff[q]: doff[q + 1]]].all():
                        lo = lay[layer_of_patch[q] - 1] - 1
                        e0, e1 = ps["patch_ent_off"][q], ps["patch_ent_off"][q + 1]
                        sites = store[ps["entry_pos"][e0:e1].astype(np.int64)]
                        u = srank[np.concatenate([up[sites, 0], up[sites, 1]]) - 1]
                        assert final[u[u < lo]].all()
                        final[ps["patch_own_lo"][q]: ps["patch_own_lo"][q] + ps["patch_own_cnt"][q]] = True
                        done[q] = True
                    else:
                        rest.append(q)
                assert len(rest) < len(pending)                 # no cycle
                pending = rest
            assert final.all()


def test_schedule_rejects_site_without_upwind(bcc_small):
    pos, nbr, bounds = bcc_small
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    so = orc.make_sites(pos, nbr, bounds)
    up, *_ = orc.upwind_table(so, orc.direction(150.0, 10.0))
    up = up.copy()
    victim = so.perm_up[so.layers_up[2]] - 1      # a site of layer 3
    up[victim] = 0
    with pytest.raises(vrt.VrtError) as e:
        build_schedule(hs, +1, up, 3)
    assert e.value.code == _lib.VRT_EGRID


# ---- synthetic grids -----------------------------------------------------------------------------
def _symmetric(nbr):
    n = nbr.shape[1]
    pairs = set()
    for i in range(n):
        for v in nbr[1:nbr[0, i] + 1, i]:
            if v > 0:
                pairs.add((i + 1, int(v)))
    return all((b, a) in pairs for a, b in pairs)


def test_synthetic_grids_are_valid(bcc_small, voro_small):
    for pos, nbr, bounds in (bcc_small, voro_small):
        n = pos.shape[0]
        assert _symmetric(nbr)
        assert (nbr[1:] <= n).all() and (nbr[0] >= 4).all()
        z_min, z_max, x_min, x_max, y_min, y_max = bounds
        assert (pos[:, 0] > z_min).all() and (pos[:, 0] < z_max).all()
        assert (pos[:, 1] >= x_min).all() and (pos[:, 1] <= x_max).all()
    pos, nbr, _ = bcc_small
    assert pos.shape[0] == 2 * 8 * 8 * 12 and (nbr[0] >= 10).all() and nbr[0].max() == 14
    assert synth.bcc_grid(8, 12, seed=2)[0].tobytes() == pos.tobytes()      # seeded
    pos, nbr, _ = voro_small
    assert 14.0 < nbr[0].mean() < 17.0            # Poisson-Voronoi: 15.54 faces on average


def test_counter_rng_is_reproducible():
    idx = np.arange(1000, dtype=np.uint64)
    u = synth.counter_uniform(5, 1, idx)
    assert np.array_equal(u, synth.counter_uniform(5, 1, idx))
    assert not np.array_equal(u, synth.counter_uniform(6, 1, idx))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.05
    pos, _, bounds = synth.bcc_grid(4, 4, seed=1)
    S, al = synth.synthetic_fields(pos, bounds, 7, seed=3)
    assert S.shape == (pos.shape[0], 7) and (S > 0).all() and (al > 0).all()
    S2, al3 = synth.synthetic_fields(pos, bounds, 7, seed=3, n_angles=3)
    assert np.array_equal(S, S2) and al3.shape == (3, pos.shape[0], 7)


# ---- sharding across ranks ---------------------------------------------------------------------------
def test_partition_covers_all_units():
    for n_units in (1, 7, 12, 20, 51, 100):
        for world in (1, 2, 3, 4, 8):
            blocks = [distributed.partition(n_units, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n_units
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in (distributed.partition(51, 8, r) for r in range(8))] == [7, 7, 7, 6, 6, 6, 6, 6]


def test_angle_assignment_balanced():
    w, th, ph, _ = vrt.read_quadrature("ul7n12.dat")
    for world in (1, 2, 4, 8):
        parts = distributed.angle_assignment(th, world)
        allidx = np.sort(np.concatenate(parts))
        assert allidx.tolist() == list(range(12))
        sizes = [p.size for p in parts]
        assert max(sizes) - min(sizes) <= 1
    parts = distributed.angle_assignment([100.0, 90.0, 80.0], 2)      # θ = 90 is skipped
    assert sorted(np.concatenate(parts).tolist()) == [0, 2]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from voronoirt_amd import distributed
rank, world = distributed.init_process_group("gloo")
assert world == 2
n, nlam, n_angles = 500, 7, 12
rng = np.random.default_rng(0)
I = rng.random((n_angles, n, nlam))            # stand-in per-angle intensities (same on all ranks)
w = rng.random(n_angles)
J_full = np.tensordot(w, I, axes=1)
# "angle" mode: partial sums over the rank's angles, then the J all-reduce
theta = np.where(np.arange(n_angles) % 2 == 0, 120.0, 60.0)
mine = distributed.angle_assignment(theta, world)[rank]
Jp = torch.from_numpy(np.tensordot(w[mine], I[mine], axes=1).copy())
distributed.allreduce_J(Jp)
assert np.allclose(Jp.numpy(), J_full, rtol=1e-13)
# "lambda" mode: every rank owns a wavelength block of J; all-gather replicates it
a, b = distributed.partition(nlam, world, rank)
Jb = torch.from_numpy(J_full[:, a:b].copy())
Jg = distributed.allgather_J_lambda(Jb, nlam)
assert np.array_equal(Jg.numpy(), J_full)
# the same as ONE collective into a preallocated (world, n, width) buffer (what bench.py times)
buf, sizes = distributed.allgather_J_blocks(Jb, nlam)
assert np.array_equal(distributed.assemble_J_blocks(buf, sizes).numpy(), J_full)
buf2, _ = distributed.allgather_J_blocks(Jb, nlam, out=buf)
assert buf2 is buf
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gloo_world2_reduce_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {r} ok" in out


def test_header_is_valid_c_and_c_caller_links(tmp_path):
    """include/voronoirt.h must be consumable from plain C (the Julia ccall / cgo / JNI side sees a C
    ABI), and the example C host links against the library."""
    hdr = os.path.join(ROOT, "include", "voronoirt.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    exe = tmp_path / "c_caller"
    libdir = os.path.join(ROOT, "voronoirt_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_caller.c"), "-o", str(exe),
                           "-L", libdir, "-lvrt_hip", f"-Wl,-rpath,{libdir}", "-lm"])
    assert exe.exists()


def test_sorted_thread_assignment_of_the_level_kernel(bcc_small, voro_small):
    """build_sorted_slots (k_step_levels' thread assignment): inside every layer `self` is a
    permutation of the layer's storage positions, stably sorted by visit pattern (first visit
    level, then second, ...), so a wave's 64 consecutive entries share their levels; the storage
    order itself keeps the layers contiguous and the never-visited site perm[n] last."""
    from voronoirt_amd.api import layer_sorted_slots
    for pos, nbr, bounds in (bcc_small, voro_small):
        so = orc.make_sites(pos, nbr, bounds)
        hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
        n = so.n
        for t, p, dirn in ((109.7, 193.6, 1), (70.3, 346.4, -1), (152.7, 45.0, 1)):
            up, _, _, _, _ = orc.upwind_table(so, orc.direction(t, p))
            vis, nlev, nv = build_layer_schedule(hs, dirn, up, 3)
            store, self_ = layer_sorted_slots(hs, dirn, vis)
            perm = so.perm_up if dirn > 0 else so.perm_down
            lay = so.layers_up if dirn > 0 else so.layers_down
            assert sorted(store.tolist()) == list(range(1, n + 1))
            assert store[n - 1] == perm[n - 1]
            groups_sorted = groups_plain = 0
            for layer in range(1, len(lay)):
                lo, hi = lay[layer - 1] - 1, lay[layer] - 1
                assert sorted(store[lo:hi].tolist()) == sorted(perm[lo:hi].tolist())   # same layer, other order
                s = self_[lo:hi]
                assert sorted(s.tolist()) == list(range(lo, hi))
                v = vis[store[s] - 1]
                key = [tuple(int((x >> (8 * j)) & 0xFF) for j in range(4)) for x in v]
                assert key == sorted(key)
                for a, b, ka, kb in zip(s[:-1], s[1:], key[:-1], key[1:]):
                    if ka == kb:
                        assert a < b                                                  # stable
                if layer >= 2:
                    plain = vis[store[lo:hi] - 1]
                    for arr, tot in ((v, "s"), (plain, "p")):
                        g = sum(len({int((x >> (8 * j)) & 0xFF) for x in arr[w:w + 64] for j in range(4)} - {0})
                                for w in range(0, hi - lo, 64))
                        if tot == "s":
                            groups_sorted += g
                        else:
                            groups_plain += g
            assert groups_sorted <= groups_plain          # fewer (wave, level) pairs to execute
        hs.close()


# ---- in-process tessellation ------------------------------------------------------
def _rows_as_sets(M):
    return [frozenset(M[1:M[0, i] + 1, i].tolist()) for i in range(M.shape[1])]


@pytest.mark.parametrize("case", ["uniform", "stratified", "bcc"])
def test_native_tessellation_matches_qhull_neighbour_sets(case):
    """vrt_tessellate (host C++, no GPU): the Voronoi neighbour SET of every cell -- walls -5 / -6
    included -- equals the one scipy/Qhull's Delaunay triangulation of the periodically extended,
    wall-mirrored point set gives (synth.voronoi_neighbours), and on a jittered BCC lattice the
    analytic 14 neighbours.  (The order inside a row is voro++'s secret; nothing can pin it.)"""
    if case == "uniform":
        rng = np.random.default_rng(11)
        n = 2500
        bounds = (0.0, 2.0, 0.0, 1.0, 0.0, 1.0)
        pos = np.stack([rng.uniform(0, 2, n), rng.uniform(0, 1, n), rng.uniform(0, 1, n)], axis=1)
        ref = synth.voronoi_neighbours(pos, bounds, margin=0.35, shuffle_seed=None)
    elif case == "stratified":
        pos, ref, bounds = synth.voronoi_grid(6000, seed=3, bounds=(-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6),
                                              scale_height=5.0e6, margin=0.45)
    else:
        # jittered BCC lattice: the 14 lattice neighbours of synth.bcc_grid are the exact Voronoi
        # neighbours away from the walls (its wall rows are a construction, not the tessellation's)
        pos, lattice, bounds = synth.bcc_grid(7, 9, seed=4)
        ref = synth.voronoi_neighbours(pos, bounds, margin=0.45, shuffle_seed=None)
        h = 6.0e6 / 7
        inner = (pos[:, 0] > bounds[0] + 1.5 * h) & (pos[:, 0] < bounds[1] - 1.5 * h)
        lat = _rows_as_sets(lattice)
    M = vrt.voro(pos, bounds)
    if case == "bcc":
        rows = _rows_as_sets(M)
        assert inner.sum() > 300 and all(rows[i] == lat[i] and len(rows[i]) == 14 for i in np.nonzero(inner)[0])
    assert M.shape[1] == pos.shape[0] and M.dtype == np.int64
    got, want = _rows_as_sets(M), _rows_as_sets(ref)
    wrong = [i for i in range(len(got)) if got[i] != want[i]]
    assert not wrong, (len(wrong), wrong[:5])
    assert M.shape[0] == ref.shape[0]                      # same D = maximum neighbour count
    # symmetric relation, walls only where the cell touches them
    for i in (0, len(got) // 2, len(got) - 1):
        for j in got[i]:
            if j > 0:
                assert (i + 1) in got[j - 1]


@pytest.mark.parametrize("n, box, rel_h, seed", [
    (300, (0.0, 1.0, 0.0, 3.0, 0.0, 0.5), None, 1003),        # anisotropic box only ~3 cells wide in y
    (300, (0.0, 1.0, 0.0, 3.0, 0.0, 0.5), 0.3, 1007),
    (1000, (0.0, 1.0, 0.0, 1.0, 0.0, 1.0), 0.15, 1008),       # top of the box almost empty: cells of half a period
    (3000, (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6), 0.15, 1002),
])
def test_native_tessellation_small_and_strongly_stratified(n, box, rel_h, seed):
    """Cells that reach across a good part of a period (few sites, or a density scale height of a
    seventh of the box): against Qhull on the point set extended by ALL eight periodic images
    (margin > 1), 0 differing rows."""
    H = None if rel_h is None else rel_h * (box[1] - box[0])
    pos, _, bounds = synth.voronoi_grid(n, seed=seed, bounds=box, scale_height=H, margin=0.05)
    ref = synth.voronoi_neighbours(pos, bounds, margin=1.01, shuffle_seed=None)
    got, want = _rows_as_sets(vrt.voro(pos, bounds)), _rows_as_sets(ref)
    wrong = [i for i in range(n) if got[i] != want[i]]
    assert not wrong, (len(wrong), wrong[:5])


def test_native_tessellation_feeds_read_cell(tmp_path):
    """voro -> "%i %n" text file -> read_cell: the file round trip reproduces the matrix, and the
    grid built from either has the same layers and permutations."""
    pos, _, bounds = synth.voronoi_grid(1500, seed=8, bounds=(0.0, 2.0, 0.0, 1.0, 0.0, 1.0), scale_height=0.7)
    f = str(tmp_path / "neighbours.txt")
    M = vrt.voro(pos, bounds, neighbours_file=f)
    back = orc.read_neighbours(f, pos.shape[0])
    assert np.array_equal(back, M)
    a = vrt.VoronoiSites(pos, M, bounds, device=-1)
    b = vrt.read_cell(f, pos.shape[0], pos, bounds, device=-1)
    so = orc.make_sites(pos, M, bounds)
    for key in ("layers_up", "layers_down", "perm_up", "perm_down"):
        assert np.array_equal(getattr(a, key), getattr(b, key)) and np.array_equal(getattr(a, key), getattr(so, key))


def test_native_tessellation_errors():
    L = _lib.load()
    rng = np.random.default_rng(0)
    pos = rng.random((40, 3))
    with pytest.raises(vrt.VrtError):                      # a site outside the box
        vrt.voro(pos + 2.0, (0, 1, 0, 1, 0, 1))
    M3 = vrt.voro(pos[:3], (0, 1, 0, 1, 0, 1))             # 3 sites in a periodic box: every cell reaches its own
    rows = _rows_as_sets(M3)                               # images across the period; those faces name no neighbour
    assert all((i + 1) not in rows[i] for i in range(3))
    assert all(j <= 0 or (i + 1) in rows[j - 1] for i in range(3) for j in rows[i])
    M = np.zeros((5, 40), dtype=np.int64)                  # a matrix too narrow for the rows
    mx = ctypes.c_int64()
    b = np.array([0, 1, 0, 1, 0, 1], dtype=np.float64)
    rc = L.vrt_tessellate(40, pos.ctypes.data_as(_lib.p_dbl), b.ctypes.data_as(_lib.p_dbl), 5,
                          M.ctypes.data_as(_lib.p_i64), ctypes.byref(mx))
    assert rc == _lib.VRT_EGRID


@pytest.mark.parametrize("order", [None, "strips:3", "strips:4096", "morton"])
@pytest.mark.parametrize("grid", ["bcc", "voronoi"])
def test_storage_order_keeps_the_layers_contiguous(grid, order, bcc_small, voro_small, monkeypatch):
    """The storage order permutes sites only INSIDE a layer: layer l occupies the same positions as in the sweep order
    (so the boundary layer is storage positions [0, n1) -- what k_chain_prepare relies on when it writes I_0 there by
    position, vrt_layout_kernels.h), and the never-visited last site of the order stays at position n - 1.  For the
    default order (strips of 20 lattice columns, rows inside, vrt_grid.cpp), other strip widths and the Morton curve of
    rounds 1-4 (VRT_STORE_ORDER, read at grid creation); in the strip order a row's sites are stored by increasing x."""
    pos, nbr, bounds = bcc_small if grid == "bcc" else voro_small
    so = orc.make_sites(pos, nbr, bounds)
    if order is None:
        monkeypatch.delenv("VRT_STORE_ORDER", raising=False)
    else:
        monkeypatch.setenv("VRT_STORE_ORDER", order)
    hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
    if order == "strips:4096":           # one strip: inside a layer the sites come row by row, x increasing along a row
        store = hs.storage_order(+1) - 1
        lo, hi = int(so.layers_up[1]) - 1, int(so.layers_up[2]) - 1      # the second layer
        x = pos[store[lo:hi], 1]
        a = np.sqrt(2.0 * (bounds[3] - bounds[2]) * (bounds[5] - bounds[4]) / (hi - lo))
        row = np.minimum(0x3FFF, ((pos[store[lo:hi], 2] - bounds[4]) / (0.5 * a) + 0.25).astype(np.int64))
        assert np.all(np.diff(row) >= 0)
        same = np.diff(row) == 0
        assert np.all(np.diff(x)[same] >= -1e-9 * (bounds[3] - bounds[2]))
    for dirn, perm, lay in ((+1, so.perm_up, so.layers_up), (-1, so.perm_down, so.layers_down)):
        store = hs.storage_order(dirn)
        assert sorted(store) == list(range(1, so.n + 1))
        assert store[-1] == perm[-1]
        edges = [0] + [int(x) - 1 for x in lay[1:]]
        for lo, hi in zip(edges[:-1], edges[1:]):
            hi = min(hi, so.n - 1)
            assert set(store[lo:hi]) == set(perm[lo:hi])


def test_rccl_constants_declared_by_hand_match_the_installed_header():
    """csrc/vrt_multi.cpp declares the few RCCL entry points and enum values it uses itself (librccl is dlopen'ed; the
    library builds without the RCCL headers).  Where the header is installed, its values are the ones declared."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no RCCL header on this machine")
    h = open(hdr).read()
    src = open(os.path.join(ROOT, "voronoirt_amd", "csrc", "vrt_multi.cpp")).read()
    mine = {k: int(v) for k, v in re.findall(r"\b(ncclSuccess|ncclDouble|ncclSum)\s*=\s*(\d+)", src)}
    assert set(mine) == {"ncclSuccess", "ncclDouble", "ncclSum"}
    for name, value in mine.items():
        m = re.search(r"\b" + name + r"\s*=\s*(\d+)", h)
        assert m and int(m.group(1)) == value, name
    # the prototypes: argument lists of the header, reduced to their types
    for fn, nargs in (("ncclCommInitAll", 3), ("ncclAllReduce", 7), ("ncclReduce", 8), ("ncclCommDestroy", 1)):
        m = re.search(r"ncclResult_t\s+" + fn + r"\s*\(([^)]*)\)", h)
        assert m and len([a for a in m.group(1).split(",") if a.strip()]) == nargs, fn


def test_host_sanitizer_screen_still_builds():
    """tools/asan_host.sh (ASan + UBSan and TSan builds of every host translation unit against generated stubs of the device
    side) must not go stale: its `check` mode -- the same build without a sanitizer -- has to link and load with every
    symbol resolved.  The sanitizer runs themselves are logged under profiles/."""
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh"), "check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "loads with every symbol resolved" in r.stdout


def test_gloo_world_of_one_runs_the_collectives(tmp_path):
    """distributed.init_process_group(force=True): a process group also for ONE rank, so that the collectives of the
    N > 1 modes execute on a single member (what bench.py's VRT_BENCH_FORCE_DIST=1 uses on a one-GPU box, there with
    the nccl backend); here with gloo on the CPU."""
    script = tmp_path / "w1.py"
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "import torch, torch.distributed as dist\n"
        "from voronoirt_amd import distributed as D\n"
        "rank, world = D.init_process_group('gloo', force=True)\n"
        "assert (rank, world) == (0, 1) and dist.is_initialized()\n"
        "J = torch.arange(12, dtype=torch.float64).reshape(4, 3)\n"
        "assert torch.equal(D.allreduce_J(J.clone()), J)\n"
        "buf, sizes = D.allgather_J_blocks(J, 3)\n"
        "assert sizes == [(0, 3)] and torch.equal(D.assemble_J_blocks(buf, sizes), J)\n"
        "dist.destroy_process_group()\n"
        "print('ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
