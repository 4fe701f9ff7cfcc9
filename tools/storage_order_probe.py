#!/usr/bin/env python3
"""diagnostics (host only): 128-byte lines a patch's gathers touch under candidate storage orders.

For every patch of the patch schedule (own sites + halo, as the kernel executes them) count the distinct
128-byte lines (8 positions of a 16-byte wavelength pair) touched by (a) its entries' own reads, (b) the gathers that
leave the layer (earlier / later layer upwinds), (c) all upwind gathers -- against the bytes actually used.  The patch
composition is that of the present schedule (cut along the Morton curve); only the POSITION of a site changes.
usage: python tools/storage_order_probe.py [C2|C4|strat:<n>] [quadrature]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
import voronoirt_amd.api  # noqa: E402,F401
from oracle import oracle as orc  # noqa: E402
from voronoirt_amd import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "C2"
quad = sys.argv[2] if len(sys.argv) > 2 else "ul7n12.dat"
if what.startswith("strat:"):
    n_sites = int(what.split(":")[1])
    H = 2.0e6
    bounds = (-0.5e6, 14.0e6, 0.0, 6.0e6, 0.0, 6.0e6)
    rng = np.random.default_rng(11)
    u = rng.random(n_sites)
    Lz = bounds[1] - bounds[0]
    pos = np.stack([bounds[0] - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H))),
                    bounds[2] + rng.random(n_sites) * (bounds[3] - bounds[2]),
                    bounds[4] + rng.random(n_sites) * (bounds[5] - bounds[4])], axis=1)
    nbr = vrt.voro(pos, bounds)
else:
    a, c = synth.BCC_CONFIGS[what]
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=2022)
hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
so = orc.make_sites(pos, nbr, bounds)
w, th, ph, nq = vrt.read_quadrature(quad)
n = hs.n
LINE = 8


def lines_of(positions_per_patch_flat, patch_id):
    """distinct (patch, line) pairs"""
    key = patch_id.astype(np.int64) * (1 << 32) + (positions_per_patch_flat // LINE)
    return np.unique(key).size


def uniq_sites(positions, patch_id):
    key = patch_id.astype(np.int64) * (1 << 32) + positions
    return np.unique(key).size


for d in (1, -1):
    store = hs.storage_order(d) - 1 if hasattr(hs, "storage_order") else None
    if store is None:
        raise SystemExit("need VoronoiSites.storage_order")
    layers = (hs.layers_up if d > 0 else hs.layers_down).astype(np.int64)       # reduced offsets, 1-based
    lay_lo = layers - 1
    pos_of = np.empty(n, dtype=np.int64)
    pos_of[store] = np.arange(n)
    layer_of_pos = np.searchsorted(lay_lo, np.arange(n), side="right")          # 1-based layer of a position
    layer_of_pos[n - 1] = layer_of_pos[n - 2] if n > 1 else 1
    layer_of_site = np.empty(n, dtype=np.int64)
    layer_of_site[store] = layer_of_pos
    ups = []
    sel = [i for i in range(nq) if (th[i] > 90) == (d > 0) and th[i] != 90]
    for ai in sel:
        k = orc.direction(th[ai], ph[ai])
        ups.append(orc.upwind_table(so, k)[0])          # (n, 2) 1-based ids
    # how many angles gather a site from ANOTHER layer
    cnt_next = np.zeros(n, dtype=np.int64)
    for up in ups:
        for r in range(2):
            u = up[:, r] - 1
            other = layer_of_site[u] != layer_of_site
            # only sites that are themselves visited count (layer >= 2)
            g = np.unique(u[other & (layer_of_site >= 2)])
            cnt_next[g] += 1
    # geometric proxy: neighbours in the next layer
    nb = np.asarray(nbr)
    # candidate orders: rank inside the layer
    morton_rank = pos_of.copy()                          # present order = (layer, Morton)

    def order_from_class(cls):
        key = layer_of_site * (1 << 40) + cls.astype(np.int64) * (1 << 32) + morton_rank
        key[store[n - 1]] = np.iinfo(np.int64).max      # the never-visited site stays last
        o = np.argsort(key, kind="stable")
        p = np.empty(n, dtype=np.int64)
        p[o] = np.arange(n)
        return p

    cands = {"morton": pos_of}
    A = len(ups)
    cands["split any"] = order_from_class((cnt_next == 0).astype(int))
    cands["split all"] = order_from_class((cnt_next < A).astype(int))
    cands["split half"] = order_from_class((cnt_next * 2 < A).astype(int))
    cands["3 classes"] = order_from_class(np.where(cnt_next == 0, 2, np.where(cnt_next < A, 1, 0)))
    cands["by count"] = order_from_class(A - cnt_next)
    print(f"{what} dir {d:+d}: {n} sites, {len(sel)} angles; gathered by n angles: "
          f"{np.bincount(cnt_next, minlength=A + 1).tolist()}", flush=True)
    for j, ai in enumerate(sel):
        up = ups[j]
        ps = vrt.api.build_patch_schedule(hs, d, up, 3, 512, 512)
        eo = ps["patch_ent_off"]
        pid = np.repeat(np.arange(ps["patches"]), np.diff(eo))
        esite = store[ps["entry_pos"]]
        u1 = up[esite, 0] - 1
        u2 = up[esite, 1] - 1
        el = layer_of_site[esite]
        g_this = np.zeros(n, dtype=bool)
        for r in range(2):
            u = up[:, r] - 1
            other = layer_of_site[u] != layer_of_site
            g_this[u[other & (layer_of_site >= 2)]] = True
        g_early = np.zeros(n, dtype=bool)
        for r in range(2):
            u = up[:, r] - 1
            g_early[u[(layer_of_site[u] < layer_of_site) & (layer_of_site >= 2)]] = True
        cands["this angle"] = order_from_class((~g_this).astype(int))
        cands["this, early"] = order_from_class((~g_early).astype(int))
        out = f"  angle {ai:2d} theta {th[ai]:6.1f} entries/own {ps['entries'] / ps['patch_own_cnt'].sum():5.3f}:"
        print(out, flush=True)
        for name, P in cands.items():
            own_lines = lines_of(P[esite], pid)
            own_bytes = esite.size
            leave = np.concatenate([(layer_of_site[u1] != el), (layer_of_site[u2] != el)])
            gu = np.concatenate([u1, u2])
            gp = np.concatenate([pid, pid])
            lv_lines = lines_of(P[gu[leave]], gp[leave])
            lv_sites = uniq_sites(gu[leave], gp[leave])
            earlier = np.concatenate([(layer_of_site[u1] < el), (layer_of_site[u2] < el)])
            e_lines = lines_of(P[gu[earlier]], gp[earlier])
            e_sites = uniq_sites(gu[earlier], gp[earlier])
            all_lines = lines_of(np.concatenate([P[esite], P[gu]]), np.concatenate([pid, gp]))
            print(f"      {name:11s} own lines x{own_lines * LINE / own_bytes:5.2f}  leaving-layer gathers: "
                  f"{lv_sites / esite.size:5.3f} sites/entry, lines x{lv_lines * LINE / max(lv_sites, 1):5.2f}  "
                  f"earlier (I): x{e_lines * LINE / max(e_sites, 1):5.2f}  "
                  f"all lines of the patch / entry {all_lines * LINE / esite.size:5.2f}", flush=True)
