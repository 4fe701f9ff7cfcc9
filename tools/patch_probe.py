#!/usr/bin/env python3
"""diagnostics (host only): size of the in-layer dependency cones of the patch schedule -- how many
entries / visits the fused patch kernel executes per owned site for a given patch size.
usage: python tools/patch_probe.py [C4|C2|C5|strat:<n>] [quadrature] [angle indices...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
import voronoirt_amd.api  # noqa: E402,F401
from oracle import oracle as orc  # noqa: E402
from voronoirt_amd import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "C4"
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
angles = [int(x) for x in sys.argv[3:]] or list(range(nq))
print(f"{what}: {hs.n} sites, layers up {len(hs.layers_up) - 1} (max {np.diff(hs.layers_up).max()}), "
      f"down {len(hs.layers_down) - 1} (max {np.diff(hs.layers_down).max()})", flush=True)
for ai in angles:
    k = orc.direction(th[ai], ph[ai])
    up = orc.upwind_table(so, k)[0]
    d = 1 if th[ai] > 90 else -1
    for own, cap in ((512, 512),):
        t0 = time.time()
        ps = vrt.api.build_patch_schedule(hs, d, up, 3, own, cap)
        own_sites = int(ps["patch_own_cnt"].sum())
        ent = np.diff(ps["patch_ent_off"])
        print(f"angle {ai:2d} theta {th[ai]:6.1f} own {own:5d} cap {cap:5d}: patches {ps['patches']:6d} "
              f"entries/own {ps['entries'] / own_sites:5.3f} visits/live {ps['visits'] / ps['live_visits']:5.3f} "
              f"live/site {ps['live_visits'] / own_sites:4.2f} max entries {ps['max_entries']:5d} "
              f"mean {ent.mean():6.0f} nlev mean {ps['patch_nlev'].mean():4.1f} max {ps['patch_nlev'].max():3d} "
              f"({time.time() - t0:.1f} s)", flush=True)


def barrier_stats(ps, order_fn=None, wave=64):
    """Level-loop barriers a patch needs when only CROSS-wave hazards are fenced: a barrier before level t is
    needed when a visit of level t reads a tile slot of another wave written since the last barrier (RAW), or
    overwrites its own slot that another wave has read since the last barrier (WAR).  order_fn(patch entries) ->
    permutation of the entries (thread assignment); None = the schedule's own order.  Returns (levels, barriers
    needed, fraction of in-patch read-from edges inside one wave) summed over the patches."""
    eo = ps["patch_ent_off"]
    tot_lev = tot_bar = 0
    edges = same = 0
    for q in range(ps["patches"]):
        e0, e1 = int(eo[q]), int(eo[q + 1])
        ne = e1 - e0
        vis = ps["entry_vis"][e0:e1]
        loc = ps["entry_loc"][e0:e1]
        l1, l2 = (loc & 0xFFFF).astype(np.int64), (loc >> 16).astype(np.int64)
        perm = np.arange(ne) if order_fn is None else order_fn(vis, l1, l2)
        thread_of = np.empty(ne, dtype=np.int64)
        thread_of[perm] = np.arange(ne)                       # entry -> thread
        wv = thread_of // wave
        nlev = int(ps["patch_nlev"][q])
        written_since = np.zeros(ne, dtype=bool)              # slot written since the last barrier
        readers_since = [set() for _ in range(ne)]            # waves that read the slot since the last barrier
        bars = 0
        for t in range(1, nlev + 1):
            hit = np.nonzero(((vis & 0xFF) == t) | (((vis >> 8) & 0xFF) == t) | (((vis >> 16) & 0xFF) == t) | ((vis >> 24) == t))[0]
            need = False
            for e in hit:
                for s in (l1[e], l2[e]):
                    if s == 0xFFFF:
                        continue
                    edges += 1
                    same += wv[s] == wv[e]
                    if wv[s] != wv[e] and written_since[s]:
                        need = True
                if any(w != wv[e] for w in readers_since[e]):
                    need = True
            if need:
                bars += 1
                written_since[:] = False
                readers_since = [set() for _ in range(ne)]
            for e in hit:
                for s in (l1[e], l2[e]):
                    if s != 0xFFFF:
                        readers_since[s].add(wv[e])
            written_since[hit] = True
        tot_lev += nlev
        tot_bar += bars
    return tot_lev, tot_bar, same / max(edges, 1)


def cluster_order(vis, l1, l2):
    """entries grouped with their in-patch upwinds: depth-first along the read-from edges, roots by first visit level"""
    ne = vis.size
    children = [[] for _ in range(ne)]
    for e in range(ne):
        for s in (l1[e], l2[e]):
            if s != 0xFFFF and s != e:
                children[s].append(e)
    seen = np.zeros(ne, dtype=bool)
    out = []
    for r in np.argsort(vis & 0xFF, kind="stable"):
        if seen[r]:
            continue
        stack = [r]
        while stack:
            e = stack.pop()
            if seen[e]:
                continue
            seen[e] = True
            out.append(e)
            stack.extend(c for c in children[e] if not seen[c])
    return np.array(out, dtype=np.int64)


if os.environ.get("PROBE_BARRIERS"):
    for ai in angles:
        k = orc.direction(th[ai], ph[ai])
        up = orc.upwind_table(so, k)[0]
        d = 1 if th[ai] > 90 else -1
        ps = vrt.api.build_patch_schedule(hs, d, up, 3, 512, 512)
        # a sample of the patches (the analysis is a Python loop)
        keep = np.linspace(0, ps["patches"] - 1, min(ps["patches"], int(os.environ.get("PROBE_SAMPLE", "200")))).astype(int)
        sub = dict(ps)
        sub["patches"] = keep.size
        eo = ps["patch_ent_off"]
        sub["patch_ent_off"] = np.concatenate([[0], np.cumsum((eo[keep + 1] - eo[keep]))])
        sub["patch_nlev"] = ps["patch_nlev"][keep]
        sub["entry_vis"] = np.concatenate([ps["entry_vis"][eo[q]:eo[q + 1]] for q in keep])
        sub["entry_loc"] = np.concatenate([ps["entry_loc"][eo[q]:eo[q + 1]] for q in keep])
        for name, fn in (("schedule order", None), ("clustered", cluster_order)):
            lev, bar, frac = barrier_stats(sub, fn)
            print(f"angle {ai:2d} theta {th[ai]:6.1f} {name:15s}: levels/patch {lev / keep.size:5.1f} cross-wave barriers/patch "
                  f"{bar / keep.size:5.1f} in-wave edges {frac:5.3f}", flush=True)
