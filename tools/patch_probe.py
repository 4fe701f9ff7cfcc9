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
