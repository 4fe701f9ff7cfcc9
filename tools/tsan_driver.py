#!/usr/bin/env python3
"""ThreadSanitizer driver of the library's threaded HOST code (tools/asan_host.sh tsan; CPU only, host-only handles):
the layer-parallel patch builder on an atomic work counter (csrc/vrt_patch.cpp, VRT_HOST_THREADS = 16 builder threads),
several of them at once from Python threads (the way plan creation runs one job per angle), the angle-parallel level
schedules (run_workers pools), the in-process tessellation; results compared with a single-threaded build."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
import voronoirt_amd.api as api  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from voronoirt_amd import synth  # noqa: E402

pos, nbr, bounds = synth.bcc_grid(12, 10, seed=3)
hs = vrt.VoronoiSites(pos, nbr, bounds, device=-1)
so = orc.make_sites(pos, nbr, bounds)
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
ups = {ai: orc.upwind_table(so, orc.direction(th[ai], ph[ai]))[0] for ai in range(nq)}
os.environ["VRT_HOST_THREADS"] = "1"
serial = {ai: api.build_patch_schedule(hs, 1 if th[ai] > 90 else -1, ups[ai], 3, 512, 512) for ai in (0, 1, 2, 3)}
os.environ["VRT_HOST_THREADS"] = "16"
results = {}


def job(ai):
    results[ai] = api.build_patch_schedule(hs, 1 if th[ai] > 90 else -1, ups[ai], 3, 512, 512)
    api.build_layer_schedule(hs, 1 if th[ai] > 90 else -1, ups[ai], 3)
    api.build_schedule(hs, 1 if th[ai] > 90 else -1, ups[ai], 3)


threads = [threading.Thread(target=job, args=(ai,)) for ai in (0, 1, 2, 3)]
for t in threads:
    t.start()
for t in threads:
    t.join()
for ai, ref in serial.items():
    for key in ("patch_own_lo", "patch_own_cnt", "patch_ent_off", "entry_pos", "entry_vis", "entry_loc", "dep_off", "dep_list", "layer_vis"):
        assert np.array_equal(results[ai][key], ref[key]), (ai, key)
rng = np.random.default_rng(1)
p2 = np.stack([rng.random(3000) * 2.0, rng.random(3000), rng.random(3000)], axis=1)
n1 = vrt.voro(p2, (0.0, 2.0, 0.0, 1.0, 0.0, 1.0))
n2 = vrt.voro(p2, (0.0, 2.0, 0.0, 1.0, 0.0, 1.0))
assert np.array_equal(n1, n2)
hs.close()
print("tsan driver: 4 concurrent builds x 16 builder threads equal the serial build; tessellation twice equal")
