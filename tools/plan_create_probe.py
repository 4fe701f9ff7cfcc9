#!/usr/bin/env python3
"""diagnostics: vrt_plan_create at C4 size (250k-site bcc grid, 12 angles), phase by phase when the -DVRT_DIAG library is
loaded (VRT_LIB_PATH=voronoirt_amd/libvrt_hip_diag.so), and end to end.  usage: python tools/plan_create_probe.py [a c [quadrature]]   (C4: 59 143 ul7n12.dat, C5: 94 227 ul9n20.dat)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt
from voronoirt_amd import synth, api
nz, nxy = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (59, 143)
quad = sys.argv[3] if len(sys.argv) > 3 else "ul7n12.dat"
pos, nbr, bounds = synth.bcc_grid(nz, nxy, seed=2022)
sites = vrt.VoronoiSites(pos, nbr, bounds, device=0)
print("sites", sites.n, "host threads", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    plan, w = api._quadrature_plan(sites, quad, 3)
    print("plan_create %d: %.3f s" % (rep, time.perf_counter() - t0), flush=True)
    sites._plans.clear()
    del plan
