"""Soak of several chained patch launches SHARING the chip (GPU): vrt_multi handles on one device (1, 3 and 4 of them),
the wavelength count changing from call to call (every call rebuilds the chained launch's item sets), wavelength-block
and angle sharding alternating; every J against the oracle at 1e-10 element-wise.  (Found: progress words zeroed on the
null stream racing with launches on the plans' non-blocking streams.)   usage: python tools/soak_multi.py [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import voronoirt_amd as vrt
from oracle import oracle as orc
from oracle.parity import rel
from voronoirt_amd import synth, _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pos, nbr, bounds = synth.voronoi_grid(1500, seed=5, bounds=(0.0, 2.0, 0.0, 1.0, 0.0, 1.0), scale_height=0.7)
so = orc.make_sites(pos, nbr, bounds)
n = so.n
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
dirs = [1 if t > 90 else -1 for t in th]
os.environ["VRT_PATCH_CHAIN"] = "1"
bad = tot = err = 0
t0 = time.time()
for devices in ((0,), (0, 0, 0), (0, 0, 0, 0)):
    mp = vrt.MultiDevicePlan(pos, nbr, bounds, vrt.quadrature_directions(th, ph), dirs=dirs, devices=devices)
    rng = np.random.default_rng(len(devices))
    cases = {}
    for nlam in (1, 2, 3):
        S = 1 + rng.random((n, nlam))
        al = 5 * 10 ** rng.uniform(-2, 2, (n, 1)) * (1 + rng.random((n, nlam)))
        I0u, I0d = rng.random((so.layers_up[1] - 1, nlam)), rng.random((so.layers_down[1] - 1, nlam))
        cases[nlam] = (S, al, I0u, I0d, orc.J_voronoi(w, th, ph, S, al, so, I0_up=I0u, I0_down=I0d, nthreads=4))
    for rep in range(reps):
        nlam = (1, 3, 2, 3, 1, 2, 2)[rep % 7]           # the wavelength count changes from call to call: the item sets are rebuilt
        S, al, I0u, I0d, ref = cases[nlam]
        mode = "angle" if (rep % 3 and len(devices) > 1) else "lambda"
        mp.set_shard(mode)
        try:
            J = mp.execute(S, al, w, I0_up=I0u, I0_down=I0d)
        except Exception as e:
            err += 1
            print("handles", len(devices), "nlam", nlam, "rep", rep, mode, "ERROR", str(e)[-300:], flush=True)
            continue
        e = rel(J, ref)
        tot += 1
        if not e < 1e-10:
            bad += 1
            nb = int((np.abs(J - ref) > 1e-10 * np.abs(ref)).sum())
            print("handles", len(devices), "nlam", nlam, "rep", rep, mp.last_shard, e, "elements", nb, flush=True)
    mp.close()
    print("handles", len(devices), "done: bad", bad, "errors", err, "of", tot, f"({time.time() - t0:.0f} s)", flush=True)
