#!/usr/bin/env python3
"""diagnostics: BASELINE configs[0] (regular 60^3 searchlight, one vertical ray) through the regular solver -- wall time of
vrt_regular_execute_dev per call and of the solve kernels alone (HIP events), for the split xy form and the single
kernel (VRT_REG_XY=0).  usage: python tools/c1_probe.py [n]     (under rocprofv3 --kernel-trace for per-kernel times)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
z = x = y = np.linspace(0, 1, n)
rng = np.random.default_rng(1)
S = rng.random((n, n, n)); al = rng.random((n, n, n)) * 3; I0 = rng.random((n, n))
dev = torch.device("cuda", 0)
w, th, ph, _ = vrt.read_quadrature("n1.dat")
k = vrt.direction(th[0], ph[0])[None]
Sd, Ad, I0d = (torch.from_numpy(a).to(dev) for a in (S, al, I0[None]))
out = torch.empty((1, n, n, n), device=dev, dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ("0", "1", "2"):
    os.environ["VRT_REG_XY"] = mode
    solver = vrt.RegularSolver(z, x, y, device=0)
    for _ in range(3):
        solver.execute_dev(k, [True], Sd.data_ptr(), 0, Ad.data_ptr(), 0, I0d.data_ptr(), out.data_ptr(), 3, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        solver.execute_dev(k, [True], Sd.data_ptr(), 0, Ad.data_ptr(), 0, I0d.data_ptr(), out.data_ptr(), 3, st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    res[mode] = out.clone()
    print("VRT_REG_XY=%s: %.3f ms per call (wall, 50 calls back to back), solve kernels alone %.3f ms" % (mode, wall, solver.last_solve_ms()), flush=True)
    solver.close()
print("bitwise equal:", bool(torch.equal(res["0"], res["1"]) and torch.equal(res["0"], res["2"])))
