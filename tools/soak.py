"""Soak / race screen (GPU): repeats the formal solve on every device path and checks that the
result is bitwise identical from run to run and equal across paths to 1e-12."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
from voronoirt_amd import _lib, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
a, c, nlam = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (30, 40, 24)
pos, nbr, bounds = synth.bcc_grid(a, c, seed=3)
hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
n = hs.n
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
plan = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
z = torch.as_tensor(pos[:, 0], device=dev)
S = 1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
al = (1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6))[:, None] * (1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64))
n1 = int(hs.layers_up[1] - 1)
I0 = torch.rand((n1, nlam), generator=g, device=dev, dtype=torch.float64)
stream = torch.cuda.current_stream().cuda_stream
ref = {}
for path in ("levels", "steps", "tiles", "patches"):
    plan.set_option("VRT_PATH", path)
    first = None
    for r in range(reps):
        J = torch.empty((n, nlam), dtype=torch.float64, device=dev)
        plan.execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(),
                         dI0_up=I0.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        if first is None:
            first = J.clone()
        elif not torch.equal(J, first):
            bad = (J != first).sum().item()
            print(f"RACE? path {path} rep {r}: {bad} elements differ, max {float((J - first).abs().max())}")
            sys.exit(1)
    ref[path] = first
    print(path, "ok:", reps, "identical runs,", n, "sites")
for p in ("steps", "tiles", "patches"):
    d = float((ref[p] - ref["levels"]).abs().max() / ref["levels"].abs().max())
    print(p, "vs levels max rel diff", d)
    assert d < 5e-12
print("soak ok")
