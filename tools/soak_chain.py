"""Soak of the chained patch launch (GPU): `reps` steps on the chained launch, every one compared BIT FOR BIT
with the per-layer launches' J; between the steps the inputs change (so that a stale intensity, progress word or
cache line of the previous step would show) and the GPU is kept unevenly loaded by a second stream of elementwise work.
usage: python tools/soak_chain.py [reps] [a c nlam]     (exit code 1 on the first difference or give-up)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voronoirt_amd as vrt  # noqa: E402
from voronoirt_amd import _lib, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
a, c, nlam = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (24, 40, 6)
pos, nbr, bounds = synth.bcc_grid(a, c, seed=3)
hs = vrt.VoronoiSites(pos, nbr, bounds, device=0)
n = hs.n
w, th, ph, nq = vrt.read_quadrature("ul7n12.dat")
os.environ["VRT_PATH"] = "patches"
os.environ["VRT_PATCH_OWN"] = os.environ.get("VRT_PATCH_OWN", "200")          # several patches per layer
plans = {}
for name, chain in (("chain", "1"), ("launches", "0")):
    os.environ["VRT_PATCH_CHAIN"] = chain
    plans[name] = vrt.FormalPlan(hs, vrt.quadrature_directions(th, ph), 3)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
z = torch.as_tensor(pos[:, 0], device=dev)
n1 = int(hs.layers_up[1] - 1)
stream = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
junk = torch.rand(1 << 22, device=dev)
variants = []
for v in range(4):                                   # four input sets, their reference J from the per-layer launches
    S = 1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64)
    al = (1e-2 * torch.exp(-(z - bounds[0]) / 0.7e6))[:, None] * (1 + torch.rand((n, nlam), generator=g, device=dev, dtype=torch.float64))
    I0 = torch.rand((n1, nlam), generator=g, device=dev, dtype=torch.float64)
    J = torch.empty((n, nlam), dtype=torch.float64, device=dev)
    plans["launches"].execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=J.data_ptr(),
                                  dI0_up=I0.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    variants.append((S, al, I0, J))
t0 = time.time()
Jc = torch.empty((n, nlam), dtype=torch.float64, device=dev)
for r in range(reps):
    S, al, I0, Jref = variants[r % 4]
    if r % 3 == 0:                                   # uneven load beside the sweep
        with torch.cuda.stream(side):
            junk.mul_(1.0000001).add_(1e-9)
    Jc.fill_(float("nan"))
    plans["chain"].execute_dev(nlam, nlam, S.data_ptr(), al.data_ptr(), _lib.ALPHA_SITE_LAM, w, dJ=Jc.data_ptr(),
                               dI0_up=I0.data_ptr(), stream=stream)
    if r % 16 == 15 or r == reps - 1:                # (every step is checked; the comparison is batched to keep the queue full)
        torch.cuda.synchronize()
    if not torch.equal(Jc, Jref):
        torch.cuda.synchronize()
        bad = int((Jc != Jref).sum().item())
        print(f"MISMATCH at step {r}: {bad} elements differ, max {float((Jc - Jref).abs().nan_to_num(nan=1e300).max()):.3e}")
        sys.exit(1)
    if r % 500 == 499:
        assert plans["chain"].last_launches == 1     # (also raises if a chained launch gave up)
        print(f"step {r + 1}: ok ({time.time() - t0:.0f} s)", flush=True)
assert plans["chain"].last_launches == 1
print(f"soak ok: {reps} chained steps bitwise equal to the per-layer launches ({n} sites x 12 angles x {nlam} wavelengths)")
