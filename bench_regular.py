#!/usr/bin/env python3
"""Measurement of the regular-grid short-characteristics solver, the comparison
solver the reference times beside the Voronoi one (src/compare_searchlight.jl:358-499 `do_timing`:
the Bifrost cube read with skip = 2, one formal solve per quadrature direction).

A step = one batch of formal solves (every direction of the quadrature x `--nlam` wavelengths that
share the direction but have their own S and alpha) on a synthetic cube of the same shape
(nz, nx, ny) = (215, 128, 128) + the reference's one-cell periodic ghost border in x and y, fields
resident in HBM.  Reported: cell-updates/s (cell = one (grid point, direction, wavelength)
intensity), the solve kernel's HIP-event time and its fraction of the HBM roofline on algorithmic
bytes (read S, alpha 16 B + write I 8 B per cell-update), and the CPU oracle beside it on a bounded
sample.  Prints ONE JSON line.

    python bench_regular.py [--nlam L] [--steps K] [--warmup W] [--shape nz nx ny]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nlam", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", type=int, nargs=3, default=[215, 128, 128])
    ap.add_argument("--quadrature", default="ul7n12.dat")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import torch
    import voronoirt_amd as vrt
    if not torch.cuda.is_available():
        raise SystemExit("bench_regular.py needs a HIP device (there is no CPU fallback)")
    dev = torch.device("cuda", 0)
    nz, nx, ny = args.shape[0], args.shape[1] + 2, args.shape[2] + 2       # ghost border in x, y
    w, th, ph, nq = vrt.read_quadrature(args.quadrature)
    ks = vrt.quadrature_directions(th, ph)
    z = np.linspace(-0.5e6, 14.0e6, nz)
    x = np.linspace(0.0, 6.0e6, nx)
    y = np.linspace(0.0, 6.0e6, ny)
    ns = nq * args.nlam
    k_all = np.repeat(ks, args.nlam, axis=0)
    up_all = np.repeat(th > 90, args.nlam)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1998)
    zz = torch.as_tensor(z, device=dev)
    # per-wavelength fields (nlam, ny, nx, nz) = Julia (nz, nx, ny, nlam); the directions share them
    strat = 1.0e-5 * torch.exp(-(zz - z[0]) / 1.0e6)
    S = 1.0 + 0.1 * torch.rand((args.nlam, ny, nx, nz), generator=gen, device=dev, dtype=torch.float64)
    al = strat[None, None, None, :] * (1.0 + torch.rand((args.nlam, ny, nx, nz), generator=gen, device=dev,
                                                        dtype=torch.float64))
    # solves ordered direction-major: solve (a, l) uses field l -> gather the per-solve arrays once
    # per wavelength by running one execute per direction with per-solve stride
    I0 = torch.rand((ns, ny, nx), generator=gen, device=dev, dtype=torch.float64)
    out = torch.empty((ns, ny, nx, nz), device=dev, dtype=torch.float64)
    solver = vrt.RegularSolver(z, x, y, device=0)
    vol = nz * nx * ny
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # every (direction, wavelength) solve in one batched launch: solve a * nlam + l reads the
        # fields of wavelength l (field_period = nlam)
        solver.execute_dev(k_all, up_all, S.data_ptr(), vol, al.data_ptr(), vol, I0.data_ptr(), out.data_ptr(),
                           3, stream, field_period=args.nlam)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / args.steps
    # solve-kernel time of one direction batch (events), measured on the last execute
    last_ms = solver.last_solve_ms()
    updates = vol * ns
    alg_bytes = 24.0 * vol * ns
    res = {
        "metric": "regular-grid formal-solve cell-updates/sec", "value": updates / elapsed, "unit": "cell-updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3,
        "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"regular grid (nz, nx, ny) = ({nz}, {nx}, {ny}) incl. ghost border, "
                               f"{args.quadrature} ({nq} directions) x {args.nlam} wavelengths, n_sweeps=3"},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (last_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg_bytes / (last_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_regular_solve (one workgroup per solve)", "solve_kernel_ms": last_ms,
                     "algorithmic_bytes_per_launch": alg_bytes},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle as orc
        S_h, al_h, I0_h = S[0].cpu().numpy(), al[0].cpu().numpy(), I0[0].cpu().numpy()
        t0 = time.time()
        f = orc.short_characteristics_up if th[0] > 90 else orc.short_characteristics_down
        ref = f(ks[0], S_h, I0_h, al_h, z, x, y, 3)
        t_cpu = time.time() - t0
        got = out[0].cpu().numpy()              # solve 0 = direction 0, wavelength 0 of the last step
        res["cpu_baseline"] = {"value": vol / t_cpu, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                               "sample": f"one direction x one wavelength of the same cube ({vol} cells, {t_cpu:.1f} s)"}
        res["parity_vs_oracle_max_rel_err"] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
