#!/usr/bin/env python3
"""Benchmark of the Voronoi formal solve (BASELINE.json metric: formal-solve cell-updates/s and
achieved HBM GB/s on a ~1M-site Voronoi grid with the 12 angles of ul7n12).

A "step" is one evaluation of the angle x wavelength loop of J_λ_voronoi
(src/lambda_iteration.jl:84-111 of the reference): every site x angle x wavelength intensity
(each including its 3 Gauss-Seidel sweeps) plus the J = Σ w·I reduction, with S, α and I_0
already resident in HBM and the per-angle upwind tables / sweep schedule built beforehand (they
depend on the grid and the quadrature only; their one-time cost is reported separately).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C2|C3|C5|tiny] [--nlam L]
                    [--shard lambda|lambda-strong|angle]

N > 1: one rank per GPU over RCCL.  Launched by `python -m torch.distributed.run ...` (RANK /
WORLD_SIZE in the environment) the script joins that job; launched bare (`python bench.py --gpus 8`)
it starts its own N ranks through torch.distributed.run BEFORE anything touches a GPU and relays
rank 0's JSON line.  Sharding; the default for N > 1 is the FIXED workload
(`auto`: lambda-strong when the workload has at least N wavelengths, angle otherwise):
  lambda-strong  the FIXED problem of the workload: its nlam wavelengths are split into contiguous
                 blocks (51 over 8 ranks -> 7,7,7,6,6,6,6,6), every rank solves all angles for its
                 block, J is all-gathered over RCCL inside every timed step
  angle          strong scaling over the angles (needed when nlam < N): partial J's summed with one
                 RCCL all-reduce per step -- the scheme BASELINE.json's north star names
  lambda         weak scaling, opt-in: every rank owns a block of `nlam` wavelengths of a global
                 N x nlam problem and all angles for it, i.e. whole rows of J: no data-path collective

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# the CPU-baseline leg runs the oracle on an OpenMP team; idle OpenMP workers that spin after their region would eat
# the box's CPU share while the secondary (C2) measurement enqueues its ~10 launches per 0.45-ms step (seen: 1.76 ms wall
# per step against 0.45 ms of HIP-event time).  Read by libgomp when it is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E vendor peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r5"

WORKLOADS = {
    # name: (a, c, quadrature, nlam, alpha per angle, seed)   -- SURVEY.md 8d, BASELINE.md sec. 3
    "C4": (59, 143, "ul7n12.dat", 51, True, 2022),    # ~1M sites, 12 angles x 51 λ (line)
    "C3": (59, 143, "ul9n20.dat", 20, False, 1998),   # ~1M sites, 20 angles x 20 λ (continuum)
    "C2": (37, 90, "ul7n12.dat", 1, False, 1998),     # ~250k sites, 12 angles x 1 λ
    "C5": (94, 227, "ul9n20.dat", 100, False, 1998),  # ~4M sites, 20 angles x 100 λ (use --dtype f32)
    "tiny": (8, 12, "ul7n12.dat", 4, True, 7),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--nlam", type=int, default=0, help="override the workload's wavelength count")
    ap.add_argument("--shard", default="auto", choices=["auto", "lambda", "lambda-strong", "angle"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary C2 measurement")
    ap.add_argument("--no-critical-path", action="store_true",
                    help="skip the one-wavelength-pair latency-floor measurement (profiling runs)")
    ap.add_argument("--cpu-lam", type=int, default=0, help="wavelengths in the CPU sample")
    ap.add_argument("--angle-groups", type=int, default=1,
                    help="diagnostics: run the angles in this many sequential groups (separate plans)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="storage type of S, alpha, I, J (arithmetic is always fp64)")
    ap.add_argument("--alpha-layout", default="native", choices=["native", "caller"],
                    help="per-angle alpha handed over in the library's native storage-pair layout "
                         "(converted once before the timed region, the way a producer such as the "
                         "opacity prologue writes it) or in the caller's (n_angles, n, nlam) layout "
                         "(transposed inside every step)")
    ap.add_argument("--sj-layout", default="auto", choices=["auto", "native", "caller"],
                    help="S and J of the timed step: 'native' = the sweep's own per-direction plane sets, what the "
                         "device-resident Λ-iteration (vrt_lambda_iterate) keeps between its steps -- no layout change "
                         "in the step; 'caller' = (n, nlam) arrays in, J (n, nlam) out (two layout changes per step); "
                         "auto: native on one GPU wherever the library offers it (per-angle alpha in its native layout; "
                         "alpha per (site, wavelength) -- the continuum's, fixed over the iterations -- laid out once in sweep "
                         "order), the caller-layout time is reported beside it")
    ap.add_argument("--no-caller-layout", action="store_true",
                    help="with --sj-layout native: do not time the caller-layout step beside it (profiling runs: the process "
                         "then executes the headline step only)")
    ap.add_argument("--alpha0", type=float, default=1.0e-2,
                    help="opacity scale at z_min [1/m] (diagnostics: tiny values take the Taylor branch)")
    ap.add_argument("--dump-J", default="", help="rank 0 saves the (gathered) J of the last step as .npy")
    return ap.parse_args(argv)


def source_digest() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: ties a committed PMC profile to the
    library it was taken with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "voronoirt_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


ONE_TIME_KERNELS = ("k_upwind_table", "k_permute_table", "k_delaunay_lines", "k_sorted_tables", "k_sorted_loc", "k_sorted_code",
                    "k_gpos", "k_patch_entries")


def pmc_traffic_per_step(workload, path, nlam, world, alpha_layout, sj_layout="caller"):
    """Fabric-side bytes of ONE step from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/c4_summary.json; PMC counters cannot be read from inside the run).
    bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B): on gfx950 FETCH_SIZE tallies the 128-B requests
    of these streams at 64 B (MI355X_MICROARCH.md, HBM section).  None unless the profile was
    taken with exactly this library (source digest), workload, path and layout."""
    if workload != "C4" or nlam != WORKLOADS["C4"][3] or world != 1:
        return None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "c4_summary.json")))
        if prof.get("source_digest") != source_digest() or prof.get("alpha_layout") != alpha_layout \
                or prof.get("path") != path or prof.get("sj_layout", "caller") != sj_layout:
            return None
        total = 0.0
        for name, c in prof["pmc_one_step"].items():
            if name.startswith(("vrt::", "void vrt::")) and not any(k in name for k in ONE_TIME_KERNELS) \
                    and "note_not_part_of_a_step" not in c:
                total += (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        return total if total > 0 else None
    except Exception:
        return None


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measure(args, workload, torch, dist, vrt, _lib, distributed, synth, rank, world, local, dev, rehearse,
            steps, warmup, cpu_baseline):
    """One workload: setup, warm-up, timed steps, roofline, optional CPU baseline.  Returns the
    result dict (rank 0 holds the CPU baseline / parity fields)."""
    a, c, quad, nlam_total, per_angle, seed = WORKLOADS[workload]
    if args.nlam > 0 and workload == args.workload:
        nlam_total = args.nlam
    dist_on = world > 1 or (dist.is_available() and dist.is_initialized())
    shard = args.shard if dist_on else "lambda"
    if shard == "auto":                     # N > 1: the fixed workload, RCCL on J inside the timed step
        shard = "lambda-strong" if nlam_total >= world else "angle"
    weights, theta, phi, n_angles = vrt.read_quadrature(quad)

    # wavelengths this rank solves: its own block of `nlam_total` (weak), or its part of the fixed set
    lam_lo, lam_hi = 0, nlam_total
    if shard == "lambda-strong":
        lam_lo, lam_hi = distributed.partition(nlam_total, world, rank)
    nlam = lam_hi - lam_lo

    # ---- one-time setup (untimed; reported) --------------------------------------------------
    t0 = time.time()
    pos, nbr, bounds = synth.bcc_grid(a, c, seed=seed)
    t_gen = time.time() - t0
    t0 = time.time()
    sites = vrt.VoronoiSites(pos, nbr, bounds, device=local)
    t_grid = time.time() - t0
    n = sites.n

    my_angles = np.arange(n_angles)
    if shard == "angle":
        my_angles = distributed.angle_assignment(theta, world)[rank]
    k_all = vrt.quadrature_directions(theta, phi)
    dirs_all = np.array([1 if t > 90 else (-1 if t < 90 else 0) for t in theta])
    t0 = time.time()
    plan = vrt.FormalPlan(sites, k_all[my_angles], 3, dirs=dirs_all[my_angles])
    t_plan = time.time() - t0
    A = len(my_angles)

    # ---- synthetic fields, generated on the device (SURVEY 8d shapes) -------------------------
    # the fixed-problem modes (lambda-strong, angle) generate the SAME global field on every rank
    # (same seed) and keep their part of it, so the N-rank J can be compared with the 1-rank J
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + (1000 * rank if shard == "lambda" else 0))
    z = torch.as_tensor(pos[:, 0], device=dev)
    z_min, z_max = bounds[0], bounds[1]
    lam = torch.arange(nlam_total, device=dev, dtype=torch.float64)
    centre, sigma = 0.5 * (nlam_total - 1), max(nlam_total / 6.0, 1.0)
    S = 1.0 + 0.5 * torch.sin(2 * np.pi * (z - z_min) / (z_max - z_min))[:, None] \
        + 0.1 * torch.rand((n, nlam_total), generator=gen, device=dev, dtype=torch.float64)
    strat = args.alpha0 * torch.exp(-(z - z_min) / 0.7e6)
    if per_angle:
        alpha = torch.empty((A, n, nlam), device=dev, dtype=torch.float64)
        mine = {int(x): j for j, x in enumerate(my_angles)}
        for ai in range(n_angles):           # every rank draws the same stream and keeps its angles
            if ai not in mine and shard == "lambda":
                continue
            shift = 0.15 * sigma * np.cos(2 * np.pi * ai / n_angles)
            psi = 1.0 + 9.0 * torch.exp(-((lam - centre - shift) / sigma) ** 2)
            noise = torch.rand((n, nlam_total), generator=gen, device=dev, dtype=torch.float64)
            if ai in mine:
                alpha[mine[ai]] = (strat[:, None] * (1.0 + 0.1 * noise) * psi[None, :])[:, lam_lo:lam_hi]
            del noise
        alpha_mode = _lib.ALPHA_ANGLE_SITE_LAM
    else:
        psi = 1.0 + 9.0 * torch.exp(-((lam - centre) / sigma) ** 2)
        alpha = (strat[:, None] * (1.0 + 0.1 * torch.rand((n, nlam_total), generator=gen, device=dev,
                                                          dtype=torch.float64)) * psi[None, :])[:, lam_lo:lam_hi]
        alpha = alpha.contiguous()
        alpha_mode = _lib.ALPHA_SITE_LAM
    S = S[:, lam_lo:lam_hi].contiguous()
    n1_up = int(sites.layers_up[1] - 1)
    bottom = torch.as_tensor(sites.perm_up[:n1_up] - 1, device=dev)
    I0_up = S[bottom].contiguous()          # I_0 = S at the bottom layer for up rays; down: zeros
    J = torch.zeros((n, nlam), device=dev, dtype=torch.float64)
    f32 = args.dtype == "f32"
    if f32:
        S, alpha, I0_up, J = (t.to(torch.float32).contiguous() for t in (S, alpha, I0_up, J))
    w_mine = weights[my_angles]
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    alpha_native = None
    if per_angle and args.alpha_layout == "native" and args.angle_groups == 1:
        # one-time layout change outside the timed region: per-angle alpha is produced once per
        # Λ-iteration (lambda_iteration.jl:89-96) -- by vrt_line_opacity_dev directly in this layout
        alpha_native = torch.empty(plan.native_alpha_count(nlam), device=dev, dtype=torch.float32 if f32 else torch.float64)
        plan.alpha_to_native_dev(nlam, nlam, alpha.data_ptr(), alpha_native.data_ptr(), stream=stream, f32=f32)
        torch.cuda.synchronize()

    groups = None
    if args.angle_groups > 1:
        # diagnostics only: per-group plans, J of each group into its own buffer (the final add is
        # not timed); groups = contiguous runs of the angle list sorted up-first
        order = sorted(range(A), key=lambda j: (dirs_all[my_angles[j]] < 0, j))
        chunks = np.array_split(np.array(order), args.angle_groups)
        groups = []
        for ch in chunks:
            ch = np.sort(ch)
            gp = vrt.FormalPlan(sites, k_all[my_angles[ch]], 3, dirs=dirs_all[my_angles[ch]])
            ga = alpha[ch].contiguous() if per_angle else alpha
            groups.append((gp, ga, w_mine[ch], torch.zeros_like(J)))

    # S and J in the sweep's own layout (vrt_plan_execute_native_dev): the device-resident Λ-iteration produces S and
    # consumes J in this form (vrt_lambda_iterate), so its sweep step has no layout change; converted once, untimed
    # (alpha per (site, wavelength) -- the continuum's, fixed over the iterations -- then goes along in sweep order too:
    # VRT_ALPHA_SITE_LAM_NATIVE, both directions' plane sets one behind the other, converted once, untimed)
    shared_native = not per_angle and alpha_mode == _lib.ALPHA_SITE_LAM and args.alpha_layout == "native"
    sj_native = args.sj_layout in ("native", "auto") and (alpha_native is not None or shared_native) \
        and groups is None and not dist_on
    S_nat = J_nat = None
    native_alpha_mode = _lib.ALPHA_ANGLE_NATIVE
    vdt = torch.float32 if f32 else torch.float64
    if sj_native and alpha_native is None:
        cnt = plan.native_plane_count(nlam)
        alpha_native = torch.empty(2 * cnt, device=dev, dtype=vdt)
        plan.to_native_dev(nlam, nlam, alpha.data_ptr(), alpha_native.data_ptr(), alpha_native.data_ptr() + alpha_native.element_size() * cnt,
                           stream=stream, f32=f32)
        native_alpha_mode = _lib.ALPHA_SITE_LAM_NATIVE
    if sj_native:
        cnt = plan.native_plane_count(nlam)
        S_nat = [torch.empty(cnt, device=dev, dtype=vdt) for _ in range(2)]
        J_nat = [torch.zeros(cnt, device=dev, dtype=vdt) for _ in range(2)]
        plan.to_native_dev(nlam, nlam, S.data_ptr(), S_nat[0].data_ptr(), S_nat[1].data_ptr(), stream=stream, f32=f32)
        torch.cuda.synchronize()

    gathered = {}     # lambda-strong: the gathered (n, nlam_total) J of the last step
    coll = {"op": None, "bytes_per_step": 0, "ms": 0.0, "calls": 0}
    coll_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def timed_collective(fn, op, nbytes):
        """the data-path collective of a step, between two events on the launch stream"""
        coll_ev[0].record()
        out_ = fn()
        coll_ev[1].record()
        coll_ev[1].synchronize()
        coll["op"], coll["bytes_per_step"] = op, int(nbytes)
        coll["ms"] += coll_ev[0].elapsed_time(coll_ev[1])
        coll["calls"] += 1
        return out_

    def step():
        if groups is not None:
            for gp, ga, gw, gJ in groups:
                gp.execute_dev(nlam, nlam, S.data_ptr(), ga.data_ptr(), alpha_mode, gw,
                               dJ=gJ.data_ptr(), dI0_up=I0_up.data_ptr(), stream=stream, f32=f32)
            return
        if sj_native:
            plan.execute_native_dev(nlam, S_nat[0].data_ptr(), S_nat[1].data_ptr(), alpha_native.data_ptr(),
                                    native_alpha_mode, w_mine, dJ_up=J_nat[0].data_ptr(), dJ_down=J_nat[1].data_ptr(),
                                    dI0_up=I0_up.data_ptr(), stream=stream, f32=f32)
        elif alpha_native is not None and native_alpha_mode == _lib.ALPHA_ANGLE_NATIVE:
            plan.execute_dev(nlam, nlam, S.data_ptr(), alpha_native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE,
                             w_mine, dJ=J.data_ptr(), dI0_up=I0_up.data_ptr(), stream=stream, f32=f32)
        else:
            plan.execute_dev(nlam, nlam, S.data_ptr(), alpha.data_ptr(), alpha_mode, w_mine,
                             dJ=J.data_ptr(), dI0_up=I0_up.data_ptr(), stream=stream, f32=f32)
        if dist_on and shard == "angle":
            def reduce_():
                if rehearse:       # gloo reduces host tensors
                    Jh = J.cpu()
                    distributed.allreduce_J(Jh)
                    J.copy_(Jh)
                else:
                    distributed.allreduce_J(J)
            timed_collective(reduce_, "all_reduce(sum) of J (n, nlam) over the angle shards", J.numel() * J.element_size())
        elif dist_on and shard == "lambda-strong":
            gathered["buf"], gathered["sizes"] = timed_collective(
                lambda: distributed.allgather_J_blocks(J.cpu() if rehearse else J, nlam_total, out=gathered.get("buf")),
                "all_gather of the wavelength blocks of J (n, nlam_total)", n * nlam_total * J.element_size())

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    coll["ms"], coll["calls"] = 0.0, 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()              # on the stream the library launches on (its internal streams fork from and
    for _ in range(steps):    # join back into it, so the pair brackets every kernel of the steps)
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    step_event_ms = ev0.elapsed_time(ev1) / steps
    caller_layout = None
    if sj_native:
        # J in the caller's layout for the checks below; and the same step over the caller's layout, timed beside it
        plan.J_from_native_dev(nlam, nlam, J_nat[0].data_ptr(), J_nat[1].data_ptr(), J.data_ptr(), stream=stream, f32=f32)
        torch.cuda.synchronize()
    if sj_native and not args.no_caller_layout:
        J_keep = J.clone()
        sweep_keep = plan.last_sweep_timing()
        sj_native = False
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        sj_native = True
        c_ms = e0.elapsed_time(e1) / steps
        caller_layout = {"ms_per_step": c_ms, "value": n * n_angles * nlam / (c_ms * 1e-3), "unit": "cell-updates/s",
                         "bitwise_equal_J": bool(torch.equal(J, J_keep)),
                         "note": "the same step through vrt_plan_execute_dev: S (n, nlam) in, J (n, nlam) out -- two layout "
                                 "changes (k_to_sweep_order, k_combine_J) inside every step"}
        step()                       # (the plan's last step is a native one again)
        torch.cuda.synchronize()
        J.copy_(J_keep)
        del J_keep
    if groups is not None:
        tl = [gp.last_sweep_timing() for gp, _, _, _ in groups]
        sweep_ms, launches = sum(t[0] for t in tl), sum(t[1] for t in tl)
    elif caller_layout is not None:
        sweep_ms, launches = sweep_keep                   # (of the last TIMED step, taken before the caller-layout run)
    else:
        sweep_ms, launches = plan.last_sweep_timing()    # HIP events around the sweep launches, last step
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # cell-updates: one (site, direction, wavelength) intensity incl. its 3 sweeps
    if shard == "lambda":
        updates_per_step = n * n_angles * nlam * world    # weak: every rank adds a λ block
    else:
        updates_per_step = n * n_angles * nlam_total      # strong: the whole fixed job
    ms_per_step = elapsed / steps * 1e3
    value = updates_per_step / (elapsed / steps)

    out = {
        "metric": "formal-solve cell-updates/sec", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak" if shard == "lambda" else "strong",
        "vs_baseline": None, "dtype": "f64", "storage_dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": f"{workload}: jittered-BCC Voronoi grid a={a} c={c} ({n} sites, "
                        f"L_up={len(sites.layers_up) - 1} layers), {quad} ({n_angles} angles), "
                        f"nlam={nlam} on this rank of {nlam * world if shard == 'lambda' else nlam_total} in the job, "
                        f"alpha per {'angle,site,lambda' if per_angle else 'site,lambda'}"
                        f"{' (native storage-pair layout)' if alpha_native is not None else ''}, n_sweeps=3",
            "sites": n, "angles": n_angles, "nlam_per_rank": nlam, "shard": shard,
            "alpha_layout": ("native" if native_alpha_mode == _lib.ALPHA_ANGLE_NATIVE else "sweep order, both directions (VRT_ALPHA_SITE_LAM_NATIVE)")
                            if alpha_native is not None else "caller",
            "sj_layout": "sweep order per direction (vrt_plan_execute_native_dev%s: what vrt_lambda_iterate keeps between its steps)" % ("_f32" if f32 else "")
                         if sj_native else "caller (n, nlam)",
        },
        "setup_s": {"grid_generate": t_gen, "grid_create": t_grid, "plan_create": t_plan},
    }
    if caller_layout is not None:
        out["caller_layout"] = caller_layout
    if dist_on:
        out["collective"] = {"op": coll["op"] or "none (every rank owns whole rows of J)",
                             "bytes_per_step": coll["bytes_per_step"],
                             "ms": coll["ms"] / max(coll["calls"], 1),
                             "backend": "gloo (rehearsal on one GPU)" if rehearse else "nccl (RCCL)",
                             "note": "rank 0's time of the collective inside every timed step (HIP events on the launch stream)"}
    # secondary bound: the dependency critical path of the sweep -- the same plan with
    # ONE wavelength pair, where bandwidth plays no role and only the per-layer launch + Gauss-Seidel
    # level chain of the most inclined angle is left
    floor_ms = None
    if rank == 0 and world == 1 and groups is None and not f32 and nlam > 2 and not args.no_critical_path:
        for _ in range(3):
            plan.execute_dev(2, nlam, S.data_ptr(), alpha.data_ptr(), alpha_mode, w_mine,
                             dJ=J.data_ptr(), dI0_up=I0_up.data_ptr(), stream=stream, f32=False)
        torch.cuda.synchronize()
        floor_ms = plan.last_sweep_timing()[0]
        step()                     # restore J of the full problem (parity check below)
        if sj_native:              # (the native step leaves J in its plane sets)
            plan.J_from_native_dev(nlam, nlam, J_nat[0].data_ptr(), J_nat[1].data_ptr(), J.data_ptr(), stream=stream, f32=f32)
        torch.cuda.synchronize()
    # practical ceiling of this box beside the vendor peak: a device triad b = a + b
    # over 2 x 1 GB (2 reads + 1 write per element), measured live
    triad_gbs = None
    if rank == 0 and world == 1:
        ta_ = torch.rand(1 << 27, device=dev, dtype=torch.float64)
        tb_ = torch.rand(1 << 27, device=dev, dtype=torch.float64)
        torch.add(ta_, tb_, out=tb_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            torch.add(ta_, tb_, out=tb_)
        torch.cuda.synchronize()
        triad_gbs = 10 * 3 * ta_.numel() * 8 / (time.perf_counter() - t0) / 1e9
        del ta_, tb_
    # ---- roofline: algorithmic bytes of the WHOLE step / HIP-event time of the whole step -------
    # read S_c, α_c + write I + read-modify-write J = 40 B (20 B for fp32 values) per
    # cell-update, + the 40-B upwind-table entry per (site, angle) amortised over the wavelengths
    v = 4.0 if f32 else 8.0
    bytes_per_update = 5.0 * v + 40.0 / nlam
    local_updates = n * A * nlam
    alg_bytes = local_updates * bytes_per_update
    achieved = alg_bytes / (step_event_ms * 1e-3) / 1e9
    sweep_bytes = local_updates * (3.0 * v + 40.0 / nlam)        # without the J read-modify-write
    out["roofline"] = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic_per_step(workload, plan.last_path, nlam, world,
                                        "native" if alpha_native is not None else "caller", "native" if sj_native else "caller"),
        "kernel": {"levels": "k_sweep_level (one launch per dependency level)",
                   "steps": "k_step_coeffs + k_step_levels (two launches per BFS layer)",
                   "tiles": "k_sweep_tiles (one persistent launch)",
                   "patches": (("k_patch_chain_quad (fp32 storage: two wavelength pairs per lane; " if f32 and ((nlam + 1) // 2) % 2 == 0
                                else "k_patch_chain (64 registers, four workgroups per CU; ") +
                               "ONE chained launch for every BFS layer of every angle: a workgroup takes the next item "
                               "(patch, angle, block of wavelength pairs) of its XCD queue and waits, pair by pair, for the patches "
                               "whose stored intensities it gathers; J_dir formed by items of the same launch)") if launches == 1
                   else (("k_patch_quad (fp32 storage: two wavelength pairs per lane; " if f32 and ((nlam + 1) // 2) % 2 == 0
                          else "k_patch_lean (64 registers, four workgroups per CU; " if (nlam + 1) // 2 >= 2
                          else "k_patch_solve (") +
                         "one fused launch per BFS layer and direction: coefficients + "
                         "Gauss-Seidel levels of every patch, J reduction of the previous layer riding along)")
                   }.get(plan.last_path, plan.last_path),
        "path": plan.last_path, "launches_per_step": launches,
        "step_event_ms": step_event_ms,
        "algorithmic_bytes_per_step": alg_bytes,
        "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1),
        "avg_launch_us": step_event_ms * 1e3 / max(launches, 1),
        "bytes_per_cell_update": bytes_per_update,
        "triad_ceiling_GBs": triad_gbs,
        "critical_path_ms": floor_ms,
        "sweep_only": {"ms": sweep_ms, "bytes_per_cell_update": 3.0 * v + 40.0 / nlam,
                       "achieved": sweep_bytes / (sweep_ms * 1e-3) / 1e9,
                       "frac": sweep_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "note": "achieved = algorithmic bytes of one whole step (sweep + layout changes + J reduction) / "
                "HIP-event time of the whole step on the launch stream (traffic and achieved are per "
                "step; per launch = / launches_per_step); `traffic` = fabric-side bytes of one step from "
                "the committed rocprofv3 PMC passes, quoted only when that profile was taken with this "
                "exact library; sweep_only = the sweep launches alone, with bytes that exclude the "
                "16 B/update J read-modify-write done outside that window",
    }

    # ---- CPU baseline: the oracle, threaded the way the reference is (angles serial, λ split
    # over threads), on a bounded sample of the same workload; also a full-size parity check ----
    if rank == 0 and world == 1 and cpu_baseline:      # reported at N = 1 only
        from oracle import oracle as orc
        cores = len(os.sched_getaffinity(0))
        t0 = time.time()
        so = orc.make_sites(pos, nbr, bounds)
        t_osites = time.time() - t0
        # (a) the baseline, threaded the way the reference is: angles serial, ALL nlam wavelengths dealt to min(nlam, cores)
        # threads (Threads.@threads over λ, lambda_iteration.jl:91); bounded (~3e8 cell-updates) by taking a SUBSET OF THE
        # ANGLES, ups and downs alternating -- not fewer wavelengths, which would leave the thread team short
        threads = args.cpu_lam if args.cpu_lam > 0 else min(nlam, cores)
        lam_b = nlam if args.cpu_lam <= 0 else min(nlam, args.cpu_lam)
        th_m = theta[my_angles]
        ups_, downs_ = [j for j in range(A) if th_m[j] > 90], [j for j in range(A) if th_m[j] < 90]
        inter = [x for pair in zip(ups_, downs_) for x in pair] + ups_[len(downs_):] + downs_[len(ups_):]
        A_b = max(1, min(A, int(3.0e8 // (n * lam_b))))
        selb = np.array(sorted(inter[:A_b]))
        S_b = S[:, :lam_b].contiguous().cpu().numpy().astype(np.float64)
        al_b = (alpha[selb][:, :, :lam_b] if per_angle else alpha[:, :lam_b]).contiguous().cpu().numpy().astype(np.float64)
        I0_b = I0_up[:, :lam_b].contiguous().cpu().numpy().astype(np.float64)
        t0 = time.time()
        orc.J_voronoi(w_mine[selb], th_m[selb], phi[my_angles][selb], S_b, al_b, so, I0_up=I0_b, nthreads=threads)
        t_cpu = time.time() - t0
        cpu_updates = n * len(selb) * lam_b
        del S_b, al_b
        # (b) the parity sample: all angles x the first wavelengths (<= 1e8 cell-updates), against the J of the timed steps
        lam_s = max(1, min(nlam, int(1.0e8 // (n * A))))
        S_h = S[:, :lam_s].contiguous().cpu().numpy().astype(np.float64)
        if per_angle:
            al_h = alpha[:, :, :lam_s].contiguous().cpu().numpy().astype(np.float64)
        else:
            al_h = alpha[:, :lam_s].contiguous().cpu().numpy().astype(np.float64)
        I0_h = I0_up[:, :lam_s].contiguous().cpu().numpy().astype(np.float64)
        J_ref = orc.J_voronoi(w_mine, theta[my_angles], phi[my_angles], S_h, al_h, so, I0_up=I0_h,
                              nthreads=min(cores, lam_s))
        # T = 1: one wavelength, one thread (T = 1 and T = all cores); all angles, or
        # one up + one down ray when that alone would take more than ~30 s
        sel = np.arange(A)
        if n * A > 3.0e7:
            th_m = theta[my_angles]
            sel = np.array([int(np.argmax(th_m > 90)), int(np.argmax(th_m < 90))])
        al1 = np.ascontiguousarray(al_h[sel][:, :, :1] if per_angle else al_h[:, :1])
        t0 = time.time()
        orc.J_voronoi(w_mine[sel], theta[my_angles][sel], phi[my_angles][sel], np.ascontiguousarray(S_h[:, :1]), al1,
                      so, I0_up=np.ascontiguousarray(I0_h[:, :1]), nthreads=1)
        t_cpu1 = time.time() - t0
        J_gpu = J[:, :lam_s].cpu().numpy().astype(np.float64)
        from oracle.parity import rel as _parity_rel
        _pe = _parity_rel(J_gpu, J_ref)            # element-wise: |a - b| / (|b| + smallest non-zero |b|), its maximum
        parity = float(_pe)
        parity_maxnorm = float(_pe.maxnorm)
        out["cpu_baseline"] = {
            "value": cpu_updates / t_cpu, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"same grid and fields, {len(selb)} of the {A} angles (ups and downs alternating) x all {lam_b} "
                      f"wavelengths on {threads} threads = min(nlam, host cores), one wavelength each like "
                      f"Threads.@threads ({cpu_updates} cell-updates in {t_cpu:.1f} s wall); oracle restatement; its grid "
                      f"prep (read_cell equivalent) took {t_osites:.1f} s and is not counted; parity: all {A} angles x the "
                      f"first {lam_s} wavelengths against the J of the timed steps",
            "seconds": t_cpu,
            "single_thread": {"value": n * len(sel) / t_cpu1, "unit": "cell-updates/s", "cores": 1, "seconds": t_cpu1,
                              "sample": f"{len(sel)} of {A} angles x 1 wavelength ({n * len(sel)} cell-updates)"},
            "cpu_model": cpu_model(), "host_cores": cores,
        }
        out["parity_vs_oracle_max_rel_err"] = parity                   # element-wise (every |a - b| / (|b| + floor))
        out["parity_vs_oracle_max_norm_err"] = parity_maxnorm
    if args.dump_J and rank == 0:
        Jout = distributed.assemble_J_blocks(gathered["buf"], gathered["sizes"]) if "buf" in gathered else J
        np.save(args.dump_J, Jout.cpu().numpy())
    return out


def measure_c1(torch, vrt):
    """BASELINE configs[0]: compare_searchlight.jl's regular-grid searchlight, 60^3, n1.dat (one
    vertical up ray), through the regular-grid HIP solver; CPU oracle beside it."""
    from oracle import oracle as orc
    w, th, ph, _ = vrt.read_quadrature("n1.dat")
    n = 60
    z = x = y = np.linspace(0, 1, n)
    S = np.zeros((n, n, n))
    al = np.zeros((n, n, n))
    I0 = np.zeros((n, n))
    ii, jj = np.meshgrid(np.arange(1, n + 1), np.arange(1, n + 1), indexing="ij")
    I0[(np.sqrt((ii / n - 0.5) ** 2 + (jj / n - 0.5) ** 2) < 0.1).T] = 1.0     # compare_searchlight.jl:180-190
    dev = torch.device("cuda", torch.cuda.current_device())
    solver = vrt.RegularSolver(z, x, y, device=dev.index)
    Sd, Ad, I0d = (torch.from_numpy(a).to(dev) for a in (S, al, I0[None]))
    out = torch.empty((1, n, n, n), device=dev, dtype=torch.float64)
    k = vrt.direction(th[0], ph[0])[None]
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        solver.execute_dev(k, [True], Sd.data_ptr(), 0, Ad.data_ptr(), 0, I0d.data_ptr(), out.data_ptr(), 3, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        solver.execute_dev(k, [True], Sd.data_ptr(), 0, Ad.data_ptr(), 0, I0d.data_ptr(), out.data_ptr(), 3, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.time()
    ref = orc.short_characteristics_up(orc.direction(th[0], ph[0]), S, I0, al, z, x, y, 3)
    t_cpu = time.time() - t0
    got = out[0].cpu().numpy()
    solver.close()
    return {"config": {"workload": "C1: regular 60^3 searchlight, n1.dat (theta 180, phi 0), alpha = S = 0, n_sweeps=3"},
            "ms_per_step": ms, "value": n ** 3 / (ms * 1e-3), "unit": "cell-updates/s",
            "cpu_baseline": {"value": n ** 3 / t_cpu, "unit": "cell-updates/s", "cores": 1, "kind": "port"},
            "parity_vs_oracle_max_abs_err": float(np.abs(got - ref).max()),
            "top_plane_equals_I0": bool(np.array_equal(got[1:-1, 1:-1, -1], I0[1:-1, 1:-1]))}


def launch_own_ranks(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, before
    this process has touched a GPU, and exit with the job's code (rank 0 prints the JSON line)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse_args()
    # the host driver supports dmabuf IPC only: RCCL and cross-process tensor sharing need this
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (the sweep advances on the caller's stream and ONE internal stream: the default of four hardware queues
    # per process leaves room for RCCL's -- no GPU_MAX_HW_QUEUES override is needed any more)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_own_ranks(args))

    import torch
    import torch.distributed as dist

    import voronoirt_amd as vrt
    from voronoirt_amd import _lib, distributed, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # VRT_BENCH_REHEARSE=1: every rank uses GPU 0 and the gloo backend (single-GPU rehearsal of
    # the N > 1 code path; RCCL refuses two ranks on one device)
    rehearse = os.environ.get("VRT_BENCH_REHEARSE") == "1"
    # VRT_BENCH_FORCE_DIST=1: a process group also for ONE rank (nccl = RCCL unless rehearsing), so that the collectives of the
    # N > 1 modes execute on a one-GPU box (tests/test_physics.py)
    force_dist = os.environ.get("VRT_BENCH_FORCE_DIST") == "1"
    rank, world = distributed.init_process_group("gloo" if rehearse else None, force=force_dist)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    out = measure(args, args.workload, torch, dist, vrt, _lib, distributed, synth, rank, world, local, dev,
                  rehearse, args.steps, args.warmup, not args.no_cpu_baseline)
    # BASELINE.json configs[1] (the reference's continuum case: ~250k sites x 12 angles x 1 λ) rides
    # along as a secondary measurement of the default single-GPU run; the headline stays the
    # configuration the metric is quoted on (1M sites, 12 angles).
    if world == 1 and args.workload == "C4" and args.nlam == 0 and args.dtype == "f64" \
            and args.angle_groups == 1 and not args.no_secondary:
        sec = measure(args, "C2", torch, dist, vrt, _lib, distributed, synth, rank, world, local, dev,
                      rehearse, 20, 3, not args.no_cpu_baseline)
        out["other_configs"] = {"C2": {k: sec[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup",
                                                            "config", "roofline", "cpu_baseline",
                                                            "parity_vs_oracle_max_rel_err") if k in sec}}
        if not args.no_cpu_baseline:
            out["other_configs"]["C1"] = measure_c1(torch, vrt)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
