#!/usr/bin/env python3
"""Copies the judged summaries of a tools/make_profiles.sh run from gpurun_out/ into profiles/<round>/
(kernel-trace stats csv, bench JSON lines, and the per-kernel PMC sums inside c4_summary.json).

    python tools/summarise_profiles.py <tag> <round-dir>     e.g.  r2 r2

c4_summary.json carries the digest of the kernel sources the profile was taken with
(bench.source_digest): bench.py quotes `roofline.traffic` from it only while the library is that
exact one.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (source_digest, ONE_TIME_KERNELS)

go, out = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(out, exist_ok=True)

stats_csv = glob.glob(os.path.join(go, f"prof_{tag}_default", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats_csv, os.path.join(out, "c4_default_kernel_stats.csv"))
steps_total = 4    # tools/prof_kernels.sh: 1 warm-up + 3 timed steps under the profiler
ks = {}
for r in csv.DictReader(open(stats_csv)):
    if r["Name"].startswith(("vrt::", "void vrt::")):
        name = r["Name"].split("(")[0]
        once = any(k in name for k in bench.ONE_TIME_KERNELS)
        ks[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                    "ms_per_step": float(r["TotalDurationNs"]) / 1e6 / (1 if once else steps_total)}
# launches of the two sweep directions run concurrently on two streams: beside the SUM of a kernel's durations
# (ms_per_step) give the wall time its dispatches cover (union of their intervals) and the queues they ran on
trace = glob.glob(os.path.join(go, f"prof_{tag}_default", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    spans = collections.defaultdict(list)
    queues = collections.defaultdict(set)
    for r in csv.DictReader(open(trace[0])):
        if r["Kernel_Name"].startswith(("vrt::", "void vrt::")):
            name = r["Kernel_Name"].split("(")[0]
            spans[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            queues[name].add(r["Queue_Id"])
    for name, iv in spans.items():
        if name not in ks:
            continue
        iv.sort()
        covered, cur_lo, cur_hi = 0, iv[0][0], iv[0][1]
        for lo, hi in iv[1:]:
            if lo > cur_hi:
                covered += cur_hi - cur_lo
                cur_lo, cur_hi = lo, hi
            else:
                cur_hi = max(cur_hi, hi)
        covered += cur_hi - cur_lo
        once = any(k in name for k in bench.ONE_TIME_KERNELS)
        ks[name]["covered_wall_ms_per_step"] = covered / 1e6 / (1 if once else steps_total)
        ks[name]["queues"] = len(queues[name])
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_EA0_RDREQ_DRAM_sum"):
    for f in glob.glob(os.path.join(go, f"pmc_{tag}_default_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("vrt::", "void vrt::")):
                pmc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
bench_line = [ln for ln in open(os.path.join(go, f"{tag}_bench_default.json")).read().splitlines() if ln.startswith("{")][-1]
b = json.loads(bench_line)
# with alpha handed over in the native layout its one-time layout change (one k_to_sweep_order launch per
# angle, before the timed region) is in the profiled process too: only the per-step launches of that kernel
# (S of the two directions; every launch moves the same (n, nlam) plane) count towards a step
n_angles = b["config"]["angles"]
for k, c in pmc.items():
    if "k_to_sweep_order" in k and b["config"].get("alpha_layout") == "native":
        calls = ks.get(k, {}).get("calls", 0)
        per_run = n_angles + 2                      # one profiled step: n_angles one-time + 2 per-step launches
        for name in list(c):
            c[name] *= 2.0 / per_run
        c["note_scaled_to_per_step_launches"] = 1.0
sj_native = str(b["config"].get("sj_layout", "")).startswith("sweep order")
if sj_native:
    # S and J stay in sweep order: every layout-change launch of the profiled process is set-up (alpha and S once, J for
    # the parity check after the timed steps), none belongs to a step
    for k in list(pmc):
        if "k_to_sweep_order" in k or "k_combine_J" in k or "k_from_sweep_order" in k:
            pmc[k]["note_not_part_of_a_step"] = 1.0
per_step = {k: v for k, v in pmc.items() if not any(o in k for o in bench.ONE_TIME_KERNELS) and "note_not_part_of_a_step" not in v}
total = sum((2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 for c in per_step.values())
summary = {
    "note": "rocprofv3 on MI355X, `python bench.py` default command (C4: 995566 sites x 12 angles x 51 wavelengths, "
            "fp64, per-angle alpha in the native layout, S and J in sweep order: vrt_plan_execute_native_dev).  kernel_stats: --kernel-trace --stats over 1 warm-up + 3 "
            "timed steps (tools/prof_kernels.sh); one-time kernels (tables, layout change of alpha) are plan "
            "creation / set-up, not part of a step.  pmc_one_step: separate --pmc passes of ONE step "
            "(tools/prof_pmc.sh), sums over all dispatches of a kernel, FETCH_SIZE / WRITE_SIZE in KiB; per "
            "MI355X_MICROARCH.md FETCH_SIZE tallies 128-B requests at 64 B on gfx950, so bytes = 2 x FETCH_SIZE + "
            "WRITE_SIZE.  The sweep runs on two internal streams, so launches of the two directions overlap and "
            "their summed durations (kernel_stats.*.ms_per_step) exceed the wall time of the sweep: "
            "kernel_stats.*.covered_wall_ms_per_step is the union of a kernel's dispatch intervals per step and "
            "`queues` the number of hardware queues they ran on.",
    "source_digest": bench.source_digest(),
    "alpha_layout": b["config"].get("alpha_layout"),
    "sj_layout": "native" if sj_native else "caller",
    "path": b["roofline"]["path"],
    "bench": {"ms_per_step": b["ms_per_step"], "step_event_ms": b["roofline"]["step_event_ms"],
              "sweep_ms": b["roofline"]["sweep_only"]["ms"], "frac": b["roofline"]["frac"]},
    "traffic_bytes_per_step": total,
    "algorithmic_bytes_per_step": b["roofline"]["algorithmic_bytes_per_step"],
    "kernel_stats": ks,
    "pmc_one_step": {k: dict(v) for k, v in pmc.items()},
}
json.dump(summary, open(os.path.join(out, "c4_summary.json"), "w"), indent=1)
for src, dst in ((f"{tag}_bench_default.json", "bench_default.json"), (f"{tag}_bench_c3.json", "bench_c3_default.json"),
                 (f"{tag}_bench_c5_f32.json", "bench_c5_f32.json")):
    path = os.path.join(go, src)
    if os.path.exists(path):
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(out, dst), "w").write(lines[-1] + "\n")
for src, dst in ((f"prof_{tag}_real1m", "real_grid_1m_kernel_stats.csv"), (f"prof_{tag}_C3", "c3_default_kernel_stats.csv"),
                 (f"prof_{tag}_C5", "c5_f32_kernel_stats.csv")):
    found = glob.glob(os.path.join(go, src, "**", "*kernel_stats.csv"), recursive=True)
    if found:
        shutil.copy(found[0], os.path.join(out, dst))
for src, dst in ((f"{tag}_real_grid_1m.txt", "real_grid_1m.txt"), (f"{tag}_pmc.txt", "c4_pmc_table.txt")):
    if os.path.exists(os.path.join(go, src)):
        shutil.copy(os.path.join(go, src), os.path.join(out, dst))
print(json.dumps(ks, indent=1)[:1500])
print("traffic per step %.1f GB, algorithmic %.1f GB" % (total / 1e9, b["roofline"]["algorithmic_bytes_per_step"] / 1e9))
