#!/usr/bin/env python3
"""Copies the judged summaries of a tools/make_profiles.sh run from gpurun_out/ into profiles/<round>/
(kernel-trace stats csv, bench JSON lines, and the per-kernel PMC sums inside c4_summary.json).

    python tools/summarise_profiles.py <tag> <round-dir>     e.g.  r1b r1
"""
import collections, csv, glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, out = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(out, exist_ok=True)

stats_csv = glob.glob(os.path.join(go, f"prof_{tag}_default", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats_csv, os.path.join(out, "c4_steps_default_kernel_stats.csv"))
bench_log = open(os.path.join(go, f"prof_{tag}_default", "bench.log")).read().splitlines()
steps_total = 7    # tools/make_profiles.sh: 2 warm-up + 5 timed steps under the profiler
ks = {}
for r in csv.DictReader(open(stats_csv)):
    if r["Name"].startswith(("vrt::", "void vrt::")):
        name = r["Name"].split("(")[0]
        once = any(k in name for k in ("k_upwind_table", "k_permute_table", "k_delaunay_lines", "k_sorted_tables", "k_gpos"))
        ks[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                    "ms_per_step": float(r["TotalDurationNs"]) / 1e6 / (1 if once else steps_total)}
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(os.path.join(go, f"pmc_{tag}_default_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("vrt::", "void vrt::")):
                pmc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
summary_path = os.path.join(out, "c4_summary.json")
summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {"kernel_stats": {}, "pmc_one_step": {}}
summary["kernel_stats"]["steps_default"] = ks
summary["pmc_one_step"]["steps_default"] = {k: dict(v) for k, v in pmc.items()}
summary["note_steps_default"] = (
    "steps_default re-profiled with tools/make_profiles.sh after the wavelength-pair layout, the sorted "
    "thread assignment of k_step_levels and the compact coupling list (kernel_stats: 2 warm-up + 5 timed "
    "steps, two internal streams, so launches of the two angle groups overlap and the summed durations "
    "exceed the sweep window; pmc: ONE step, separate passes per counter, FETCH_SIZE/WRITE_SIZE in KiB).  "
    "The levels and tiles sections are the earlier profiles of those (unchanged) kernels.")
json.dump(summary, open(summary_path, "w"), indent=1)
for src, dst in ((f"{tag}_bench_default.json", "bench_default.json"), (f"{tag}_bench_c3.json", "bench_c3_default.json")):
    line = [l for l in open(os.path.join(go, src)).read().splitlines() if l.startswith("{")][-1]
    open(os.path.join(out, dst), "w").write(line + "\n")
print(json.dumps(ks, indent=1)[:1500])
