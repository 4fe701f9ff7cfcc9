#!/usr/bin/env python3
"""diagnostics: the kernels of the LAST timed step of a rocprofv3 --kernel-trace of bench.py, in start order with their
queue, start offset, duration and the gap to the previous kernel's end.  usage: python tools/step_timeline.py <dir with *kernel_trace.csv> [anchor kernel substring]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_combine_J"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(ends) < 2:
    sys.exit("anchor kernel not found twice")
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
names = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0][-44:]
    names.setdefault(nm, [0, 0.0])
    names[nm][0] += 1
    names[nm][1] += (e - s) / 1e3
    if hi - lo <= 60:
        print("%-44s q%-2s start %9.1f us  dur %8.1f  gap %7.1f" % (nm, r["Queue_Id"], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("step: %.1f us from the first start to the last end, %d kernels" % ((prev_end - t0) / 1e3, hi - lo))
for nm, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s x%-4d %9.1f us in all" % (nm, c, d))
