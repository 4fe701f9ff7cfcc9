#!/bin/bash
# diagnostics: per-kernel durations (rocprofv3 --kernel-trace --stats) of one bench configuration
# usage: tools/prof_kernels.sh <tag> [bench args...]; env (VRT_*) is inherited
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout "$@" > $out/bench.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print('%-40s calls %6s avg_us %10.2f total_ms %9.3f'%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
