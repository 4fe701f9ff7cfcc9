#!/bin/bash
# diagnostics: arbitrary PMC counters of ONE C4 bench step, summed per kernel (separate rocprofv3 --pmc pass, no tracing
# alongside).  usage: tools/prof_counters.sh <tag> "<COUNTER ...>" [bench args...]; env (VRT_*) is inherited
tag=$1; ctr=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/ctr_${tag}
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out -o p -- python3 bench.py --workload C4 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-critical-path "$@" > $out/bench.log 2>&1 || { echo "pass failed: $ctr"; tail -5 $out/bench.log; exit 1; }
python3 - $out <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name'].split('(')[0][:44]][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(tot.items(), key=lambda kv: -sum(kv[1].values()))[:3]:
    print('%-44s ' % k + '  '.join('%s %.4g' % (n, v) for n, v in sorted(c.items())))
PY
