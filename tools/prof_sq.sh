#!/bin/bash
# diagnostics: SQ wave-state counters of the dominant kernels of ONE bench step (separate rocprofv3 --pmc pass,
# no tracing alongside).  usage: tools/prof_sq.sh <tag> [bench args...]; env (VRT_*) is inherited
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/sq_${tag}
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out -o p -- python3 bench.py --workload C4 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-critical-path "$@" > $out/bench.log 2>&1 || { echo "pass failed"; tail -3 $out/bench.log; exit 1; }
python3 - $out <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:44]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
names = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS']
print('%-44s ' % 'kernel' + ' '.join('%12s' % n[3:] for n in names) + '  wait/wave  valu/wave')
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:6]:
    wc = max(c.get('SQ_WAVE_CYCLES', 0), 1)
    print('%-44s ' % k + ' '.join('%12.4g' % c.get(n, 0) for n in names) + '  %8.3f  %8.3f' % (c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc))
PY
