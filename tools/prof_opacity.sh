#!/bin/bash
# diagnostics: SQ counters of k_line_opacity over one Λ-iteration breakdown run (separate rocprofv3 --pmc passes)
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
out=gpurun_out/sq_opacity_$(echo $pass | cut -c1-12 | tr ' ' _)
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $out -o p -- python3 tools/iteration_breakdown.py --reps 1 > $out/run.log 2>&1 || { echo "pass failed: $pass"; tail -3 $out/run.log; }
python3 - $out <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:44]
        if 'k_line_opacity' not in k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, c in tot.items():
    print(k, 'dispatches', max(v for (kk, _), v in cnt.items() if kk == k))
    for n, v in sorted(c.items()): print('   %-26s %.5g' % (n, v))
PY
done
