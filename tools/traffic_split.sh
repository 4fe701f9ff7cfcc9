#!/bin/bash
# diagnostics (WRONG results; -DVRT_DIAG build): fabric read bytes of ONE C4 step with pieces of the patch kernel's
# traffic switched off.  VRT_DEBUG_FLAGS: 16 / 32 / 64 no upwind gathers of I / alpha / S, 2 none of them, 128 no
# J reduction.  usage: tools/traffic_split.sh [flags...]
export TMPDIR=/tmp
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
for f in ${@:-0 16 32 64 2 128}; do
  out=gpurun_out/split_$f
  rm -rf $out; mkdir -p $out
  VRT_DEBUG_FLAGS=$f timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o p -- python3 bench.py --workload C4 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-critical-path > $out/bench.log 2>&1 || { echo "flags $f failed"; tail -3 $out/bench.log; exit 1; }
  python3 - $out $f <<'PY'
import csv, sys, glob, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            tot[r['Kernel_Name'].split('(')[0][:40]] += float(r['Counter_Value'])
print('flags %4s ' % sys.argv[2] + '  '.join('%s %.2f GB' % (k, 2 * v * 1024 / 1e9) for k, v in tot.most_common(3)))
PY
done
