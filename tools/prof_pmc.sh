#!/bin/bash
# diagnostics: per-kernel PMC sums (separate rocprofv3 --pmc passes, no tracing alongside) of ONE
# bench step.  usage: tools/prof_pmc.sh <tag> [bench args...]; env (VRT_*) is inherited
tag=$1; shift
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  name=$(echo $pass | cut -d' ' -f1)
  out=gpurun_out/pmc_${tag}_$name
  rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $out -o p -- python3 bench.py --workload C4 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout "$@" > $out/bench.log 2>&1 || { echo "pass $name failed"; tail -3 $out/bench.log; if [ "$pass" = "${PMC_EXTRA:-}" ]; then continue; fi; exit 1; }
  echo "pass $name done"
done
python3 - gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}_TCC_HIT_sum gpurun_out/pmc_${tag}_TCC_EA0_RDREQ_DRAM_sum <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:36]
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] in ('FETCH_SIZE',): calls[k] += 1
print('%-36s %6s %12s %12s %8s %14s %14s' % ('kernel', 'calls', '2xFETCH GB', 'WRITE GB', 'L2 hit', 'DRAM rd GB', 'DRAM wr GB'))
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', 0))[:8]:
    hit = c.get('TCC_HIT_sum', 0); miss = c.get('TCC_MISS_sum', 0)
    # DRAM-side request counters (where the build of rocprofv3 offers them): 64 B per read request is assumed,
    # 32 / 64 B writes are not told apart -- an upper bound of the HBM share of the fabric bytes
    print('%-36s %6d %12.2f %12.2f %8.3f %14.2f %14.2f' % (k, calls[k], 2 * c.get('FETCH_SIZE', 0) * 1024 / 1e9,
          c.get('WRITE_SIZE', 0) * 1024 / 1e9, hit / max(hit + miss, 1),
          c.get('TCC_EA0_RDREQ_DRAM_sum', float('nan')) * 64 / 1e9, c.get('TCC_EA0_WRREQ_DRAM_sum', float('nan')) * 64 / 1e9))
PY
