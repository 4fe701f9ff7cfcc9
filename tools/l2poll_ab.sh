#!/bin/bash
# C2 / one pair / two pairs on the chained data-as-flag launch (tools for A/B of two builds of the library: VRT_LIB_PATH)
run() { label=$1; args=$2; shift 2
  line=$(env "$@" timeout -k 10 300 python bench.py $args --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-critical-path 2>>gpurun_out/l2poll_err.log | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.4f sweep_ms %.4f launches %d" % (j["ms_per_step"], r["sweep_only"]["ms"], r["launches_per_step"]))')"
}
for rep in 1 2; do
run "C2" "--workload C2" X=1
run "1M nlam=1" "--nlam 1" X=1
run "1M nlam=3" "--nlam 3" X=1
done
