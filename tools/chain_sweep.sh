#!/bin/bash
# diagnostics: C4 step time of the chained patch launch over pairs per item, against the per-layer launches, and
# (-DVRT_DIAG build, WRONG results) with pieces of the chain switched off: 256 no waits, 512 plain gathers of I, 1024 plain stores of I
run() { # label, bench args, env...
  label=$1; args=$2; shift 2
  line=$(env "$@" python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path 2>>gpurun_out/chain_sweep_err.log | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f launches %d" % (j["ms_per_step"], r["sweep_only"]["ms"], r["launches_per_step"]))')"
}
run "C4 launches" "" VRT_PATCH_CHAIN=0
for P in 2 3 5 7 9 13 26; do run "C4 chain pairs=$P" "" VRT_CHAIN_PAIRS=$P; done
for L in 7 13 26; do run "nlam=$L launches" "--nlam $L" VRT_PATCH_CHAIN=0; run "nlam=$L chain" "--nlam $L" VRT_PATCH_CHAIN=1; run "nlam=$L chain pairs=2" "--nlam $L" VRT_CHAIN_PAIRS=2; done
run "C3 launches" "--workload C3" VRT_PATCH_CHAIN=0
run "C3 chain" "--workload C3" VRT_PATCH_CHAIN=1
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
for f in 0 256 512 1024 768 1792 1; do run "diag flags=$f" "" VRT_DEBUG_FLAGS=$f; done
