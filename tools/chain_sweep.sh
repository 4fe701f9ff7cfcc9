#!/bin/bash
# diagnostics: C4 step time of the chained patch launch over pairs per item, against the per-layer launches, and
# (-DVRT_DIAG build, WRONG results) with pieces of the chain switched off: 128 no J reduction, 256 no waits, 512 plain
# gathers of I, 1024 plain stores of I, 2048 no pair loop (the items' overhead alone)
run() { # label, bench args, env...
  label=$1; args=$2; shift 2
  line=$(env "$@" python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path 2>>gpurun_out/chain_sweep_err.log | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f launches %d" % (j["ms_per_step"], r["sweep_only"]["ms"], r["launches_per_step"]))')"
}
if [ "${1:-all}" != diag ]; then
run "C4 launches" "" VRT_PATCH_CHAIN=0
for P in 3 5 7 9; do run "C4 chain pairs<=$P" "" VRT_CHAIN_PAIRS=$P; done
for L in 1 7 13 26; do run "nlam=$L launches" "--nlam $L" VRT_PATCH_CHAIN=0; run "nlam=$L chain" "--nlam $L" VRT_PATCH_CHAIN=1; done
run "C3 launches" "--workload C3" VRT_PATCH_CHAIN=0
run "C3 chain" "--workload C3" VRT_PATCH_CHAIN=1
run "C2 launches" "--workload C2" VRT_PATCH_CHAIN=0
run "C2 chain" "--workload C2" VRT_PATCH_CHAIN=1
fi
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
for f in 0 128 256 2048 2176; do run "diag chain flags=$f" "" VRT_DEBUG_FLAGS=$f; done
for f in 0 128; do run "diag launches flags=$f" "" VRT_DEBUG_FLAGS=$f VRT_PATCH_CHAIN=0; done
