#!/bin/bash
# diagnostics (-DVRT_DIAG build, WRONG results): C4 step of the chained launch with the hand-off protocol's pieces
# switched off: 512 plain gathers of I, 1024 plain stores of I, 256 no waits, 128 no J reduction
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
run() { label=$1; shift
  line=$(env "$@" python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path 2>>gpurun_out/chain_flags_err.log | tail -1)
  echo "$label $(echo "$line" | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f launches %d" % (j["ms_per_step"], r["sweep_only"]["ms"], r["launches_per_step"]))')"
}
for f in 0 512 1024 1536 1792 1920; do run "diag chain flags=$f" VRT_PATCH_CHAIN=1 VRT_DEBUG_FLAGS=$f; done
run "diag launches flags=0" VRT_PATCH_CHAIN=0 VRT_DEBUG_FLAGS=0
