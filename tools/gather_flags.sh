#!/bin/bash
# diagnostics (WRONG results; -DVRT_DIAG build): C4 step TIME with the upwind gathers of the patch kernel switched off
# VRT_DEBUG_FLAGS: 16 / 32 / 64 no upwind gathers of I / alpha / S (read the own position / the zero site instead), 2 none of them
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
for f in ${@:-0 16 32 64 48 112 2 0}; do
  line=$(VRT_DEBUG_FLAGS=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path 2>/dev/null | tail -1)
  echo "flags $f $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f" % (j["ms_per_step"], r["sweep_only"]["ms"]))')"
done
