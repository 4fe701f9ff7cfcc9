#!/bin/bash
# diagnostics (WRONG results; -DVRT_DIAG build): C4 step time of the patch kernel with pieces switched off
# VRT_DEBUG_FLAGS: 1 no levels, 2 gathers -> coalesced centre reads, 4 no weights arithmetic, 8 no stores
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so VRT_PATH=patches VRT_PATCH_K=1 VRT_PATCH_Q=1 VRT_PATCH_NT=512 VRT_PATCH_TARGET=768
for f in 0 1 2 4 8 3 5 6 7 15; do
  line=$(VRT_DEBUG_FLAGS=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path 2>/dev/null | tail -1)
  echo "flags $f $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f" % (j["ms_per_step"], r["sweep_only"]["ms"]))')"
done
