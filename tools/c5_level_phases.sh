#!/bin/bash
# diagnostics: where the single-wavelength level kernel spends its time on C5 (fp32 values): per-kernel
# durations (rocprofv3 --kernel-trace --stats) with pieces switched off (WRONG results; -DVRT_DIAG build).
#   8 no coefficient loads, 16 no I stores, 256 no permutation, 512 levels polled without visits,
#   1024 I stored contiguously; flags:1 = no level loop at all.   SPECS="0:0 1024:0" overrides the list
python -m voronoirt_amd.build --diag > /dev/null || exit 1
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
export VRT_STEP_STREAMS=${STREAMS:-1}
for spec in ${SPECS:-0:0 8:0 16:0 256:0 512:0 0:1 280:1}; do
  f=${spec%%:*}; sk=${spec##*:}
  if [ "$sk" = 1 ]; then export VRT_DEBUG_SKIP_LEVELS=1; else unset VRT_DEBUG_SKIP_LEVELS; fi
  echo "== flags $f skip_levels $sk"
  VRT_DEBUG_FLAGS=$f bash tools/prof_kernels.sh c5ph_${f}_$sk --workload ${WORKLOAD:-C5} --dtype ${DTYPE:-f32} --steps 2 --warmup 1 | grep -E "k_step_levels|k_step_coeffs"
done
