#!/bin/bash
# diagnostics: time the steps path with pieces of its memory traffic switched off (WRONG results):
# needs the -DVRT_DIAG build of the library, which only this script loads
python -m voronoirt_amd.build --diag > /dev/null || exit 1
export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
for f in 0 1 2 3 4 7 8 16 24 31; do
  VRT_STEP_STREAMS=${STREAMS:-1} VRT_DEBUG_FLAGS=$f python bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags',$f,'ms_per_step %.2f sweep_ms %.2f'%(d['ms_per_step'],d['roofline']['sweep_ms_per_step']))"
done
