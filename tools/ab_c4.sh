#!/bin/bash
# diagnostics: C4 step time of several builds of the library in one box: LIBS="a.so b.so" tools/ab_c4.sh [bench args]
for rep in 1 2; do
for lib in ${LIBS:-libvrt_hip.so libvrt_hip_b.so}; do
  VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3))"
done; done
