#!/bin/bash
# diagnostics: A/B of two builds of the library in one box (C4 twice, C3, C2): LIBS="libvrt_hip.so libvrt_hip_b.so" bash tools/ab_libs.sh
for rep in 1 2; do
for lib in ${LIBS:-libvrt_hip.so libvrt_hip_b.so}; do
  VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'C4', round(d['ms_per_step'],3))"
done; done
for lib in ${LIBS:-libvrt_hip.so libvrt_hip_b.so}; do
  VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 200 python3 bench.py --workload C3 --no-cpu-baseline --no-secondary --no-critical-path --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'C3', round(d['ms_per_step'],3))"
  VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 200 python3 bench.py --workload C2 --no-cpu-baseline --no-secondary --no-critical-path --steps 50 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'C2', round(d['ms_per_step'],3))"
done
