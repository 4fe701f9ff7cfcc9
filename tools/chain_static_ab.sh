#!/bin/bash
# A/B in one box: chained launch with tickets (libvrt_hip.so) against static block -> item assignment (libvrt_hip_b.so, -DVRT_CHAIN_STATIC)
run() { lib=$1; shift; VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --sj-layout caller "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$*', 'ms/step', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3), 'launches', d['roofline']['launches_per_step'])"; }
for rep in 1 2; do for lib in libvrt_hip.so libvrt_hip_b.so; do
  run $lib --workload C2 --steps 50 --warmup 5
  run $lib --nlam 7 --steps 30 --warmup 3
  run $lib --nlam 1 --steps 30 --warmup 3
done; done
for lib in libvrt_hip.so libvrt_hip_b.so; do VRT_PATCH_CHAIN=1 run $lib --steps 10 --warmup 2; done
