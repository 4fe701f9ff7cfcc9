#!/bin/bash
# A/B of two builds in one box over the bench workloads: LIBS="libvrt_hip.so libvrt_hip_b.so" tools/ab_all.sh
run() { lib=$1; shift; VRT_LIB_PATH=voronoirt_amd/$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s %-40s ms/step %8.3f sweep %8.3f' % ('$lib', '$*', d['ms_per_step'], d['roofline']['sweep_only']['ms']))"; }
for rep in 1 2; do for lib in ${LIBS:-libvrt_hip.so libvrt_hip_b.so}; do run $lib --steps 20 --warmup 3; done; done
for lib in ${LIBS:-libvrt_hip.so libvrt_hip_b.so}; do
  run $lib --workload C3 --steps 20 --warmup 3
  run $lib --nlam 7 --steps 30 --warmup 3
  run $lib --workload C2 --steps 50 --warmup 5
  [ -n "$AB_C5" ] && run $lib --workload C5 --dtype f32 --steps 3 --warmup 1
done
