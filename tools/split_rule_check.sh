#!/bin/bash
# the default split of an item's pair steps (balanced, chosen per launch: launch_patch_layer) against the fixed 768 workgroups per launch of rounds 3-5
run() { e=$1; shift; env $e timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %-46s ms/step %8.3f' % ('$e', '$*', d['ms_per_step']))"; }
for e in VRT_PATCH_TARGET=0 VRT_PATCH_TARGET=768; do
  for nl in 16 20 26 36 51 70 100; do VRT_PATCH_CHAIN=0 run $e --nlam $nl --steps 15 --warmup 3; done
  run $e --workload C3 --steps 20 --warmup 3
  run $e --workload C5 --dtype f32 --steps 3 --warmup 1
  env $e REAL_GRID_DEFAULT_ONLY=1 python3 tools/real_grid_check.py 1000000 24 2>/dev/null | grep -i "ms per J" | head -1
done
