#!/bin/bash
# A/B in one box: storage order of the layers = strips of 20 lattice columns (default) against the Morton curve of rounds 1-4
run() { e=$1; shift; env $e timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %-44s ms/step %8.3f sweep %8.3f' % ('$e', '$*', d['ms_per_step'], d['roofline']['sweep_only']['ms']))"; }
for rep in 1 2; do for e in VRT_STORE_ORDER=strips VRT_STORE_ORDER=morton; do run $e --steps 20 --warmup 3; done; done
for e in VRT_STORE_ORDER=strips VRT_STORE_ORDER=morton; do
  run $e --workload C3 --steps 20 --warmup 3
  run $e --nlam 7 --steps 30 --warmup 3
  run $e --nlam 1 --steps 30 --warmup 3
  run $e --workload C2 --steps 50 --warmup 5
  run $e --workload C5 --dtype f32 --steps 3 --warmup 1
  env $e REAL_GRID_DEFAULT_ONLY=1 python3 tools/real_grid_check.py 1000000 24 2>/dev/null | grep -i "ms per J\|patches" | head -2
done
