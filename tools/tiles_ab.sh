#!/bin/bash
# EXPERIMENT: storage order = Morton curve (default) against row-major tiles (VRT_EXP_TILES=cols,rows), C4 step time
for e in ${TILES:-"" "VRT_EXP_TILES=16,32" "VRT_EXP_TILES=12,24" "VRT_EXP_TILES=20,40" "VRT_EXP_TILES=8,16" ""}; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s ms/step %7.3f sweep %7.3f' % ('$e' or 'morton', d['ms_per_step'], d['roofline']['sweep_only']['ms']))"
done
