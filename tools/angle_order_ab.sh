#!/bin/bash
# A/B in one box: the per-angle planes in the direction's storage order (VRT_ANGLE_ORDER=0) against the angles' own orders (1)
for rep in 1 2; do for ao in 0 1; do
  VRT_ANGLE_ORDER=$ao timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VRT_ANGLE_ORDER=$ao', '$*', 'ms/step', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3), 'caller-layout', round(d.get('caller_layout',{}).get('ms_per_step',0),3))"
done; done
