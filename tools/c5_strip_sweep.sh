#!/bin/bash
# diagnostics: C5 (fp32 values, four wavelengths per lane) step time over the strip width of the storage order
for w in ${WIDTHS:-16 20 24 28}; do
  VRT_STORE_ORDER=strips:$w timeout -k 10 400 python3 bench.py --workload C5 --dtype f32 --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 f32 strips:$w ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
done
