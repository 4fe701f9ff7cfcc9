#!/bin/bash
# diagnostics: C4 step time over strip width of the storage order x VRT_PATCH_TARGET
for w in ${WIDTHS:-16 20 24}; do for t in ${TARGETS:-704 768 832 1024}; do
  VRT_STORE_ORDER=strips:$w VRT_PATCH_TARGET=$t timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strips:$w VRT_PATCH_TARGET=$t ms/step', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3))"
done; done
