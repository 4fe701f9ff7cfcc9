#!/bin/bash
# diagnostics: C4 step time over the number of internal streams of the per-layer launches x VRT_PATCH_TARGET
run() { env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3), 'launches', d['roofline']['launches_per_step'])"; }
run VRT_STEP_STREAMS=1
run VRT_STEP_STREAMS=1 VRT_PATCH_TARGET=1024
run VRT_STEP_STREAMS=1 VRT_PATCH_TARGET=1536
run VRT_STEP_STREAMS=2
run VRT_STEP_STREAMS=4
run VRT_STEP_STREAMS=4 VRT_PATCH_TARGET=384
run VRT_STEP_STREAMS=4 VRT_PATCH_TARGET=512
run VRT_STEP_STREAMS=3 VRT_PATCH_TARGET=512
