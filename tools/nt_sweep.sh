#!/bin/bash
# diagnostics: C4 step time over workgroup size (entries per patch) x strip width of the storage order
run() { env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['sweep_only']['ms'],3))"; }
run VRT_PATCH_NT=1024 VRT_STORE_ORDER=strips:20
run VRT_PATCH_NT=1024 VRT_STORE_ORDER=strips:28
run VRT_PATCH_NT=1024 VRT_STORE_ORDER=strips:36
run VRT_PATCH_NT=1024 VRT_STORE_ORDER=strips:28 VRT_PATCH_TARGET=512
run VRT_PATCH_NT=256 VRT_STORE_ORDER=strips:14
run VRT_PATCH_NT=256 VRT_STORE_ORDER=strips:20
run VRT_PATCH_NT=512 VRT_STORE_ORDER=strips:20
