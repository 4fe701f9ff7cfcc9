#!/bin/bash
# diagnostics: C4 layer-step throughput against the number of wavelengths (fixed per-layer latency
# vs per-wavelength bandwidth cost; working set vs the 256 MB Infinity Cache)
export VRT_PATH=steps
for nl in ${NLAMS:-2 12 24 36 51 64 100}; do
  python bench.py --workload C4 --nlam $nl --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nlam %3d streams ${VRT_STEP_STREAMS:-2} ms_per_step %.2f sweep_ms %.2f  G-updates/s %.1f  sweep-only G/s %.1f'%($nl,d['ms_per_step'],d['roofline']['sweep_ms_per_step'],d['value']/1e9, 995566*12*$nl/d['roofline']['sweep_ms_per_step']/1e6))"
done
