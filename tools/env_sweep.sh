#!/bin/bash
# diagnostics: bench C4 under a list of environment settings: tools/env_sweep.sh "A=1 B=2" "A=2" ...
for e in "$@"; do
  env $e python bench.py --workload C4 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s ms_per_step %.2f sweep_ms %.2f'%('$e',d['ms_per_step'],d['roofline']['sweep_ms_per_step']))"
done
