#!/bin/bash
# diagnostics: one-line summaries of bench.py runs: tools/bench_brief.sh "<env> -- <bench args>" ...
for spec in "$@"; do
  e="${spec%%--*}"; a="${spec#*--}"
  env $e python bench.py $a --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-46s ms/step %6.2f sweep_ms %6.2f frac %.3f path %s launches %d  G-upd/s %.1f'%('''$spec'''[:46],d['ms_per_step'],r['sweep_only']['ms'],r['frac'],r['path'],r.get('launches_per_step',0),d['value']/1e9))"
done
