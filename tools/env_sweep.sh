#!/bin/bash
# diagnostics: bench C4 under a list of environment settings: tools/env_sweep.sh "A=1 B=2" "A=2" ...   (BENCH_ARGS: extra bench arguments)
for e in "$@"; do
  env $e python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path $BENCH_ARGS 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%-50s ms_per_step %.3f sweep_ms %.3f" % (sys.argv[1], d["ms_per_step"], d["roofline"]["sweep_only"]["ms"]))' "$e"
done
