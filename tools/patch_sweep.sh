#!/bin/bash
# diagnostics: C4 step time of the fused patch path over kernel shapes (K entries/thread, Q pairs, NT threads)
# and patch sizes, next to the layer-step path.  usage: tools/patch_sweep.sh [workload] [extra bench args]
W=${1:-C4}; shift
OUT=gpurun_out/patch_sweep_$W.txt; : > $OUT
run() { # label, env...
  label=$1; shift
  line=$(env "$@" python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path $EXTRA 2>>gpurun_out/patch_sweep_err.log | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f path %s launches %d frac %.4f plan_s %.2f" % (j["ms_per_step"], r["sweep_only"]["ms"], r["path"], r["launches_per_step"], r["frac"], j["setup_s"]["plan_create"]))')" | tee -a $OUT
}
EXTRA="$*"
run steps VRT_PATH=steps
for shape in "1 2 1024" "1 4 1024" "2 1 512" "1 2 512" "2 2 512" "4 1 512" "1 4 512"; do
  set -- $shape
  run "patches K=$1 Q=$2 NT=$3" VRT_PATH=patches VRT_PATCH_K=$1 VRT_PATCH_Q=$2 VRT_PATCH_NT=$3
done
