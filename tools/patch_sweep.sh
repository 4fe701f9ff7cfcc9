#!/bin/bash
# diagnostics: step time of the fused patch path over kernel shapes (K entries/thread, Q pairs, NT threads)
# and patch sizes.  usage: tools/patch_sweep.sh WORKLOAD "K Q NT OWN" ["K Q NT OWN" ...]   (OWN 0 = default;
# "steps" = the layer-step path);  EXTRA="--dtype f32" adds bench arguments
W=${1:-C4}; shift
OUT=gpurun_out/patch_sweep_$W.txt
run() { # label, env...
  label=$1; shift
  line=$(env "$@" python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-critical-path $EXTRA 2>>gpurun_out/patch_sweep_err.log | tail -1)
  echo "$W $label $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f path %s launches %d frac %.4f plan_s %.2f" % (j["ms_per_step"], r["sweep_only"]["ms"], r["path"], r["launches_per_step"], r["frac"], j["setup_s"]["plan_create"]))')" | tee -a $OUT
}
for cfg in "$@"; do
  if [ "$cfg" = steps ]; then run steps VRT_PATH=steps; continue; fi
  set -- $cfg
  if [ "${4:-0}" != 0 ]; then own="VRT_PATCH_OWN=$4"; else own="VRT_PATCH_OWNX=0"; fi
  run "patches K=$1 Q=$2 NT=$3 OWN=${4:-0} $5" VRT_PATH=patches VRT_PATCH_K=$1 VRT_PATCH_Q=$2 VRT_PATCH_NT=$3 $own $5
done
