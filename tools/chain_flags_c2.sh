export VRT_LIB_PATH=$PWD/voronoirt_amd/libvrt_hip_diag.so
run() { label=$1; shift
  line=$(env "$@" python3 bench.py --workload C2 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-critical-path 2>>gpurun_out/chain_flags_err.log | tail -1)
  echo "$label $(echo "$line" | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ms_per_step %.3f sweep_ms %.3f launches %d" % (j["ms_per_step"], r["sweep_only"]["ms"], r["launches_per_step"]))')"
}
for f in 0 256 512 1024 1792 1; do run "C2 diag chain flags=$f" VRT_PATCH_CHAIN=1 VRT_DEBUG_FLAGS=$f; done
