#!/bin/bash
# Round-5 additions to tools/make_profiles.sh (run AFTER it, same tag): the Λ-iteration session under the kernel trace in both
# layouts, the PMC passes of C5 (fp32 storage), the chained launch at the sizes of a GPU's share in a node, the iteration breakdown.
tag=${1:-r5}
export TMPDIR=/tmp
for nat in 1 0; do
  out=gpurun_out/prof_${tag}_lambda_native$nat; rm -rf $out; mkdir -p $out
  VRT_LAMBDA_NATIVE=$nat rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/lambda_session_trace.py > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
  tail -1 $out/run.log
done
bash tools/prof_pmc.sh ${tag}_c5 --workload C5 --dtype f32 > gpurun_out/${tag}_c5_pmc.txt 2>&1 || { tail -5 gpurun_out/${tag}_c5_pmc.txt; exit 1; }
python3 tools/iteration_breakdown.py > gpurun_out/${tag}_iteration_breakdown.txt 2>&1 || exit 1
{
for args in "--workload C2 --steps 50 --warmup 5" "--nlam 1 --steps 30 --warmup 3" "--nlam 3 --steps 30 --warmup 3" "--nlam 7 --steps 30 --warmup 3" "--nlam 13 --steps 20 --warmup 3" "--nlam 26 --steps 20 --warmup 3" "--workload C3 --steps 20 --warmup 3" "--steps 20 --warmup 3"; do
  for lay in native caller; do
    python3 bench.py $args --sj-layout $lay --no-cpu-baseline --no-secondary --no-caller-layout 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-36s S/J %-7s ms/step %7.3f sweep %7.3f launches %4d critical_path_ms %s' % ('$args', '$lay' if 'sweep order' in d['config']['sj_layout'] or '$lay' == 'caller' else 'caller*', d['ms_per_step'], r['sweep_only']['ms'], r['launches_per_step'], r.get('critical_path_ms')))"
  done
done
} > gpurun_out/${tag}_chain_sweep_final.txt 2>&1
cat gpurun_out/${tag}_chain_sweep_final.txt
