#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/: kernel-trace stats of the default C4
# bench command and the PMC passes of ONE step (separate runs per counter group, no tracing
# alongside), then the plain bench lines.  Run on the GPU box:  bash tools/make_profiles.sh <round-tag>
# (results under gpurun_out/; tools/summarise_profiles.py copies the summaries into profiles/<round>/;
# bench.py quotes roofline.traffic only once that summary carries the library's digest, so the committed
# bench_default.json is a `python bench.py > gpurun_out/<tag>_bench_default.json` run AFTER the summary
# step, followed by the summary step once more)
tag=${1:-r2}
bash tools/prof_kernels.sh ${tag}_default > gpurun_out/${tag}_kernel_stats.txt 2>&1 || { tail -5 gpurun_out/${tag}_kernel_stats.txt; exit 1; }
PMC_EXTRA="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" bash tools/prof_pmc.sh ${tag}_default > gpurun_out/${tag}_pmc.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc.txt; exit 1; }
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 1
python3 bench.py --workload C3 --no-secondary > gpurun_out/${tag}_bench_c3.json 2>/dev/null || exit 1
python3 bench.py --workload C5 --dtype f32 --steps 3 --warmup 1 --no-secondary > gpurun_out/${tag}_bench_c5_f32.json 2>/dev/null || exit 1
python3 tools/real_grid_check.py 1000000 24 > gpurun_out/${tag}_real_grid_1m.txt 2>&1 || exit 1
# kernel trace of the same on the default path only (1 M density-stratified, tessellated sites x 12 x 24 wavelengths)
rm -rf gpurun_out/prof_${tag}_real1m; mkdir -p gpurun_out/prof_${tag}_real1m
TMPDIR=/tmp REAL_GRID_DEFAULT_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_real1m -o p -- python3 tools/real_grid_check.py 1000000 24 > gpurun_out/prof_${tag}_real1m/run.log 2>&1 || exit 1
for w in C3 C5; do
  rm -rf gpurun_out/prof_${tag}_$w; mkdir -p gpurun_out/prof_${tag}_$w
  extra=""; [ $w = C5 ] && extra="--dtype f32"
  TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$w -o p -- python3 bench.py --workload $w $extra --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-critical-path > gpurun_out/prof_${tag}_$w/bench.log 2>&1 || exit 1
done
cat gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_pmc.txt
