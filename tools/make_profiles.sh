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
bash tools/prof_pmc.sh ${tag}_default > gpurun_out/${tag}_pmc.txt 2>&1 || { tail -5 gpurun_out/${tag}_pmc.txt; exit 1; }
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 1
python3 bench.py --workload C3 --no-secondary > gpurun_out/${tag}_bench_c3.json 2>/dev/null || exit 1
python3 bench.py --workload C5 --dtype f32 --steps 3 --warmup 1 --no-secondary > gpurun_out/${tag}_bench_c5_f32.json 2>/dev/null || exit 1
cat gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_pmc.txt
