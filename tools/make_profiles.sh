#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/: kernel-trace stats of the default C4
# bench command (7 steps) and the PMC passes of ONE step (separate runs per counter group).
# Run on the GPU box:  bash tools/make_profiles.sh <round-tag>   (results under gpurun_out/)
tag=${1:-r1}
bash tools/prof_kernels.sh ${tag}_default --steps 5 --warmup 2 > gpurun_out/${tag}_kernel_stats.txt 2>&1 || exit 1
bash tools/prof_pmc.sh ${tag}_default > gpurun_out/${tag}_pmc.txt 2>&1 || exit 1
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 1
python3 bench.py --workload C3 --no-secondary > gpurun_out/${tag}_bench_c3.json 2>/dev/null || exit 1
cat gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_pmc.txt
