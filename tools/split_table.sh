#!/bin/bash
# C4's grid: ms/step over the wavelength count (pair steps P per item) x workgroups per item (VRT_PATCH_SPLIT = s), balanced splits only
run() { python3 bench.py --nlam $1 --no-cpu-baseline --no-secondary --no-critical-path --no-caller-layout --steps 15 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nlam %3d P %2d split %2d  ms/step %7.3f launches %d' % ($1, ($1+1)//2, $2, d['ms_per_step'], d['roofline']['launches_per_step']))"; }
for spec in "16:8 4 3" "20:10 5 4" "26:13 7 5 4" "36:18 9 6 5" "51:9 7 6 5" "70:9 7 6 5 4" "100:10 8 7 6 5 4"; do
  nl=${spec%%:*}
  for s in ${spec#*:}; do VRT_PATCH_CHAIN=0 VRT_PATCH_SPLIT=$s run $nl $s; done
done
