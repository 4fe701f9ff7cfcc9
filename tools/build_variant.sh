#!/bin/bash
# usage: build_variant.sh <out .so name> [extra flags]  -- builds the product library with extra -D flags
out=$1; shift
cd /root/repo
S="vrt_api.cpp vrt_grid.cpp vrt_schedule.cpp vrt_patch.cpp vrt_lambda.cpp vrt_multi.cpp vrt_tessellate.cpp vrt_kernels.hip vrt_tables.hip vrt_layers.hip vrt_patch.hip vrt_regular.hip vrt_physics.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -I include -I voronoirt_amd/csrc -o voronoirt_amd/$out "$@" $(for s in $S; do echo voronoirt_amd/csrc/$s; done) -lpthread -ldl
