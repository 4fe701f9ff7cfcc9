#!/bin/bash
# Host-side sanitizer screen: builds the C-ABI library's host sources with g++ -fsanitize=address,
# undefined (device launchers stubbed: tools/asan/stubs.cpp) and runs the CPU host tests on it.
# GPU AddressSanitizer is not available on this pool; the kernels are covered by the parity tests.
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/asan; mkdir -p $out
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ivoronoirt_amd/csrc \
    voronoirt_amd/csrc/vrt_api.cpp voronoirt_amd/csrc/vrt_grid.cpp voronoirt_amd/csrc/vrt_schedule.cpp voronoirt_amd/csrc/vrt_tessellate.cpp tools/asan/stubs.cpp \
    -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib -o $out/libvrt_hip.so
VRT_LIB_PATH=$PWD/$out/libvrt_hip.so LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_host.py -x -q -p no:cacheprovider "$@"
