#!/bin/bash
# Host-side sanitizer screen of the C-ABI library: every host translation unit (grid preparation, schedules, the patch
# builder with its layer-parallel workers, the copy lanes, the multi-device worker pools, the tessellation) is built with
# g++ and a sanitizer; the device translation units (*.hip) are replaced by generated stubs; the CPU host tests then run
# on that library.  usage: tools/asan_host.sh [asan|tsan] [pytest args...]
#   asan  -fsanitize=address,undefined (default)      tsan  -fsanitize=thread, the threaded builders driven with 16 threads
#   check no sanitizer, build + load only (tests/test_host.py runs this: the screen cannot go stale unnoticed)
# GPU AddressSanitizer is not available on this pool; the kernels are covered by the parity tests.
set -e
cd "$(dirname "$0")/.."
mode=${1:-asan}; [ $# -gt 0 ] && shift
case $mode in
  asan) san="-fsanitize=address,undefined"; pre="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)";;
  tsan) san="-fsanitize=thread"; pre="$(g++ -print-file-name=libtsan.so)";;
  check) san=""; pre="";;     # no sanitizer, no test run: does the host-only build (generated stubs included) still link and load?
  *) echo "usage: $0 [asan|tsan|check] [pytest args]"; exit 2;;
esac
out=gpurun_out/$mode; rm -rf $out; mkdir -p $out
opt="-O1 -g"; [ $mode = check ] && opt="-O0"
flags="-std=c++17 $opt -fPIC $san -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ivoronoirt_amd/csrc"
hosts="vrt_api vrt_grid vrt_schedule vrt_patch vrt_lambda vrt_multi vrt_tessellate"
for s in $hosts; do g++ $flags -c voronoirt_amd/csrc/$s.cpp -o $out/$s.o & done
g++ $flags -c tools/asan/stubs.cpp -o $out/stubs.o &
wait
# stubs of everything else: undefined vrt:: functions of the host objects + the header's extern "C" entry points that no host object defines
nm --defined-only $out/*.o | awk '$2 ~ /[TW]/ {print $3}' | sort -u > $out/defined.txt
nm -u $out/*.o | awk '$1 == "U" && ($2 ~ /^_ZN3vrt/ || $2 ~ /^vrt_/) {print $2}' | sort -u > $out/undef.txt
sed -e 's,/\*.*\*/,,g' include/voronoirt.h | tr '\n' ' ' | sed -e 's,/\*[^*]*\*\+\([^/*][^*]*\*\+\)*/,,g' | grep -o '\bvrt_[a-z_0-9]\+ *(' | tr -d ' (' | sort -u > $out/declared.txt
cat $out/undef.txt $out/declared.txt | sort -u | comm -23 - $out/defined.txt > $out/to_stub.txt
{ echo '.text'; while read sym; do printf '.globl %s\n.type %s,@function\n%s:\n  movl $-3, %%eax\n  ret\n' $sym $sym $sym; done < $out/to_stub.txt; echo '.section .note.GNU-stack,"",@progbits'; } > $out/stubs_auto.s
g++ -c $out/stubs_auto.s -o $out/stubs_auto.o
g++ -shared $san $out/*.o -L/opt/rocm/lib -lamdhip64 -lpthread -ldl -Wl,-rpath,/opt/rocm/lib -o $out/libvrt_hip.so
echo "[$mode] built $out/libvrt_hip.so: $(wc -l < $out/to_stub.txt) device-side symbols stubbed"
if [ $mode = check ]; then
  python3 -c "import ctypes, sys; ctypes.CDLL('$PWD/$out/libvrt_hip.so'); print('[check] loads with every symbol resolved')"
  exit 0
fi
export VRT_LIB_PATH=$PWD/$out/libvrt_hip.so LD_PRELOAD=$pre
if [ $mode = asan ]; then
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_host.py -x -q -p no:cacheprovider "$@"
else
  # the layer-parallel patch builder (atomic work counter, 16 threads), four of them at once, the level schedules and the
  # tessellation through the introspection entry points (tools/tsan_driver.py: no pytest, no torch -- both crawl under
  # TSan); a race is a non-zero exit
  TSAN_OPTIONS="halt_on_error=1 exitcode=66 second_deadlock_stack=1" VRT_NO_TORCH=1 timeout -k 10 1500 python tools/tsan_driver.py
fi
