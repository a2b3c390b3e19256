#!/bin/bash
# Host code of libsbgpu.so under AddressSanitizer + UBSan (CPU build only: GPU ASan is not available on this
# pool).  Builds a second copy of the library with the host side instrumented (-fno-gpu-sanitize) and runs
# the CPU test suite against it.  Usage (repo root): bash tools/asan_cpu.sh
set -e
OUT=${1:-/tmp/sbgpu_asan}
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
make -s -C strawberry_amd/csrc OUT=$OUT HIPFLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off \
  -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -I$(pwd)/include"
# the two tests left out link a plain C / C++ program against the library (they would need the runtime too)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  SBGPU_LIB=$OUT/libsbgpu.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider -k "not plain_c and not cxx14"
# ThreadSanitizer variant (threaded host bookkeeping), by hand:
#   make -s -C strawberry_amd/csrc OUT=/tmp/sbgpu_tsan HIPFLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=thread -fno-gpu-sanitize -I$(pwd)/include"
#   LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so) SBGPU_HOST_THREADS=8 \
#     SBGPU_LIB=/tmp/sbgpu_tsan/libsbgpu.so python -m pytest tests/test_exonbin_oracle.py tests/test_abi.py -q -m "not gpu" -k "not plain_c"
