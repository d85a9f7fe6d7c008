#!/bin/bash
# Host side of the library under UndefinedBehaviorSanitizer: the .cpp files of csrc/ compiled with
# -fsanitize=undefined on the host pass only (device code as always), linked with the regular kernel
# objects into build/ubsan/libmoptimizer_hip.so, and the C++ drop-in programs built against it.
#   scripts/ubsan_host.sh build      (here or on the GPU box)
#   scripts/ubsan_host.sh run        (GPU box: runs the programs; any report fails the run)
# (GPU AddressSanitizer is not available on this pool; this instruments host code only.)
set -e
cd "$(dirname "$0")/.."
OUT=build/ubsan
HIPCC=/opt/rocm/bin/hipcc
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN="-fsanitize=undefined -fno-sanitize-recover=undefined"
if [ "$1" = "build" ]; then
  make >/dev/null
  mkdir -p $OUT
  for f in c_abi lm combine group icp jit_model device_pool aql; do
    $HIPCC -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Imoptimizer_0_amd/csrc -x hip \
      -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
      -c moptimizer_0_amd/csrc/$f.cpp -o $OUT/$f.o
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libmoptimizer_hip.so $OUT/*.o \
    build/obj/sweep_kernels.o build/obj/fd_kernels.o build/obj/icp_grid.o build/obj/lm_kernels.o \
    -L/opt/rocm/lib -lrccl -lhiprtc -lhsa-runtime64 -lpthread -lrt -Wl,-rpath,/opt/rocm/lib
  for t in dropin_point2point dropin_models dropin_device_lm; do
    $CLANG -O1 -g -std=c++17 $SAN -Iinclude -Ioracle -Itests/support -o $OUT/$t tests/cpp/$t.cpp \
      -L$OUT -lmoptimizer_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -lpthread
  done
  ls -la $OUT | tail -6
elif [ "$1" = "run" ]; then
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  python3 -c "import numpy as np; from tests import datasets as ds; np.ascontiguousarray(ds.facade_pair()[0], dtype='<f8').tofile('$OUT/facade.f64')"
  for t in dropin_point2point dropin_models dropin_device_lm; do
    if [ $t = dropin_point2point ]; then args="$OUT/facade.f64"; else args=""; fi
    timeout -k 10 300 $OUT/$t $args > $OUT/$t.log 2>&1 || { echo "$t FAILED"; tail -20 $OUT/$t.log; exit 1; }
    echo "$t: $(grep -c '^PASS' $OUT/$t.log) PASS lines, $(grep -c 'runtime error' $OUT/$t.log) runtime errors; $(tail -1 $OUT/$t.log)"
  done
  # the Python-driven random walk over the stateful calls, through the instrumented library
  # (GPU box only: overwrites the scratch copy's library; the sanitizer runtime is preloaded)
  if [ -n "$GRAFT_REPO_ROOT" ]; then
    cp $OUT/libmoptimizer_hip.so moptimizer_0_amd/lib/libmoptimizer_hip.so
    RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
    for seed in 201 202 203; do
      LD_PRELOAD=$RT timeout -k 10 300 python3 tests/tools/api_fuzz.py --seed $seed --steps 5000 > $OUT/fuzz_$seed.log 2>&1 \
        || { echo "fuzz seed $seed FAILED"; grep -v amdgpu $OUT/fuzz_$seed.log | tail -20; exit 1; }
      echo "fuzz seed $seed: $(grep -c 'runtime error' $OUT/fuzz_$seed.log) runtime errors; $(grep -v amdgpu $OUT/fuzz_$seed.log | tail -1 | cut -c1-60)"
    done
  fi
else
  echo "usage: $0 build|run"; exit 2
fi
