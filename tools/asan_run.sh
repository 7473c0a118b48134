#!/bin/bash
# ASan + UBSan run of the HOST side of libzkp_accel.so (SURVEY §5 "sanitizers" row): 4 300 lines of host
# orchestration with raw pointers, thread-local caches, 32 streams and worker threads, so far covered by functional tests only.
#   tools/asan_run.sh build      (anywhere: cross-compiles)     -> variants/asan/libzkp_accel.so, oracle/build/libzkp_oracle_asan.so
#   tools/asan_run.sh run        (GPU box)                      -> gpurun_out/asan/*.log + gpurun_out/r06_asan.txt
# Device code is compiled as always (-fno-gpu-sanitize); the sanitizer runtime is LD_PRELOADed into the Python process.
# detect_leaks=0: the interpreter and the HIP runtime never free their arenas; everything else is on.
set -u
cd "$(dirname "$0")/.."
# Runtime: GCC's libasan / libubsan, NOT the ROCm clang's.  The ROCm compiler-rt intercepts hsa_amd_memory_pool_allocate & co (it is
# built for device-side ASan with xnack+ and the instrumented /opt/rocm/lib/asan libraries, which this image lacks): preloaded into
# a HIP process it aborts inside the first device allocation ("AddressSanitizer: out of memory ... hsa_amd_memory_pool_allocate",
# first attempt of round 6).  The instrumentation clang emits only needs the plain __asan_* / __ubsan_* entry points, and GCC 11's
# runtimes export every one of them except __ubsan_handle_function_type_mismatch (-fno-sanitize=function); so the host objects are
# compiled by hipcc with -fsanitize=address,undefined, the library is linked WITHOUT a runtime, and GCC's runtimes are preloaded.
RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
case "${1:-run}" in
build)
  ZKP_BUILD_TAG=asan \
  ZKP_BUILD_DEFS="-fsanitize=address,undefined -fno-sanitize=function -fno-gpu-sanitize -fno-omit-frame-pointer -g" \
  ZKP_BUILD_LDFLAGS="" \
  python -m ckb_zkp_amd.build || exit 1
  nm -D --undefined-only variants/asan/libzkp_accel.so | grep -c "__asan_\|__ubsan_" | sed 's/^/sanitizer entry points referenced: /'
  ;;
run)
  mkdir -p gpurun_out/asan
  OUT=gpurun_out/r06_asan.txt
  export ZKP_ACCEL_LIB=$PWD/variants/asan/libzkp_accel.so
  export ZKP_C_DRIVER_FLAGS="-fsanitize=address,undefined"      # tests/c_driver.py: the C99 ABI program is sanitized by GCC itself
  rm -rf tests/c/build
  export LD_PRELOAD="$RT"
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$PWD/gpurun_out/asan/asan
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$PWD/gpurun_out/asan/ubsan
  {
    echo "# ASan + UBSan host build of libzkp_accel.so ($(date -u +%F)); runtime $RT"
    echo "# ASAN_OPTIONS=$ASAN_OPTIONS"
    echo "# UBSAN_OPTIONS=$UBSAN_OPTIONS"
    # (tests that start torch in a child process are left out: torch's lazy CUDA init dlopens libcaffe2_nvrtc.so through an
    #  $ORIGIN-relative RPATH, which the sanitizer's dlopen interceptor does not honour — "Error in dlopen: libcaffe2_nvrtc.so"
    #  before any of this library's code runs)
    run() { echo "## python -m pytest $* -m gpu -x -q"; timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4; }
    run tests/test_gpu_cabi.py
    run tests/test_gpu_concurrency.py
    run tests/test_gpu_multi.py -k "not multiprocess and not torchrun"
    run tests/test_gpu_config.py
    run tests/test_gpu_fuzz.py -k groth16
    run tests/test_gpu_fuzz.py -k marlin
    run tests/test_gpu_groth16.py -k "full_size and bn254-20"
    run tests/test_gpu_marlin.py -k native
    echo "## sanitizer reports"
    n=$(ls gpurun_out/asan 2>/dev/null | wc -l)
    echo "report files: $n"
    for f in gpurun_out/asan/*; do [ -f "$f" ] && { echo "--- $f"; head -60 "$f"; }; done
    [ "$n" = 0 ] && echo "0 errors: no AddressSanitizer / UndefinedBehaviorSanitizer report was written"
  } > $OUT 2>&1
  tail -40 $OUT
  ;;
esac
