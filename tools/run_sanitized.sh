#!/bin/bash
# Runs pytest against a sanitizer build of the two native libraries (HOST code; device code is compiled as usual).
#   tools/run_sanitized.sh                         ASan + UBSan (make -C csrc asan), CPU box: ABI / loader / error paths /
#                                                  fake-tracer tests -- everything that needs no device
#   SAN=ubsan tools/run_sanitized.sh -m gpu        UBSan alone (make -C csrc ubsan) on a GPU box: the host paths under real
#                                                  use.  (With the ASan runtime preloaded HSA initialisation aborts: its
#                                                  address-space reservations collide with the shadow memory.)
# Python itself is not instrumented, so the sanitizer runtime is preloaded; leak checking is off (the interpreter and the
# HIP runtime keep their allocations until exit), everything else aborts the run with a report.
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
SAN=${SAN:-asan}
make -s -C "$REPO/nvidia-resiliency-ext_amd/csrc" $SAN
if [ "$SAN" = asan ]; then
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
else
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
fi
export NVRX_DEBUG_LIB_DIR="$REPO/nvidia-resiliency-ext_amd/nvrx_straggler/lib_$SAN"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0:detect_odr_violation=0${ASAN_OPTIONS:+:$ASAN_OPTIONS}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$REPO"
if [ $# -eq 0 ]; then
  set -- tests/test_host_logic.py tests/test_ktrace_datapath.py -k "datapath or per_key or key_format or engine_kernels or harvest_waits or pending_queue or random_launch or abi or symbol or exports or header or ktrace or kernel_trace or never_imports or c_dict_builders or c_flag_decoder or flag_memo or inplace_filled or recycled or beyond_their_stack or non_finite or tool_search_guard or gpu_timing_mode" -m "not gpu"
fi
LD_PRELOAD="$RT" PYTHONPATH="$REPO/nvidia-resiliency-ext_amd" python -c "from nvrx_straggler import _native, ktrace; assert 'lib_$SAN' in _native.lib_path() and 'lib_$SAN' in ktrace.lib_path(); print('sanitized libraries:', _native.lib_path())"
if [ "$SAN" = asan ]; then
  LD_PRELOAD="$RT" PYTHONPATH="$REPO/nvidia-resiliency-ext_amd" python -c "from nvrx_straggler import reporting; assert 'lib_asan' in reporting._pyread.__file__; print('sanitized dict builders:', reporting._pyread.__file__)"
fi
LD_PRELOAD="$RT" python -m pytest -x -q -p no:cacheprovider "$@"
