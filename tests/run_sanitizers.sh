#!/bin/bash
# The host side of libzkmi355 under clang's sanitizers (SURVEY 5 "race detection / sanitizers").  CPU only: the EMULATOR build of the library —
# every kernel's index logic as host C++, prover.hip's helper threads, pools, shared keys, the ZKQ1 compiler — compiled with -fsanitize=address / undefined / thread
# (`make -C zk-dcap-verifier_amd/csrc emu-asan emu-ubsan emu-tsan`), loaded by the ordinary CPU tests in place of tests/csrc/libzkmi355_emu.so.  GPU AddressSanitizer
# is not available on this pool.  usage: tests/run_sanitizers.sh [asan|ubsan|tsan ...]   logs: gpurun_out/sanitizers/<kind>.log; exit 0 = every run clean.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
KINDS="${*:-asan ubsan tsan}"
TESTS="tests/test_emu_kernels.py tests/test_quotient.py tests/test_native_prover.py tests/test_capi_prove.py tests/test_abi_no_throw.py tests/test_multi_rank.py"
mkdir -p "$ROOT/gpurun_out/sanitizers"
status=0
for kind in $KINDS; do
    make -C "$ROOT/zk-dcap-verifier_amd/csrc" -s -j4 "emu-$kind" || { echo "$kind: build failed"; status=1; continue; }
    export ZK_EMU_LIBDIR="$ROOT/tests/csrc/san/$kind"
    log="$ROOT/gpurun_out/sanitizers/$kind.log"
    case $kind in
        # python itself is not instrumented: the runtime is preloaded; leak checking would report the interpreter's own arenas, the library's device-buffer census is
        # tests/test_abi_no_throw.py's job
        asan)  pre="$RT/libclang_rt.asan-x86_64.so";  opts="ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_stack_use_after_return=0:exitcode=66" ;;
        ubsan) pre="$RT/libclang_rt.ubsan_standalone-x86_64.so"; opts="UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=66" ;;
        tsan)  pre="$RT/libclang_rt.tsan-x86_64.so";  opts="TSAN_OPTIONS=halt_on_error=0:exitcode=66:report_signal_unsafe=0:second_deadlock_stack=1" ;;
        *) echo "unknown sanitizer $kind"; exit 2 ;;
    esac
    tests="$TESTS"
    # TSan: the gloo worlds of test_multi_rank.py are separate PROCESSES (nothing for a race detector to see across them) and their collective is the uninstrumented
    # PyTorch runtime, whose own condition variables TSan reports; the ranks-as-threads runs of tests/csrc/capi_prove.c (test_capi_prove.py: 2, 4 and 8 ranks in one
    # process, shared tables, a failing rank) are the multi-rank coverage under TSan
    [ "$kind" = tsan ] && tests="${TESTS/ tests\/test_multi_rank.py/}"
    echo "== $kind: $tests"
    ( cd "$ROOT" && env "$opts" LD_PRELOAD="$pre" ZK_SANITIZER="$kind" python -m pytest $tests -x -q -m "not gpu" -p no:cacheprovider ) > "$log" 2>&1
    rc=$?
    tail -3 "$log"
    if [ $rc -ne 0 ] || grep -q "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" "$log"; then echo "$kind: FINDINGS (see $log)"; status=1; else echo "$kind: clean"; fi
done
exit $status
