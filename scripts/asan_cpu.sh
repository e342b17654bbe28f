#!/bin/bash
# ASan + UBSan run of the CPU restatement (oracle/gpt_oracle.c) over the golden-vector suite.  CPU box only (the GPU pool
# refuses sanitizer runs, and the product library has no CPU build: what can be sanitised is the checker).
#   scripts/asan_cpu.sh            -> exit code of pytest; any sanitizer report aborts the process (non-zero)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$ROOT/oracle" asan
ASAN_RT="$(gcc -print-file-name=libasan.so)"
UBSAN_RT="$(gcc -print-file-name=libubsan.so)"
[ -f "$ASAN_RT" ] || { echo "no libasan runtime next to gcc: skipped"; exit 77; }
cd "$ROOT"
# detect_leaks=0: CPython itself leaks by ASan's standards; everything else aborts on the first report
LD_PRELOAD="$ASAN_RT:$UBSAN_RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 GPT_ORACLE_LIB="$ROOT/oracle/libgpt_oracle_asan.so" OMP_NUM_THREADS=1 \
    python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider "$@"
