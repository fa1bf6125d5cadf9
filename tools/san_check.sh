#!/bin/bash
# Sanitizer pass over the host side and the oracle (CPU box only; the reference's counterpart is test/run_valgrind.sh — valgrind is not in
# this image, gcc's sanitizers are).  Builds tiebrush_amd/_build_san_* and oracle/_build_san_* and runs, under each build:
#   - the CPU tests that drive the host codec, the formats, the oracle and the files -> files tool (pytest, the libraries through ctypes:
#     the sanitizer runtime is preloaded into python),
#   - tbh_tool round trips on the reference's fixtures (cat / soa / fastsoa / mergeorder / mkbam with worker threads).
# usage: bash tools/san_check.sh [address,undefined|thread ...]   (default: both);  the summary goes to stdout, details to /tmp/tbk_san/
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
LOG=/tmp/tbk_san; mkdir -p $LOG
MODES=${@:-"address,undefined thread"}
fail=0
for SAN in $MODES; do
  tag=${SAN//,/_}
  make -s -C tiebrush_amd/csrc/host SAN=$SAN > $LOG/build_host_$tag.log 2>&1 || { echo "[$SAN] host build FAILED"; tail -5 $LOG/build_host_$tag.log; fail=1; continue; }
  make -s -C oracle SAN=$SAN > $LOG/build_oracle_$tag.log 2>&1 || { echo "[$SAN] oracle build FAILED"; tail -5 $LOG/build_oracle_$tag.log; fail=1; continue; }
  HB=$ROOT/tiebrush_amd/_build_san_$tag; OB=$ROOT/oracle/_build_san_$tag
  if [ "$SAN" = thread ]; then RT=$(gcc -print-file-name=libtsan.so); else RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"; fi
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  export TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0"
  # -- the CPU tests on the sanitized libraries and tools
  TESTS="tests/test_host_codec.py tests/test_host_formats.py tests/test_oracle_golden.py tests/test_cpu_e2e_tool.py tests/test_metamorphic_cpu.py"
  if [ "$SAN" = thread ]; then
    # (ThreadSanitizer's runtime preloaded into an uninstrumented python deadlocks at start-up: under it the tests drive the sanitized
    # TOOLS — tbh_tool, tb_cpu_e2e: the threaded loader, the inflate / deflate pools, the writer — and load the plain libraries)
    TBK_TEST_TBH_TOOL=$HB/tbh_tool TBK_TEST_CPU_E2E=$OB/tb_cpu_e2e \
      python -m pytest tests/test_host_codec.py tests/test_host_formats.py tests/test_cpu_e2e_tool.py -x -q -p no:cacheprovider > $LOG/pytest_$tag.log 2>&1
  else
    LD_PRELOAD="$RT" TBK_HOST_LIB=$HB/libtbh.so TB_ORACLE_BUILD_DIR=$OB TBK_TEST_TBH_TOOL=$HB/tbh_tool TBK_TEST_CPU_E2E=$OB/tb_cpu_e2e \
      python -m pytest $TESTS -x -q -p no:cacheprovider > $LOG/pytest_$tag.log 2>&1
  fi
  rc=$?
  echo "[$SAN] pytest rc=$rc: $(tail -1 $LOG/pytest_$tag.log)"
  [ $rc -eq 0 ] || fail=1
  # -- round trips of the tool itself (threads on: fastsoa and mkbam use the worker pools)
  W=$LOG/work_$tag; rm -rf $W; mkdir -p $W/soa $W/fsoa
  G=tests/golden
  ( set -e
    $HB/tbh_tool cat $G/t1/t1.bam $W/t1.bam
    $HB/tbh_tool mergeorder $G/t2/t2s0.bam $G/t2/t2s1.bam $G/t2/t2s2.bam > $W/order.txt
    $HB/tbh_tool soa $W/soa $G/t2/t2s*.bam
    $HB/tbh_tool fastsoa $W/fsoa $G/t2/t2s*.bam
    $OB/tb_cpu_e2e -A -o $W/e2e.bam $G/t1/t1s*.bam
  ) > $W/tools.log 2>&1
  rc=$?
  n=$(grep -c "ERROR: \(Address\|Thread\|Undefined\)Sanitizer\|WARNING: ThreadSanitizer\|runtime error:" $W/tools.log $LOG/pytest_$tag.log | awk -F: '{s+=$2} END {print s+0}')
  echo "[$SAN] tool round trips rc=$rc, sanitizer reports: $n"
  [ $rc -eq 0 ] && [ "$n" = 0 ] || fail=1
done
[ $fail -eq 0 ] && echo "san_check: clean" || echo "san_check: FINDINGS (see $LOG)"
exit $fail
