#!/bin/bash
# GPU-box helper: the whole -m gpu suite, then the layout A/B micro-benchmark (scripts/ubench/layout_ab, built on the CPU box).
# --capture=sys: messages native code writes to fd 2 (HIP runtime, glibc) reach the log even when the process aborts.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$FIRST" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q --capture=sys -k "$FIRST" > gpurun_out/gpu_first.log 2>&1
  echo "first: rc=$?"; tail -30 gpurun_out/gpu_first.log
fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 scripts/ubench/layout_ab > gpurun_out/layout_ab.txt 2>&1 || { cat gpurun_out/layout_ab.txt; exit 4; }
cat gpurun_out/layout_ab.txt
