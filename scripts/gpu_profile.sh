#!/bin/bash
# GPU-box helper: rocprofv3 kernel-trace + stats of the default bench (no PMC in this pass).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
STEPS=${1:-3}
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras --tiles 4 > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
rc=$?
cat gpurun_out/prof_bench.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
exit $rc
