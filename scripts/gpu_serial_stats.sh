#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of a SERIAL bench run (one handle, nothing overlapped): every kernel's own duration, per step.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-serial}; shift
mkdir -p gpurun_out/$TAG
STEPS=${STEPS:-6}
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps $STEPS --warmup 1 --tiles 2 --serial --no-cpu-baseline --no-extras --no-side "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/err.txt || { tail -5 gpurun_out/$TAG/err.txt; exit 1; }
f=$(find gpurun_out/$TAG/trace -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/$TAG/kernel_stats.csv
python3 - "$f" $((STEPS + 1)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations per step: %.3f ms (%d steps incl. warm-up)" % (tot / 1e6 / steps, steps))
for r in rows[:60]:
    print("%-64s calls/step %5.1f  us/step %8.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:64], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e3 / steps))
PY
