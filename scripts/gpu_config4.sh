#!/bin/bash
# GPU-box helper: BASELINE config 4 (200 kb duplex-UMI tile at 2000x): the full-size property test, then the bench line and a kernel trace.
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r02_config4}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ -z "$SKIP_TEST" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --capture=sys -k config4 > gpurun_out/$TAG/test.log 2>&1 || { tail -30 gpurun_out/$TAG/test.log; exit 2; }
tail -3 gpurun_out/$TAG/test.log
fi
timeout -k 10 900 python bench.py --umi --depth 2000 --tile-kb 200 --tiles 4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 3; }
python - gpurun_out/$TAG/bench.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("value %.3e pos/s  ms/step %.2f" % (j["value"], j["ms_per_step"]))
print("pipelined kernel_ms", j["kernel_ms"])
print("resident", j.get("resident", {}).get("ms_per_step"), j.get("resident", {}).get("kernel_ms"))
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --umi --depth 2000 --tile-kb 200 --tiles 2 --steps 4 --warmup 1 --serial --no-cpu-baseline --no-extras > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/trace.err || exit 11
f=$(find gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/trace
head -14 gpurun_out/$TAG/kernel_stats.csv | cut -c1-130
