#!/bin/bash
# GPU-box helper: -m gpu tests, then a short bench with the set_reads timing breakdown.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
UVCGPU_TIMING=1 timeout -k 10 600 python bench.py --steps 4 --warmup 1 --tiles 2 --serial --no-cpu-baseline --no-extras > gpurun_out/bench_timing.json 2> gpurun_out/bench_timing.err || { tail -30 gpurun_out/bench_timing.err; exit 3; }
tail -40 gpurun_out/bench_timing.err
timeout -k 10 900 python bench.py $BENCH_ARGS > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err || { tail -20 gpurun_out/bench_latest.err; exit 3; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_latest.json"))
print("value %.3e pos/s  ms/step %.2f  roofline %s frac %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"]))
print(j["kernel_ms"])
print(j.get("cpu_baseline"))
print("pcie_inclusive:", j.get("pcie_inclusive")); print("resident:", j.get("resident"))
PY
