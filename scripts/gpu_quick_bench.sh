#!/bin/bash
# GPU-box helper: the default bench without the CPU baseline and the side legs, key numbers on stdout.  Arguments go to bench.py.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python bench.py --no-cpu-baseline --no-side "$@" > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -20 gpurun_out/bench_quick.err; exit 3; }
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/bench_quick.json"))
print("value %.4e pos/s  ms/step %.3f  in_flight4 %s  pcie %s  resident %s" % (j["value"], j["ms_per_step"], j.get("resident_in_flight4", {}).get("ms_per_step"), j.get("pcie_inclusive", {}).get("ms_per_step"), j.get("resident", {}).get("ms_per_step")))
print("in-stream kernel_ms:", j["kernel_ms"])
print("resident kernel_ms:", j.get("resident", {}).get("kernel_ms"))
PY
