#!/bin/bash
# GPU-box helper: kernel + memory-copy trace of the default bench run, reduced to the pcie_inclusive leg (H2D copies against kernels).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/trace_pcie
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_pcie/run -- python3 bench.py --steps 4 --warmup 1 --tiles 4 --no-cpu-baseline > gpurun_out/trace_pcie/bench.json 2> gpurun_out/trace_pcie/err.txt
python3 - <<'PY'
import csv, glob, json
j = json.load(open("gpurun_out/trace_pcie/bench.json")); print("pcie_inclusive", j["pcie_inclusive"]["ms_per_step"], "ms/step; value", j["ms_per_step"])
mc = glob.glob("gpurun_out/trace_pcie/run/**/*memory_copy_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(mc)))
print(rows[0].keys())
big = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), int(r.get("Bytes", r.get("Size", 0)) or 0)) for r in rows]
big = [b for b in big if b[1] - b[0] > 200000]
big.sort()
t0 = big[0][0]
# the last 40 large copies
for s, e, d, n in big[-60:]:
    print("%10.2f ms  %8.2f ms  %s  %s MB  %.1f GB/s" % ((s - t0) / 1e6, (e - s) / 1e6, d, n // 1000000, (n / max(1, e - s))))
PY
