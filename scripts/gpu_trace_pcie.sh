#!/bin/bash
# GPU-box helper: kernel + memory-copy trace of the bench's pcie_inclusive leg (H2D copies against kernels): how busy the copy engine and
# the compute units are over the leg's steady state.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/trace_pcie
rm -rf gpurun_out/trace_pcie/run
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_pcie/run -- python3 bench.py --steps 2 --warmup 0 --tiles 6 --no-cpu-baseline "$@" > gpurun_out/trace_pcie/bench.json 2> gpurun_out/trace_pcie/err.txt
python3 - <<'PY'
import csv, glob, json
j = json.load(open("gpurun_out/trace_pcie/bench.json")); p = j["pcie_inclusive"]; print("pcie_inclusive", p["ms_per_step"], "ms/step over", p["steps"], "steps; value", j["ms_per_step"])
mc = glob.glob("gpurun_out/trace_pcie/run/**/*memory_copy_trace.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/trace_pcie/run/**/*kernel_trace.csv", recursive=True)[0]
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in csv.DictReader(open(mc))]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
big = sorted(c for c in cp if c[1] - c[0] > 2000000 and "HOST_TO_DEVICE" in c[2])   # the quality / base columns
# the leg = the window of the last 2/3 of the large H2D copies (the resident leg behind it copies nothing large)
t_lo, t_hi = big[len(big) // 3][0], big[-1][1]
def union(iv):
    iv = sorted((max(a, t_lo), min(b, t_hi)) for a, b in iv if b > t_lo and a < t_hi)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None: tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else: cur_b = max(cur_b, b)
    if cur_b is not None: tot += cur_b - cur_a
    return tot
wall = t_hi - t_lo
h2d = [(a, b) for a, b, d in cp if "HOST_TO_DEVICE" in d]
print("window %.1f ms: H2D busy %.1f %%, kernels busy %.1f %%, either %.1f %%; %d large copies" % (wall / 1e6, 100 * union(h2d) / wall, 100 * union([(a, b) for a, b, _ in ks]) / wall, 100 * union(h2d + [(a, b) for a, b, _ in ks]) / wall, sum(1 for c in big if c[0] >= t_lo)))
small = [b - a for a, b in h2d if b - a < 2000000 and a >= t_lo]
print("small H2D copies in the window: %d, mean %.3f ms, total %.1f ms" % (len(small), sum(small) / max(1, len(small)) / 1e6, sum(small) / 1e6))
PY
