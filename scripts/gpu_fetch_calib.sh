#!/bin/bash
# GPU-box helper: FETCH_SIZE against known byte counts for this library's access patterns (scripts/ubench/fetch_calib.hip, built on the CPU box).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/fc
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -o fc -- scripts/ubench/fetch_calib > gpurun_out/fetch_calib_run.txt 2>&1 || { tail -5 gpurun_out/fetch_calib_run.txt; exit 3; }
grep -v '^W2026\|^E2026' gpurun_out/fetch_calib_run.txt
f=$(find /tmp/fc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
useful = {"k_stream16": 2 << 30, "k_stream4": 2 << 30, "k_seg128": (1 << 21) * 128, "k_sparse4<16>": (1 << 24) * 4, "k_sparse4<32>": (1 << 23) * 4, "k_rec96": ((2 << 30) // 96 // 64 * 64) * 96}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k not in useful: continue
    fb = float(r["Counter_Value"]) * 1024.0
    print("%-14s FETCH_SIZE %8.1f MB   useful %8.1f MB   FETCH_SIZE / useful = %.3f" % (k, fb / 1e6, useful[k] / 1e6, fb / useful[k]))
PY
