#!/bin/bash
# GPU-box helper: raw TCC counters for scripts/ubench/fetch_calib (which requests does FETCH_SIZE see?)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o 'TCC_EA0_RD[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*\|TCC_HIT[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCP_TCC_READ_REQ[A-Za-z0-9_]*' | sort -u | tr '\n' ' '; echo
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum"; do
  rm -rf /tmp/fc2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/fc2 -o fc -- scripts/ubench/fetch_calib > gpurun_out/fetch_calib_run2.txt 2>&1 || { tail -5 gpurun_out/fetch_calib_run2.txt; continue; }
  f=$(find /tmp/fc2 -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"): d[k][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items(): print("%-14s" % k, {a: int(b) for a, b in v.items()})
PY
done
