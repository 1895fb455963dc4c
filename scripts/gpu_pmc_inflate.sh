#!/bin/bash
# GPU-box helper: PMC pass over the BGZF inflate kernel (scripts/time_inflate.py)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
tag=$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -- python3 scripts/time_inflate.py 100 > gpurun_out/pmc/$tag.out 2> gpurun_out/pmc/$tag.err
f=$(find gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if "inflate" in k: print(k, {c: int(x) for c, x in v.items()})
PY
