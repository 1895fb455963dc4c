#!/bin/bash
# GPU-box helper: PMC pass over one config-4-shaped step (deep duplex-UMI panel)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
tag=$1; shift
timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -- python3 bench.py --steps 2 --warmup 0 --tiles 2 --serial --no-cpu-baseline --no-extras --umi --depth 2000 --tile-kb 100 > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err
f=$(find gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if any(t in k for t in ("fam", "duplex", "p2_fast", "frag16", "prep_fast")): print(k, {c: int(x) for c, x in v.items()})
PY
