#!/bin/bash
# GPU-box helper: one rocprofv3 PMC pass over a short bench run (counters given as arguments).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
tag=$1; shift
timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -- python3 bench.py --steps 2 --warmup 0 --tiles 2 --serial --no-cpu-baseline --no-extras --tile-kb ${TILE_KB:-200} > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err
rc=$?
f=$(find gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1)
echo "file: $f rc=$rc"
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(k, {c: int(x) for c, x in v.items()})
PY
exit 0
