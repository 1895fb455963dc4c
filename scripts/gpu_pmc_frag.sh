#!/bin/bash
# GPU-box helper: FETCH_SIZE / WRITE_SIZE of the heavy accumulate kernels on the default tile (serial 2-step run), one line per kernel.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pf_$ctr
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pf_$ctr -- python3 bench.py --steps 2 --warmup 0 --tiles 2 --serial --no-cpu-baseline --no-extras --no-side > gpurun_out/pmc_frag.json 2> gpurun_out/pmc_frag.err || { tail -3 gpurun_out/pmc_frag.err; exit 3; }
done
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pf_%s/**/*counter_collection.csv" % ctr, recursive=True)[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg: res[k][ctr] = agg[k] / cnt[k] * 1024
for k in ("k_frag16<true>", "k_p2_fast<false, true, true>", "k_p2_fast<true, false, true>", "k_prep_fast<false>", "k_gather"):
    v = res.get(k, {})
    print("%-32s FETCH x 2 %.3f GB  WRITE %.3f GB  traffic %.3f GB" % (k, 2 * v.get("FETCH_SIZE", 0) / 1e9, v.get("WRITE_SIZE", 0) / 1e9, (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) / 1e9))
PY
