#!/bin/bash
# GPU-box helper: everything the round's profiles/ directory is made from.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (summary csv)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) -> per-kernel HBM-side traffic
#   3. the default bench.py run with the CPU baseline
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
unset UVCGPU_CHECK_PRESENCE   # (a test switch: the validator sweeps every plane behind each accumulate)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 8 --warmup 2 --tiles 4 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 11
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- python3 bench.py --steps 2 --warmup 0 --tiles 2 --serial --no-cpu-baseline --no-extras > $OUT/pmc_$ctr.json 2> $OUT/pmc_$ctr.err || exit 12
done
python3 - $OUT <<'PY'
import csv, sys, glob, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    fs = glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, ctr), recursive=True)
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg: res[k][ctr + ("_per_launch" if ctr.startswith("SQ_") else "_KB_per_launch")] = agg[k] / cnt[k]
# gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; both counters are in KiB
for k, v in res.items():
    v["traffic_bytes_per_launch"] = (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0
res["_workload"] = {"tile_kb": 1000, "depth": 300}   # the default bench.py workload
sys.path.insert(0, "."); import bench; res["_source_hash"] = bench.kernel_source_hash()   # bench.py quotes these counters only for the kernel sources they were counted on
json.dump(res, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
res.pop("_workload"); res.pop("_source_hash")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:12]: print(k, {a: round(b) for a, b in v.items()})
PY
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 13; }
cat $OUT/bench.json
timeout -k 10 900 python3 bench.py --serial --no-cpu-baseline --no-extras > $OUT/bench_serial.json 2>> $OUT/bench.err || exit 14
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU
