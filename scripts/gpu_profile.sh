#!/bin/bash
# GPU-box helper: rocprofv3 kernel-trace + stats of the default bench (no PMC in this pass); prints the top kernels with short names.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
STEPS=${1:-6}
shift
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras --tiles 4 "$@" > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
rc=$?
python3 - <<'PY'
import csv, glob, json, re
try:
    j = json.load(open("gpurun_out/prof_bench.json"))
    print("value %.3e ms/step %.2f" % (j["value"], j["ms_per_step"]))
except Exception as e:
    print("no bench json:", e)
fs = sorted(glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True))
if fs:
    rows = list(csv.DictReader(open(fs[-1])))
    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::([a-z_]+)", n)
        if n.startswith("void rocprim") and m:
            k = re.findall(r"detail::([a-z_]+(?:impl|iteration|offsets|kernel|merge|sort)[a-z_]*)", n)
            return "rocprim:" + (k[0] if k else m.group(1))
        return n.split("(")[0].replace("void ", "")[:60]
    agg = {}
    for r in rows:
        k = short(r["Name"]); a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-60s calls %5d  total %9.3f ms  avg %9.3f us  %5.1f%%" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
PY
exit $rc
