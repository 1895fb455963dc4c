#!/bin/bash
# GPU-box helper: the default bench.py run (every leg) with the key numbers of the line on stdout.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout -k 10 900 python bench.py "$@" > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || { tail -20 gpurun_out/bench_full.err; exit 3; }
echo "wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/bench_full.json"))
print("value %.4e ms/step %.3f repeats %s" % (j["value"], j["ms_per_step"], j["repeats"]["ms_per_step"]))
print("roofline", j["roofline"]["kernel"], round(j["roofline"]["frac"], 4), "|", (j["roofline"]["traffic_source"] or "")[:70])
rs = j.get("roofline_score")
if rs: print("score: in-stream %.3f ms frac %.4f; undisturbed %s" % (rs["kernel_ms"], rs["frac"], {k: round(v, 4) for k, v in rs.get("undisturbed", {}).items() if k in ("kernel_ms", "frac")}))
for k in ("config4", "all_out", "config5"):
    s = j.get(k)
    if s: print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in s.items() if a in ("ms_per_step", "value", "ms_per_pair", "ms_tumor_pass", "ms_normal_pass", "ms_keys_host", "tumor_keys", "normal_records_returned")}, "cpu", round(s.get("cpu_baseline", {}).get("value", 0)))
print("cpu_baseline", round(j.get("cpu_baseline", {}).get("value", 0)), "pcie", j.get("pcie_inclusive", {}).get("ms_per_step"), "in_flight4", j.get("resident_in_flight4", {}).get("ms_per_step"), "resident", j.get("resident", {}).get("ms_per_step"))
PY
