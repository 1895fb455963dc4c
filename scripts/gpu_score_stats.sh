#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the scoring kernels alone (default gate on a 1 Mb x 300x tile, -A on a 200 kb tile).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
for leg in "default --kb 1000" "allout --kb 200 --all-out" "umi --kb 200 --depth 2000 --umi"; do
  set -- $leg; name=$1; shift
  rm -rf /tmp/sp_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$name -o sp -- python3 $R/scripts/gpu_score_profile.py "$@" > $R/gpurun_out/score_prof_$name.log 2>&1 || { tail -5 $R/gpurun_out/score_prof_$name.log; exit 3; }
  f=$(find /tmp/sp_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/score_stats_$name.csv
  tail -1 $R/gpurun_out/score_prof_$name.log
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("k_gate_scan", "k_enum", "k_gather", "k_dpv", "k_dp4", "k_qual", "k_call", "k_keep", "k_score", "k_scan", "fillBuffer", "copyBuffer")
tot = 0.0
for r in rows:
    if any(k in r["Name"] for k in keep):
        print("  %-28s calls %4s  avg %9.1f us  min %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
        if "Buffer" not in r["Name"]: tot += float(r["AverageNs"]) / 1e3
print("  sum of the scoring kernels' averages: %.1f us" % tot)
PY
done
