#!/bin/bash
# GPU-box helper: rocprofv3 kernel stats of the default-gate scoring kernels (1 Mb x 300x tile) for several builds of the library.
#   bash scripts/gpu_score_variants.sh build_ab/libuvcgpu_a.so build_ab/libuvcgpu_b.so ...   ("-" = the in-tree build)
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset UVCGPU_LIBRARY; else export UVCGPU_LIBRARY=$R/$lib; fi
  rm -rf /tmp/sv
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sv -o sv -- python3 $R/scripts/gpu_score_profile.py --kb 1000 > $R/gpurun_out/score_variant.log 2>&1 || { tail -5 $R/gpurun_out/score_variant.log; exit 3; }
  f=$(find /tmp/sv -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$lib" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("k_gate_scan", "k_enum", "k_gather(", "k_dpv", "k_dp4", "k_qual", "k_call", "k_keep")
d = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in rows if any(k in r["Name"] for k in keep)}
print("%-34s sum %.1f us  " % (sys.argv[2], sum(d.values())) + "  ".join("%s %.1f" % (k[2:], v) for k, v in sorted(d.items(), key=lambda kv: -kv[1])))
PY
done
