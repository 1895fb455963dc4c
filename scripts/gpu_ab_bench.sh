#!/bin/bash
# GPU-box helper: the quick bench alternating between the in-tree library ("-") and other builds of the same ABI, on one box.
#   bash scripts/gpu_ab_bench.sh - build_ab/libuvcgpu_x.so - build_ab/libuvcgpu_x.so
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset UVCGPU_LIBRARY; else export UVCGPU_LIBRARY=$PWD/$lib; fi
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-side > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || { tail -20 gpurun_out/bench_ab.err; exit 3; }
  python3 - "$lib" <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_ab.json"))
r = j.get("resident", {})
k = r.get("kernel_ms", {})
print("%-32s value %.3f ms  in_flight4 %.3f  resident %.3f | " % (sys.argv[1], j["ms_per_step"], j.get("resident_in_flight4", {}).get("ms_per_step", 0), r.get("ms_per_step", 0))
      + "  ".join("%s %.3f" % (n.replace("k_", ""), k[n]) for n in ("k_frag", "k_p2_fast_base", "k_p2_fast_link", "k_prep_fast", "k_p2_mism", "k_frag_generic", "k_p2_items", "k_fam_p4", "k_fam_p5", "k_score_all") if n in k))
PY
done
