#!/bin/bash
# GPU-box helper: the config-4 shaped bench (200 kb x 2000x duplex-UMI) for several builds of the library ("-" = in-tree): step time and the family kernels.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset UVCGPU_LIBRARY; else export UVCGPU_LIBRARY=$PWD/$lib; fi
  timeout -k 10 600 python3 bench.py --umi --depth 2000 --tile-kb 200 --tiles 2 --steps 6 --warmup 2 --no-cpu-baseline --no-side > gpurun_out/bench_c4v.json 2> gpurun_out/bench_c4v.err || { tail -20 gpurun_out/bench_c4v.err; exit 3; }
  python3 - "$lib" <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_c4v.json"))
k = j.get("resident", {}).get("kernel_ms", {})
print("%-34s step %.2f ms  resident %.2f | " % (sys.argv[1], j["ms_per_step"], j.get("resident", {}).get("ms_per_step", 0))
      + "  ".join("%s %.2f" % (n.replace("k_", ""), k[n]) for n in ("k_fam_p4", "k_fam_p5", "k_duplex", "k_fam_stat", "k_p2_fast_base", "k_p2_fast_link", "k_frag", "k_prep_fast", "k_score_all") if n in k))
PY
done
