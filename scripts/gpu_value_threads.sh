#!/bin/bash
# GPU-box helper: the default stream of bench.py with 1..4 host threads driving the tiles (UVC_BENCH_VALUE_THREADS).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for t in 1 2 3 4 1 2; do
  UVC_BENCH_VALUE_THREADS=$t timeout -k 10 300 python3 bench.py --no-side --no-cpu-baseline --no-extras --steps 24 --warmup 4 > gpurun_out/thr_$t.json 2> gpurun_out/thr_$t.err || { tail -3 gpurun_out/thr_$t.err; exit 1; }
  python3 - <<PY
import json
j=json.load(open("gpurun_out/thr_$t.json"))
print("threads $t: ms/step %.3f value %.1f M  dom %s %.3f ms frac %.4f" % (j["ms_per_step"], j["value"]/1e6, j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
PY
done
