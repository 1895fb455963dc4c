#!/bin/bash
# GPU-box helper: smoke, a small and the default bench, and a rocprofv3 kernel-trace of the default bench.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 400 python bench.py --tile-kb 100 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_100kb.json 2> gpurun_out/bench_100kb.err || { tail -20 gpurun_out/bench_100kb.err; exit 2; }
cat gpurun_out/bench_100kb.json
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 3; }
cat gpurun_out/bench_default.json
