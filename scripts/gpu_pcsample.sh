#!/bin/bash
# GPU-box helper: PC sampling of a short resident-style bench run (where do the waves of the hot kernels spend their time).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pcs
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L > gpurun_out/pcs/avail.txt 2>&1; grep -i -B2 -A12 "pc.sampl" gpurun_out/pcs/avail.txt | head -60
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${PCS_METHOD:-host_trap} --pc-sampling-unit ${PCS_UNIT:-time} --pc-sampling-interval ${PCS_INTERVAL:-50} --kernel-trace --output-format csv -d gpurun_out/pcs/run -- python3 bench.py --steps 3 --warmup 1 --tiles 2 --serial --no-cpu-baseline --no-extras > gpurun_out/pcs/bench.json 2> gpurun_out/pcs/err.txt
echo rc=$?
tail -5 gpurun_out/pcs/err.txt
find gpurun_out/pcs -type f | head; du -sh gpurun_out/pcs
