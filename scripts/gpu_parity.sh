#!/bin/bash
# GPU-box helper: parity tests of the HIP path against the oracle.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rocminfo | grep -m2 gfx > gpurun_out/info.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity.log 2>&1
rc=$?
tail -40 gpurun_out/parity.log
exit $rc
