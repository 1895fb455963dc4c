#!/bin/bash
# the two device inflate kernels: parity tests, then the rate on the blocks of a 1 Mb x 300x tile (scripts/time_inflate.py)
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_inflate.py -x -q -m gpu > gpurun_out/inflate_wave_tests.log 2>&1 || { tail -30 gpurun_out/inflate_wave_tests.log; exit 1; }
tail -2 gpurun_out/inflate_wave_tests.log
timeout -k 10 400 python scripts/time_inflate.py 1000 > gpurun_out/inflate_lane.log 2>&1 && tail -3 gpurun_out/inflate_lane.log
UVCGPU_INFLATE_WAVE=1 timeout -k 10 400 python scripts/time_inflate.py 1000 > gpurun_out/inflate_wave.log 2>&1 && tail -3 gpurun_out/inflate_wave.log
UVCGPU_INFLATE_WAVE=8 timeout -k 10 400 python scripts/time_inflate.py 1000 > gpurun_out/inflate_wave8.log 2>&1 && tail -3 gpurun_out/inflate_wave8.log
