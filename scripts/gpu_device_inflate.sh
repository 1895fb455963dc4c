#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_inflate.py tests/test_io.py tests/test_pipeline.py -x -q -m gpu > gpurun_out/dev_inflate_tests.log 2>&1 || { tail -30 gpurun_out/dev_inflate_tests.log; exit 1; }
tail -2 gpurun_out/dev_inflate_tests.log
timeout -k 10 700 python scripts/bench_cli_cpus.py > gpurun_out/cli_dev_inflate.log 2>&1; cut -c1-330 gpurun_out/cli_dev_inflate.log | tail -12
