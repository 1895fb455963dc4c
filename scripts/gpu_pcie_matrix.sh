#!/bin/bash
# GPU-box helper: the pcie_inclusive leg of bench.py under different numbers of hardware queues / tiles in flight.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for cfg in "4 3" "16 3" "16 4" "24 5" "16 2"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 UVC_BENCH_THREADS=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 12 --warmup 2 > gpurun_out/pcie_m.json 2> gpurun_out/pcie_m.err || { tail -3 gpurun_out/pcie_m.err; exit 3; }
  python3 -c "
import json; j=json.load(open('gpurun_out/pcie_m.json')); p=j['pcie_inclusive']
print('queues $1 threads $2: value %.2f ms/step (%.1f M/s), pcie_inclusive %.2f ms/step (%.1f M/s), resident %.2f' % (j['ms_per_step'], j['value']/1e6, p['ms_per_step'], p['value']/1e6, j['resident']['ms_per_step']))"
done
