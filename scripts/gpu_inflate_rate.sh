#!/bin/bash
# host cores of the GPU box: the three versions of the decoder against zlib on a synthetic 200 kb x 300x BAM
set -e
mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from uvc_amd import synth
import bamwriter
reads = synth.generate_region(seed=3, region_len=200000, depth=300, beg=50000)
bamwriter.write_bam("/tmp/t200.bam", [("chrT", int(reads["end"]) + 1000)], bamwriter.records_from_reads(reads))
PY
lscpu | grep -E "Model name" > gpurun_out/inflate_rate.log
for h in scripts/ubench/old/inflate_r2a.h scripts/ubench/old/inflate_r2b.h uvc_amd/csrc/uvc_inflate_fast.h; do
  g++ -O3 -std=c++17 -DHDR="\"$PWD/$h\"" -o /tmp/ir scripts/ubench/inflate_rate.cpp -lz && /tmp/ir /tmp/t200.bam >> gpurun_out/inflate_rate.log
done
cat gpurun_out/inflate_rate.log
