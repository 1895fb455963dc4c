#!/bin/bash
# CPU helper: the host-only code under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool runs no sanitizers).
#   libuvcio.so  (BGZF / BAM / BAI / FASTA readers, tumor-VCF reader, writers, planners)  -> tests/test_io.py, tests/test_tiles.py
#   liboracle.so (the checker itself: a checker that reads out of bounds checks nothing)  -> the oracle-side tests of the suite
# Builds into a scratch directory; the suite picks the builds up through UVCIO_LIBRARY / UVC_ORACLE_LIBRARY.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/uvc_sanitize}
mkdir -p $OUT
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
g++ $SAN -std=c++17 -fPIC -shared -Wall -Iinclude -o $OUT/libuvcio.so uvc_amd/csrc/uvc_io.cpp -lz -lpthread
g++ $SAN -std=c++14 -fPIC -fopenmp -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off -Iinclude -shared -o $OUT/liboracle.so \
    oracle/oracle_accumulate.cpp oracle/oracle_score.cpp oracle/oracle_capi.cpp oracle/oracle_group.cpp oracle/oracle_conblock.cpp
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export UVCIO_LIBRARY=$OUT/libuvcio.so UVC_ORACLE_LIBRARY=$OUT/liboracle.so
shift || true
python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider ${@:-tests/test_io.py tests/test_tiles.py tests/test_chain_golden.py tests/test_score_cpu.py tests/test_call_cpu.py tests/test_p45_cpu.py tests/test_group.py tests/test_conblock.py tests/test_indel_alleles_cpu.py tests/test_vcf_text.py tests/test_golden.py}
