#!/bin/bash
# round 4: the long-read suites and rates after a change of the strip kernels
set -o pipefail
O=gpurun_out/${1:-r04_len}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "long or strip" > $O/tests.log 2>&1; rc=$?; echo "long tests rc=$rc" | tee -a $O/summary.txt; tail -2 $O/tests.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scripts/long_fuzz.py > $O/fuzz.log 2>&1; rc=$?; echo "long fuzz rc=$rc" | tee -a $O/summary.txt; tail -1 $O/fuzz.log | cut -c1-300 | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS|identical|checked" | tee -a $O/summary.txt; }
run "10 kb, 4608 pairs" timeout -k 10 200 python scripts/long_read_bench.py 4608 230 10000 2 --seconds 5 &&
run "10 kb, 2304 pairs" timeout -k 10 200 python scripts/long_read_bench.py 2304 230 10000 0 --seconds 5 &&
run "8150, 6912 pairs" timeout -k 10 200 python scripts/long_read_bench.py 6912 230 8150 0 --seconds 4 &&
run "13 kb, 3072 pairs" timeout -k 10 200 python scripts/long_read_bench.py 3072 230 13000 0 --seconds 4 &&
run "16 kb, 1536 pairs" timeout -k 10 200 python scripts/long_read_bench.py 1536 230 16000 0 --seconds 4 &&
run "3 kb, 49152 pairs" timeout -k 10 200 python scripts/long_read_bench.py 49152 230 3000 0 --seconds 4
