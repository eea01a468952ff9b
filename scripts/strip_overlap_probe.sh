#!/bin/bash
# round 4: long reads beyond 16 384 target rows (several passes of the strip kernel) against the workgroup kernel they ran on (MGL_STRIP=1: never the strip kernel)
set -o pipefail
O=gpurun_out/${1:-r04_len}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "long or strip" > $O/tests.log 2>&1; rc=$?; echo "long tests rc=$rc" | tee -a $O/summary.txt; tail -2 $O/tests.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python scripts/long_fuzz.py > $O/fuzz.log 2>&1; rc=$?; echo "long fuzz rc=$rc" | tee -a $O/summary.txt; tail -1 $O/fuzz.log | cut -c1-300 | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS|identical|checked" | tee -a $O/summary.txt; }
run "10 kb, 4608 pairs" timeout -k 10 200 python scripts/long_read_bench.py 4608 230 10000 0 --seconds 4 &&
run "20 kb, 1152 pairs" timeout -k 10 200 python scripts/long_read_bench.py 1152 230 20000 2 --seconds 4 &&
run "20 kb, 256 pairs, the workgroup kernel (before)" MGL_STRIP=1 timeout -k 10 200 python scripts/long_read_bench.py 256 230 20000 0 --seconds 4 &&
run "30 kb, 768 pairs" timeout -k 10 300 python scripts/long_read_bench.py 768 230 30000 1 --seconds 4 &&
run "30 kb, 128 pairs, the workgroup kernel (before)" MGL_STRIP=1 timeout -k 10 300 python scripts/long_read_bench.py 128 230 30000 0 --seconds 4
