#!/bin/bash
# round 4: long reads, quick rates
set -o pipefail
O=gpurun_out/${1:-r04_len}; mkdir -p $O
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS|identical|checked" | tee -a $O/summary.txt; }
run "10 kb, 4608 pairs" timeout -k 10 200 python scripts/long_read_bench.py 4608 230 10000 2 --seconds 5 &&
run "10 kb, 2304 pairs" timeout -k 10 200 python scripts/long_read_bench.py 2304 230 10000 0 --seconds 5 &&
run "9000, 5376 pairs" timeout -k 10 200 python scripts/long_read_bench.py 5376 230 9000 0 --seconds 4
