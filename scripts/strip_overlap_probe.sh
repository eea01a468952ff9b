#!/bin/bash
# round 4: does the walk of one chunk hide under the fill of the next now that the 20-row kernel leaves registers for a walk wave per SIMD?
set -o pipefail
O=gpurun_out/${1:-r04_overlap}; mkdir -p $O
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS" | tee -a $O/summary.txt; }
run "10 kb, 2304 pairs, one chunk (64 GiB)" timeout -k 10 200 python scripts/long_read_bench.py 2304 64 10000 0 --seconds 5 &&
run "10 kb, 2304 pairs, chunks of 768 (24 GiB)" timeout -k 10 200 python scripts/long_read_bench.py 2304 24 10000 0 --seconds 5 &&
run "10 kb, 2304 pairs, chunks of 1536 (36 GiB)" timeout -k 10 200 python scripts/long_read_bench.py 2304 36 10000 0 --seconds 5 &&
run "10 kb, 4608 pairs, chunks of 768 (24 GiB)" timeout -k 10 200 python scripts/long_read_bench.py 4608 24 10000 0 --seconds 5 &&
run "10 kb, 4608 pairs, one chunk (128 GiB)" timeout -k 10 200 python scripts/long_read_bench.py 4608 128 10000 0 --seconds 5
