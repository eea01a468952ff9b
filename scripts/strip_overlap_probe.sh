#!/bin/bash
# round 4: four waves per SIMD for the 20-row strip kernel (build/variants/lib_strip_occ4.so; byte compare, so that four workgroups' LDS fits a CU)
set -o pipefail
O=gpurun_out/${1:-r04_len}; mkdir -p $O
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS|identical|checked" | tee -a $O/summary.txt; }
run "10 kb, 6144 pairs, this build" timeout -k 10 200 python scripts/long_read_bench.py 6144 230 10000 0 --seconds 4 &&
run "10 kb, 6144 pairs, this build, byte compare" MGL_SW_DEBUG_STRIP_CODES=0 timeout -k 10 200 python scripts/long_read_bench.py 6144 230 10000 0 --seconds 4 &&
run "10 kb, 6144 pairs, four waves per SIMD, byte compare" MGL_SW_LIB=build/variants/lib_strip_occ4.so MGL_SW_DEBUG_STRIP_CODES=0 timeout -k 10 200 python scripts/long_read_bench.py 6144 230 10000 2 --seconds 4 &&
run "10 kb, 6144 pairs, four waves per SIMD (registers), base codes (LDS: three workgroups)" MGL_SW_LIB=build/variants/lib_strip_occ4.so timeout -k 10 200 python scripts/long_read_bench.py 6144 230 10000 0 --seconds 4
