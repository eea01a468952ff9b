#!/bin/bash
# round 4: the strip kernels -- parity of the long suites, then 10 kb / 16 kb pairs on this build, on the build before the spill work and
# the base-code form (build/variants/lib_prevstrip.so), with the byte compare forced, and where two waves per SIMD start to pay
# (scripts/build_variant.sh strip_from26 sw_dp16_strip.hip -DMGL_STRIP_OCC2_FROM=26; strip_from33: never)
set -o pipefail
O=gpurun_out/${1:-r04_strip}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "long or strip" > $O/tests.log 2>&1; rc=$?; echo "long tests rc=$rc" | tee -a $O/summary.txt; tail -2 $O/tests.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
run() { # label, env..., -- args
  label=$1; shift
  echo "== $label" | tee -a $O/summary.txt
  env "$@" 2>&1 | grep -E "pairs of|GCUPS" | tee -a $O/summary.txt
}
run "10 kb, this build" timeout -k 10 200 python scripts/long_read_bench.py 2304 64 10000 0 --seconds 5 &&
run "10 kb, byte compare" MGL_SW_DEBUG_STRIP_CODES=0 timeout -k 10 200 python scripts/long_read_bench.py 2304 64 10000 0 --seconds 5 &&
run "10 kb, build before" MGL_SW_LIB=build/variants/lib_prevstrip.so timeout -k 10 200 python scripts/long_read_bench.py 2304 64 10000 0 --seconds 5 &&
run "16 kb, this build" timeout -k 10 200 python scripts/long_read_bench.py 1536 96 16000 0 --seconds 5 &&
run "16 kb, build before" MGL_SW_LIB=build/variants/lib_prevstrip.so timeout -k 10 200 python scripts/long_read_bench.py 1536 96 16000 0 --seconds 5 || exit 1
for L in 13300 14300 14800 15300; do
  run "$L, this build (two waves per SIMD from 30 rows)" timeout -k 10 200 python scripts/long_read_bench.py 1536 96 $L 0 --seconds 4 &&
  run "$L, two waves per SIMD from 26 rows" MGL_SW_LIB=build/variants/lib_strip_from26.so timeout -k 10 200 python scripts/long_read_bench.py 1536 96 $L 0 --seconds 4 &&
  run "$L, three waves per SIMD" MGL_SW_LIB=build/variants/lib_strip_from33.so timeout -k 10 200 python scripts/long_read_bench.py 1536 96 $L 0 --seconds 4 || exit 1
done
