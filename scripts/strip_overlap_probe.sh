#!/bin/bash
# round 4: long reads by length (strips of 17 .. 19 rows for targets of 8 193 .. 9 728 rows), after the long suites
set -o pipefail
O=gpurun_out/${1:-r04_len}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "long or strip" > $O/tests.log 2>&1; rc=$?; echo "long tests rc=$rc" | tee -a $O/summary.txt; tail -2 $O/tests.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
run() { label=$1; shift; echo "== $label" | tee -a $O/summary.txt; env "$@" 2>&1 | grep -E "pairs of|GCUPS|identical|checked" | tee -a $O/summary.txt; }
for L in 8300 8800 9300 9700; do
  n=$(python3 -c "print(max(768, int(4608 * (10000/$L)**2) // 768 * 768))")
  run "$L, $n pairs" timeout -k 10 200 python scripts/long_read_bench.py $n 230 $L 2 --seconds 4 || exit 1
done
