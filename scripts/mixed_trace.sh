#!/bin/bash
# round 4: a mixed-read-length batch (scripts/grouped_bench.py) by workspace, and its kernels under the trace
set -o pipefail
O=gpurun_out/${1:-r04_mixed}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "sorted or mixed or grouped or grouping or promise" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit 1
for ws in 8 72; do
  echo "== WS_GIB=$ws" | tee -a $O/summary.txt
  WS_GIB=$ws timeout -k 10 200 python scripts/grouped_bench.py 4000000 100 2>&1 | grep -E "GCUPS|identical" | cut -c1-220 | tee -a $O/summary.txt || exit 1
done
WS_GIB=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scripts/grouped_bench.py 4000000 100 > $O/mixed.log 2>&1; echo "trace rc=$?" | tee -a $O/summary.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); echo "stats: $f" | tee -a $O/summary.txt
[ -n "$f" ] && head -9 $f | cut -c1-200 | tee -a $O/summary.txt
