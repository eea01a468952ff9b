#!/bin/bash
# round 4: mixed read lengths -- the left-over pairs' kernels beside the bulk's grid (MGL_SW_DEBUG_SIDE_RESERVE wave slots left free; 0: behind it)
set -o pipefail
O=gpurun_out/${1:-r04_mixed}; mkdir -p $O
for r in 0 4 8 16 32 0 8 16 32; do
  echo "== MGL_SW_DEBUG_SIDE_RESERVE=$r" | tee -a $O/summary.txt
  MGL_SW_DEBUG_SIDE_RESERVE=$r WS_GIB=8 timeout -k 10 200 python scripts/grouped_bench.py 4000000 100 2>&1 | grep -E "GCUPS" | grep -v "host batch" | cut -c1-120 | tee -a $O/summary.txt || exit 1
done
