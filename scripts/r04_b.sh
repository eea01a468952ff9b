#!/bin/bash
# round 4, second GPU pass: the persistent grid against a wave per tile (MGL_SW_DEBUG_LANE_SLOTS), launch sizes, sorted callers
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
export LANE_MODE=0 FULL_ONLY=1 REPS=8
for n in 1250000 2500000 10000000; do
  for slots in 0 1000000 1024 1536; do
    for rep in 1 2; do
      echo -n "pairs $n slots $slots: " | tee -a $O/summary.txt
      WS_GIB=230 MGL_SW_DEBUG_LANE_SLOTS=$slots timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
    done
  done
done
timeout -k 10 300 python scripts/grouped_bench.py 4000000 100 > $O/grouped.log 2>&1; echo "grouped rc=$?" | tee -a $O/summary.txt; grep -E "GCUPS|identical" $O/grouped.log | tee -a $O/summary.txt
