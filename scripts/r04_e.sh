#!/bin/bash
set -o pipefail
O=gpurun_out/r04e; mkdir -p $O
export LANE_MODE=0 FULL_ONLY=1 REPS=8
for n in 1250000 655360 900000 2500000 10000000; do
  for nee in 0 1 0 1; do
    echo -n "pairs $n no_early_exit=$nee: " | tee -a $O/summary.txt
    MGL_SW_DEBUG_LANE_NO_EARLY_EXIT=$nee timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
  done
done
MGL_SW_DEBUG_HOST_TIMING=1 timeout -k 10 300 python scripts/host_packed_probe.py > $O/host_packed.log 2>&1; tail -12 $O/host_packed.log | tee -a $O/summary.txt
