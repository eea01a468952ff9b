#!/bin/bash
# round 4: the start stagger of the persistent grid (percent of a tile's time), launch sizes
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
export LANE_MODE=0 FULL_ONLY=1 REPS=8
for n in 1250000 2500000 655360 10000000; do
  for sg in 0 25 50 75; do
    echo -n "pairs $n stagger $sg: " | tee -a $O/summary.txt
    MGL_SW_DEBUG_LANE_STAGGER=$sg timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
  done
  echo -n "pairs $n wave per tile: " | tee -a $O/summary.txt
  WS_GIB=230 MGL_SW_DEBUG_LANE_SLOTS=1000000 timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
done
