#!/bin/bash
# launch-size sweep of the checkpointed lane kernel (one launch each): the per-rank workloads of N = 8 / 4 / 2 GPUs and the chip's rounds
export FULL_ONLY=1 LANE_MODE=0
for n in 393216 786432 1250000 2500000 5000000 10000000; do echo -n "n=$n: "; timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "per call = [0-9]* GCUPS"; done
