#!/bin/bash
# round 4: slow waves leave early at a launch's end (on / off), traced at 1.25 M; the one-pair front-ends after the service changes
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
export LANE_MODE=0 FULL_ONLY=1 REPS=8
for n in 1250000 655360 2500000 10000000 400000; do
  for nee in 0 1 0 1; do
    echo -n "pairs $n no_early_exit=$nee: " | tee -a $O/summary.txt
    MGL_SW_DEBUG_LANE_NO_EARLY_EXIT=$nee timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*" | tee -a $O/summary.txt
  done
done
MGL_SW_LIB=$PWD/build/variants/lib_trace.so timeout -k 10 120 python scripts/ck_trace.py 1250000 2>&1 | grep -v amdgpu | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane or service or native or coalesc or front_end or small or mailbox" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
tail -3 $O/parity.log | tee -a $O/summary.txt
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
