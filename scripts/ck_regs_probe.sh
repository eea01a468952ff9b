#!/bin/bash
# round 4: which of the register changes of the headline kernel's pass 2 cost time (variants built from temp copies of the source:
# A = all but the record read-back, B = all but the per-tile opaque slot, C = all but the serialised checkpoint loads, D = before +
# opaque slot only, E = before + 32-bit byte offsets only; prevck = before)
set -o pipefail
O=gpurun_out/${1:-r04_ckregs}; mkdir -p $O
for rep in 1 2; do
  for v in "" prevck ckA ckB ckC ckD ckE; do
    echo "== ${v:-this build}" | tee -a $O/summary.txt
    if [ -z "$v" ]; then FULL_ONLY=1 REPS=10 timeout -k 10 200 python scripts/lane_probe.py 10000000 2>&1 | grep GCUPS | cut -c1-70 | tee -a $O/summary.txt || exit 1
    else MGL_SW_LIB=build/variants/lib_$v.so FULL_ONLY=1 REPS=10 timeout -k 10 200 python scripts/lane_probe.py 10000000 2>&1 | grep GCUPS | cut -c1-70 | tee -a $O/summary.txt || exit 1; fi
  done
done
