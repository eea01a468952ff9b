#!/bin/bash
# sw_dp16_lane_ck_kernel with four waves per workgroup (build ck_wpb4: bash scripts/build_variant.sh ck_wpb4 sw_dp16_lane_ck.hip -DMGL_CK_WAVES_PER_BLOCK=4)
# against the regular build (one): a workgroup's registers are released when its LAST wave ends, and the waves of one workgroup need
# different numbers of block rounds.  Alternating runs on one box (boxes of the pool differ by more than the effect).
for rep in 1 2; do
for v in "" ck_wpb4; do
  lib=""; [ -n "$v" ] && lib=build/variants/lib_$v.so
  for n in 10000000 1250000; do
    echo "== ${v:-regular (1 wave per workgroup)}, $n pairs"
    MGL_SW_LIB=$lib python3 bench.py --steps 5 --warmup 1 --no-cpu --no-secondary --no-extra --pairs $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'GCUPS', d['ms_per_step'], 'ms')"
  done
done
done
