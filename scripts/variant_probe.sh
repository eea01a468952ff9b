#!/bin/bash
# times every build/variants/lib_*.so given (and the regular build first) on the lane probe: bash scripts/variant_probe.sh [pairs] name...
n=${1:-2097152}; shift
export LANE_MODE=0 FULL_ONLY=1
echo -n "regular: "; timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*"
for v in "$@"; do echo -n "$v: "; MGL_SW_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python scripts/lane_probe.py $n 2>/dev/null | grep -o "fill kernel.*"; done
