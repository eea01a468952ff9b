#!/bin/bash
# Timing-only diagnostic builds of sw_dp_kernel (results are WRONG in ablated builds; only the fill-kernel
# time is read).  Usage on the GPU box: bash scripts/ablate.sh "NAME:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  # the Makefile's own source list (every .hip / .cpp of the library), into a side-by-side .so
  make -s -B -C mgl_amd/csrc OUT="$PWD/gpurun_out/ablate/lib_$name.so" EXTRA="$flags" "$PWD/gpurun_out/ablate/lib_$name.so"
done
for spec in "$@"; do
  name="${spec%%:*}"
  MGL_SW_LIB=$PWD/gpurun_out/ablate/lib_$name.so python scripts/time_kernels.py "$name" ${PAIRS:-2000000}
done
