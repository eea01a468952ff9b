#!/bin/bash
# Timing-only diagnostic builds of sw_dp_kernel (results are WRONG in ablated builds; only the fill-kernel
# time is read).  Usage on the GPU box: bash scripts/ablate.sh "NAME:-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o gpurun_out/ablate/lib_$name.so \
      mgl_amd/csrc/sw_kernels.hip mgl_amd/csrc/sw_dp16.hip -x hip mgl_amd/csrc/sw_capi.cpp mgl_amd/csrc/sw_batcher.cpp mgl_amd/csrc/jni_exports.cpp
done
for spec in "$@"; do
  name="${spec%%:*}"
  MGL_SW_LIB=$PWD/gpurun_out/ablate/lib_$name.so python scripts/time_kernels.py "$name" ${PAIRS:-2000000}
done
