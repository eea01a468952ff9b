#!/bin/bash
# Experiment builds of ONE kernel file, made here (hipcc cross-compiles) and shipped to the GPU box with the snapshot:
#   bash scripts/build_variant.sh NAME FILE.hip "-DFLAG ..."   ->  build/variants/lib_NAME.so   (MGL_SW_LIB selects it)
# The other objects are the regular build's (build/obj/, `make -C mgl_amd/csrc` first).
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; flags=$3
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DMGL_VARIANT_BUILD $flags -c -o build/variants/$name.o mgl_amd/csrc/$file
objs=$(ls build/obj/sw_*.o build/obj/jni_exports.cpp.o | grep -v "/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/variants/lib_$name.so build/variants/$name.o $objs
rm -f build/variants/$name.o
echo build/variants/lib_$name.so
