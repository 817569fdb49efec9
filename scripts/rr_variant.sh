#!/bin/bash
# A/B builds of the register-resident layer kernel: scripts/rr_variant.sh NAME [-DFLAG ...] -> gsn_amd/lib/variants/libgsn_hip_NAME.so
# (same objects as the product library except layer_rr.o -- or layer_w.o with RR_VARIANT_SRC=layer_w; select with GSN_LIB_PATH)
set -e
cd "$(dirname "$0")/../gsn_amd/csrc"
name=$1; shift
src=${RR_VARIANT_SRC:-layer_rr}          # RR_VARIANT_SRC=layer_w: the wide kernel
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src.hip -o ../lib/variants/${src}_$name.o
objs=$(ls ../lib/obj/*.o | grep -v /$src.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libgsn_hip_$name.so $objs ../lib/variants/${src}_$name.o
echo built gsn_amd/lib/variants/libgsn_hip_$name.so
