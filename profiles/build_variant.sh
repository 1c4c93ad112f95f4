#!/bin/bash
# build_variant.sh <name> <flags...>: libvkas.so with gemm_mfma.hip (bf16 build) compiled with the extra flags -> build_variants/libvkas_<name>.so
# (timing-only experiments: schedule variants and VKAS_ABL ablations of the implicit-GEMM kernels; never shipped)
set -e
name=$1; shift
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
mkdir -p ../../build_variants
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value "$@" -c gemm_mfma.hip -o ../../build_variants/gemm_mfma_$name.o
objs=""
for f in *.hip; do
  o=build/${f%.hip}.o
  [ "$f" = gemm_mfma.hip ] && o=../../build_variants/gemm_mfma_$name.o
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/gemm_mfma_f16.o -o ../../build_variants/libvkas_$name.so
echo built build_variants/libvkas_$name.so
