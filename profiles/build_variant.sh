#!/bin/bash
# build_variant.sh <name> <flags...>: libvkas.so with ONE source (default gemm_mfma.hip, bf16 build; SRC=dwconv.hip ... selects
# another) compiled with the extra flags -> build_variants/libvkas_<name>.so
# (timing-only experiments: schedule variants and ablations of the kernels; never shipped)
set -e
name=$1; shift
SRC=${SRC:-gemm_mfma.hip}
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
mkdir -p ../../build_variants
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value "$@" -c $SRC -o ../../build_variants/${SRC%.hip}_$name.o
objs=""
for f in *.hip; do
  o=build/${f%.hip}.o
  [ "$f" = $SRC ] && o=../../build_variants/${SRC%.hip}_$name.o
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/gemm_mfma_f16.o -o ../../build_variants/libvkas_$name.so
echo built build_variants/libvkas_$name.so
