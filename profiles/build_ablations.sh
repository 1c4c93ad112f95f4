#!/bin/bash
# Builds timing-only variants of libvkas.so (VKAS_ABL bit masks, see csrc/gemm_mfma.hip) into build_variants/.
set -e
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
mkdir -p ../../build_variants
for abl in "$@"; do
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = gemm_mfma.hip ]; then
      o=../../build_variants/gemm_mfma_abl$abl.o
      hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DVKAS_ABL=$abl -c $f -o $o
    fi
    objs="$objs $o"
  done
  objs="$objs build/gemm_mfma_f16.o"
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build_variants/libvkas_abl$abl.so
done
