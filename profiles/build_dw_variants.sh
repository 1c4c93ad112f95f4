#!/bin/bash
# Timing-only variants of csrc/dwconv.hip (-DDW_ABL=<mask>) into build_variants/libvkas_dw<mask>.so; run
# VKAS_LIB_PATH=build_variants/libvkas_dw<mask>.so python profiles/bench_dw.py
set -e
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
mkdir -p ../../build_variants
for abl in "$@"; do
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = dwconv.hip ]; then
      o=../../build_variants/dwconv_abl$abl.o
      flags="-DDW_ABL=$abl"
      [ "$abl" = trace ] && flags="-DDW_TRACE"
      hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value $flags -c $f -o $o
    fi
    objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/gemm_mfma_f16.o -o ../../build_variants/libvkas_dw$abl.so
done
