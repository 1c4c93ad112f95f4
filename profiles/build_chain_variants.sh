#!/bin/bash
# Timing-only variants of csrc/mlp_chain.hip (-DCHAIN_ABL=<mask>) linked with the other objects into
# build_variants/libvkas_chain<mask>.so; compare them with profiles/bench_chain.py.
set -e
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
mkdir -p ../../build_variants
for abl in "$@"; do
  objs=""
  for f in *.hip; do
    o=build/${f%.hip}.o
    if [ "$f" = mlp_chain.hip ]; then
      o=../../build_variants/mlp_chain_abl${abl//[=,]/_}.o
      # "trace": phase timestamps (profiles/trace_chain.py); NAME=VALUE[,NAME=VALUE]: -DNAME=VALUE ...; a number: -DCHAIN_ABL=<mask>
      case "$abl" in
        trace) def="-DCHAIN_TRACE" ;;
        *=*) def="-D${abl//,/ -D}" ;;
        *) def="-DCHAIN_ABL=$abl" ;;
      esac
      hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value $def -c $f -o $o
    fi
    objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/gemm_mfma_f16.o -o ../../build_variants/libvkas_chain${abl//[=,]/_}.so
done
