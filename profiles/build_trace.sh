#!/bin/bash
# libvkas with phase timestamps in gemm_nt_mfma_kernel (-DVKAS_TRACE) -> build_variants/libvkas_trace.so; read with
# profiles/trace_nt.py.  Never shipped: the instrumented kernel waits for its stores before it exits.
set -e
cd "$(dirname "$0")/../vkit_ocr_model_adaptive_scaling_amd/csrc"
make -s
mkdir -p ../../build_variants
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DVKAS_TRACE -c gemm_mfma.hip -o ../../build_variants/gemm_mfma_trace.o
objs="../../build_variants/gemm_mfma_trace.o build/gemm_mfma_f16.o"
for f in *.hip; do
  [ "$f" = gemm_mfma.hip ] || objs="$objs build/${f%.hip}.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build_variants/libvkas_trace.so
