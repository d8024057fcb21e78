#!/bin/bash
# Diagnostic library for tools/gemm_stamps.py: the shipped objects + gemm_bf16x3_v2.hip rebuilt with -DV2_TIMING.
set -e
cd "$(dirname "$0")/../index-tts_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DV2_TIMING $EXTRA \
  -c gemm_bf16x3_v2.hip -o /tmp/gemm_bf16x3_v2_timing.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls build/*.o | grep -v gemm_bf16x3_v2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libidxtts_timing${SUFFIX}.so $OBJS /tmp/gemm_bf16x3_v2_timing.o
echo built tools/libidxtts_timing${SUFFIX}.so
