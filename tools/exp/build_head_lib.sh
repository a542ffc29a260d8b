#!/bin/bash
# Same-box A/B of kernel changes: build the library from the COMMITTED sources (git HEAD) into tools/exp/libhead.so; run
# `DSEE_LIB=tools/exp/libhead.so python bench.py ...` against the working tree's library in one gpurun call.
set -euo pipefail
cd "$(dirname "$0")/../.."
rm -rf _head && mkdir -p _head/deepsee_amd/csrc _head/include
git archive HEAD deepsee_amd/csrc include | tar -x -C _head
cd _head/deepsee_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=()
for f in *.hip *.cpp; do
  EXTRA=""; [[ "$f" == spade_fused.hip ]] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
  if [[ "$f" == *.hip ]]; then hipcc $FLAGS $EXTRA -c "$f" -o "${f%.*}.o" & else hipcc $FLAGS -x hip -c "$f" -o "${f%.*}.o" & fi
  OBJS+=("${f%.*}.o")
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o ../../../tools/exp/libhead.so
cd ../../.. && rm -rf _head && ls -la tools/exp/libhead.so
