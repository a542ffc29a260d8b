#!/bin/bash
# Build libdeepsee_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdeepsee_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=()
for f in *.hip *.cpp; do
  [ -e "$f" ] || continue
  o="build/${f%.*}.o"
  mkdir -p build
  if [ ! -e "$o" ] || [ "$f" -nt "$o" ] || [ dsee_common.h -nt "$o" ] || [ dsee_rng.h -nt "$o" ] || [ spade_fused_args.h -nt "$o" ] || [ ../../include/deepsee_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    # spade_fused.hip keeps its 256 output accumulators in AGPRs by hand: the MFMA results must then live in VGPRs
    EXTRA=""
    if [[ "$f" == spade_fused.hip || "$f" == spade_fused_w4.hip ]]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
    if [[ "$f" == *.hip ]]; then hipcc $FLAGS $EXTRA -c "$f" -o "$o"; else hipcc $FLAGS -x hip -c "$f" -o "$o"; fi
  fi
  OBJS+=("$o")
done
hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "$OUT"
echo "built $(realpath $OUT)"
