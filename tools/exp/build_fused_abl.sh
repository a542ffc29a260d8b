#!/bin/bash
# Measurement builds of the fused SPADE kernel with parts removed (DSEE_FUSED_ABL bit mask, see spade_fused.hip):
# tools/exp/libfabl_<mask>.so = the shipped library with only spade_fused.hip rebuilt.
set -euo pipefail
cd "$(dirname "$0")/../../deepsee_amd/csrc"
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DDSEE_FUSED_ABL=$m -c spade_fused.hip -o /tmp/fused_abl_$m.o
  objs=$(ls build/*.o | grep -v spade_fused.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fused_abl_$m.o -o ../../tools/exp/libfabl_$m.so
  echo "built tools/exp/libfabl_$m.so"
done
