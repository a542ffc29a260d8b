#!/bin/bash
# Measurement builds of csrc/spade_fused_w4.hip: tools/exp/libfw4_<name>.so.  usage: build_fw4.sh name "-DDSEE_FW4_ABL=2" [...]
set -euo pipefail
cd "$(dirname "$0")/../../deepsee_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $flags -c spade_fused_w4.hip -o /tmp/fw4_$name.o
  objs=$(ls build/*.o | grep -v spade_fused_w4.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fw4_$name.o -o ../../tools/exp/libfw4_$name.so
  echo "built tools/exp/libfw4_$name.so"
done
