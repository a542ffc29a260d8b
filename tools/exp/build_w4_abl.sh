#!/bin/bash
# Measurement builds of csrc/gemm_w4.hip: tools/exp/libw4_<name>.so = the shipped library with only gemm_w4.hip rebuilt with the
# given -D flags.  usage: build_w4_abl.sh name "-DDSEE_W4_ABL=2" [name2 "flags2" ...]
set -euo pipefail
cd "$(dirname "$0")/../../deepsee_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c gemm_w4.hip -o /tmp/gemm_w4_$name.o
  objs=$(ls build/*.o | grep -v gemm_w4.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_w4_$name.o -o ../../tools/exp/libw4_$name.so
  echo "built tools/exp/libw4_$name.so"
done
