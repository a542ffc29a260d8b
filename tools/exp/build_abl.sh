#!/bin/bash
# Measurement builds of the GEMM kernels with parts of the main loop removed (DSEE_GEMM_ABL bit mask, see
# deepsee_amd/csrc/gemm_bf16x3.hip): tools/exp/libabl_<mask>.so = the shipped library with only gemm_bf16x3.hip rebuilt.
set -euo pipefail
cd "$(dirname "$0")/../../deepsee_amd/csrc"
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDSEE_GEMM_ABL=$m -c gemm_bf16x3.hip -o /tmp/gemm_abl_$m.o
  objs=$(ls build/*.o | grep -v gemm_bf16x3.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_abl_$m.o -o ../../tools/exp/libabl_$m.so
  echo "built tools/exp/libabl_$m.so"
done
