#!/bin/bash
# Variant builds of the fused SPADE kernel for same-box A/B: tools/exp/libfvar_<name>.so = the shipped library with only
# spade_fused.hip rebuilt with the given -D flags.  usage: build_fused_var.sh <name> <flags...>
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../../deepsee_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form "$@" -c spade_fused.hip -o /tmp/fused_var_$name.o
objs=$(ls build/*.o | grep -v spade_fused.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fused_var_$name.o -o ../../tools/exp/libfvar_$name.so
echo "built tools/exp/libfvar_$name.so"
