#!/bin/bash
# Build an experimental variant of libnvx_hip.so with extra compiler defines for ONE source file:
#   tools/build_variant.sh NAME prefill_attn "-DPF_PRIO=1"   ->  nano_vllm_x_amd/csrc/build/var_NAME.so
# A GPU command then copies it over nano_vllm_x_amd/libnvx_hip.so in the box's scratch copy of the tree.
set -e
cd "$(dirname "$0")/../nano_vllm_x_amd/csrc"
name=$1; src=$2; defs=$3
mkdir -p build/var_$name
[ "$src" = prefill_attn ] && defs="$defs -fno-slp-vectorize"   # as the Makefile builds it
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include $defs -c $src.hip -o build/var_$name/$src.o
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  if [ "$b" = "$src" ]; then objs="$objs build/var_$name/$src.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o build/var_$name.so
echo build/var_$name.so
