#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <-DFLAG ...>: libdam_hip.so with ONE source recompiled with extra flags
# (diagnostic builds for timing experiments; load with DAM_LIB_PATH=tools/libdam_<name>.so)
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
b=$root/deep-audio-mixer_amd/csrc/_build
objs=""
for o in $b/*.o; do case $o in *.diag.o) continue;; esac; [ "$(basename $o .o)" = "$(basename $src .hip)" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form=1 -I $root/include -I $root/deep-audio-mixer_amd/csrc "$@" -c $root/deep-audio-mixer_amd/csrc/$src -o /tmp/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/libdam_$name.so $objs /tmp/var_$name.o
echo built tools/libdam_$name.so
