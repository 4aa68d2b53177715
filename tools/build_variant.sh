#!/bin/bash
# A/B builds of ONE kernel file: tools/build_variant.sh <name> <file.hip> [-DFLAG=..]...
#   -> vision-transformer-opencl_amd/build/ab/libvit_mi355x_<name>.so = the current objects with that file recompiled with the flags.
# Run after `make` (it links the other objects as they are); use with tools/gpu_session.sh ab:/abtest: steps.
set -e
cd "$(dirname "$0")/../vision-transformer-opencl_amd"
name=$1; src=$2; shift 2
base=$(basename "$src" .hip)
extra=""; [ "$base" = vit_attention_stream ] && extra="-fno-slp-vectorize"
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Wno-unused-value -Wno-inline-asm $extra "$@" -c "csrc/$base.hip" -o "build/ab/${base}_$name.o"
objs=$(ls build/*.o | grep -v "/$base.o\|vit_main.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/ab/libvit_mi355x_$name.so" $objs "build/ab/${base}_$name.o" -lm -lgomp -lpthread
echo "build/ab/libvit_mi355x_$name.so"
