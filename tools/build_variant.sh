#!/bin/bash
# build an alternate engine library for same-box A/B timing: tools/build_variant.sh <name> [extra hipcc flags...]
# -> build/ab/libfreud_sae_<name>.so   (compare with tools/ab_bench.sh)
set -e
name=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -shared -o build/ab/libfreud_sae_$name.so freud_amd/csrc/engine.hip -L/opt/rocm/lib -lrccl
echo build/ab/libfreud_sae_$name.so
