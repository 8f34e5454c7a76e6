#!/bin/bash
# Build an experiment variant of libofx.so: conv.hip recompiled with extra flags, every other object taken from the in-tree build.
#   tools/build_variant.sh <name> [hipcc flags, e.g. -DOFX_CONV_LEAN -DOFX_EXP_...]   ->  tools/variants/libofx_<name>.so
# tools/conv_bench.py takes the variants as arguments; OFX_LIB_PATH=... points the package at one.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/sd_animation_optical_flow_amd/csrc
mkdir -p $R/tools/variants /tmp/ofxvar
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $C/conv.hip -o /tmp/ofxvar/conv_$name.o \
    -Rpass-analysis=kernel-resource-usage 2> /tmp/ofxvar/conv_$name.rpass || { tail -20 /tmp/ofxvar/conv_$name.rpass; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libofx_$name.so /tmp/ofxvar/conv_$name.o $(ls $C/*.o | grep -v '/conv.o')
echo built tools/variants/libofx_$name.so
