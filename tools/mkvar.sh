#!/bin/bash
# builds exp/libexp_NAME.so with extra -D flags (compile-time variants for tools/ab.sh / ACE_SFNO_LIB)
# usage: mkvar.sh NAME -DFLAG=.. ...   -> exp/libexp_NAME.so
name=$1; shift
cd "$(dirname "$0")/.."; mkdir -p exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o exp/libexp_$name.so ace_amd/csrc/kernels.hip ace_amd/csrc/fft.hip ace_amd/csrc/strip.hip ace_amd/csrc/mlp_strip.hip ace_amd/csrc/conv_strip.hip ace_amd/csrc/conv_split.hip ace_amd/csrc/conv_ws.hip ace_amd/csrc/dhconv_strip.hip ace_amd/csrc/capi.hip ace_amd/csrc/tables.cpp && echo built exp/libexp_$name.so
