#!/bin/bash
# Development aid: another build of the library with -D overrides, for A/B runs (RDST_HIP_LIB=tools/_build/librdst_<tag>.so).
# usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -I $ROOT/include "$@" \
  $ROOT/rdst_amd/csrc/rdst_kernels.hip $ROOT/rdst_amd/csrc/rdst_tuner.cpp $ROOT/rdst_amd/csrc/rdst_regions.cpp -o $ROOT/tools/_build/librdst_$TAG.so
echo built $TAG
