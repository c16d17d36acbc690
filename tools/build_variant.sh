#!/bin/bash
# A/B partner of the in-tree library: conv.hip compiled with extra -D flags, linked with the in-tree objects of the
# other sources -> tools/_build/libcurla_<tag>.so (load it with CURLA_LIB_PATH).  Usage: tools/build_variant.sh <tag> -DX=1 ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC "$@" -c curla_amd/csrc/conv.hip -o tools/_build/conv_$TAG.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libcurla_$TAG.so tools/_build/conv_$TAG.o \
  curla_amd/csrc/gemm.o curla_amd/csrc/heads.o curla_amd/csrc/augment.o curla_amd/csrc/options.o
echo tools/_build/libcurla_$TAG.so
