#!/bin/bash
# A/B partner of the in-tree library: one source (SRC=conv|gemm|heads|augment, default conv) compiled with extra -D flags,
# linked with the in-tree objects of the other sources -> tools/_build/libcurla_<tag>.so (load it with CURLA_LIB_PATH).
# Usage: [SRC=gemm] tools/build_variant.sh <tag> -DX=1 ...
set -e
TAG=$1; shift
SRC=${SRC:-conv}
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC "$@" -c curla_amd/csrc/$SRC.hip -o tools/_build/${SRC}_$TAG.o
OBJS=""
for s in conv gemm heads augment options; do
  if [ $s = $SRC ]; then OBJS="$OBJS tools/_build/${SRC}_$TAG.o"; else OBJS="$OBJS curla_amd/csrc/$s.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libcurla_$TAG.so $OBJS
echo tools/_build/libcurla_$TAG.so
