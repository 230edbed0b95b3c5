#!/bin/bash
# Builds build/libmmfe_<name>.so: ONE source of multi_modal_amd/csrc (fbank512 | rfft400 | stft32 | dftgen | ...) compiled with the
# extra flags, every other object taken from the regular in-tree build (made by `make -C multi_modal_amd/csrc`).
# usage: scripts/build_variant_of.sh <source> <name> [extra hipcc flags...]
# SRCFILE=<path> compiles that file in place of multi_modal_amd/csrc/<source>.hip (an experimental copy kept outside the tree's sources).
set -e
cd "$(dirname "$0")/.."
SRC=$1; NAME=$2; shift; shift
mkdir -p build/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -fno-slp-vectorize -Wall -Wno-unused-function "$@" \
    -Iinclude -Imulti_modal_amd/csrc -Rpass-analysis=kernel-resource-usage -c ${SRCFILE:-multi_modal_amd/csrc/$SRC.hip} -o build/$NAME/$SRC.o 2> build/$NAME/remarks.txt || { grep -v remark build/$NAME/remarks.txt; exit 1; }
B=multi_modal_amd/csrc/build
OBJS=""
for f in mmfe_api fbank512 stft32 dftgen rfft400 frame_stats resample patch_embed; do
  if [ $f = $SRC ]; then OBJS="$OBJS build/$NAME/$SRC.o"; else OBJS="$OBJS $B/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmmfe_$NAME.so $OBJS
echo built build/libmmfe_$NAME.so
