#!/bin/bash
# Builds build/libmmfe_<name>.so: fbank512.hip compiled with the extra flags, every other object taken from the
# regular in-tree build (multi_modal_amd/csrc/build, made by `make -C multi_modal_amd/csrc`).
# usage: scripts/build_fbank_variant.sh <name> [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -fno-slp-vectorize -Wall -Wno-unused-function -mllvm -amdgpu-load-store-vectorizer=0 "$@" \
    -Iinclude -Imulti_modal_amd/csrc -Rpass-analysis=kernel-resource-usage -c multi_modal_amd/csrc/fbank512.hip -o build/$NAME/fbank512.o 2> build/$NAME/remarks.txt || { cat build/$NAME/remarks.txt | grep -v remark; exit 1; }
B=multi_modal_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmmfe_$NAME.so build/$NAME/fbank512.o $B/mmfe_api.o $B/stft32.o $B/dftgen.o $B/rfft400.o $B/frame_stats.o $B/resample.o
grep -A8 "fbank512_kernelILb0ELi13ELb1ELb1ELi2ELi7E" build/$NAME/remarks.txt | grep -E "VGPRs:|Occupancy|Spill|ScratchSize" | sed "s/.*remark: *//" | tr '\n' ' '
echo; echo built build/libmmfe_$NAME.so
