#!/bin/bash
# Builds build/libmmfe_<name>.so with extra compiler flags (kernel tuning / diagnostics).
# usage: scripts/build_variant.sh <name> [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/$NAME
for f in mmfe_api fbank512 stft32 dftgen rfft400 frame_stats resample; do
  EXTRA=""; [ $f = fbank512 ] && EXTRA="-mllvm -amdgpu-load-store-vectorizer=0"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -fno-slp-vectorize $EXTRA "$@" -Iinclude -Imulti_modal_amd/csrc -c multi_modal_amd/csrc/$f.hip -o build/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmmfe_$NAME.so build/$NAME/*.o
echo built build/libmmfe_$NAME.so
