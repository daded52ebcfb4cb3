#!/bin/bash
# Build the library of another git revision into .ab_libs/NAME.so for a same-box A/B against the working tree
# (tools/ab_bench.sh .ab_libs/NAME.so i2v-adapter-unofficial_amd/libi2v_hip.so).   usage: bash tools/build_ref.sh REV NAME
set -e
cd "$(dirname "$0")/.."
rev=$1; name=$2
tmp=$(mktemp -d /tmp/i2v_ref_XXXX)
git archive "$rev" i2v-adapter-unofficial_amd/csrc include | tar -x -C "$tmp"
mkdir -p .ab_libs "$tmp/obj"
for f in "$tmp"/i2v-adapter-unofficial_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -fno-honor-nans -c "$f" -o "$tmp/obj/$(basename "${f%.hip}").o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o .ab_libs/$name.so "$tmp"/obj/*.o
rm -rf "$tmp"
echo ".ab_libs/$name.so"
