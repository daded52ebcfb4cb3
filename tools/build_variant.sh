#!/bin/bash
# A variant build of the library for same-box A/B runs: recompiles the named sources with extra flags and links them
# with the in-tree objects of the others into .ab_libs/NAME.so (select it with I2V_LIB_PATH; the in-tree library and
# objects are untouched).
# usage: bash tools/build_variant.sh NAME "extra hipcc flags" file.hip [file.hip ...]
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; shift 2
python __graft_entry__.py > /dev/null          # in-tree objects up to date
csrc=i2v-adapter-unofficial_amd/csrc
mkdir -p .ab_libs/obj_$name
objs=()
for o in $csrc/build/*.o; do
  b=$(basename "$o" .o); keep=1
  for f in "$@"; do [ "$f" = "$b.hip" ] && keep=0; done
  [ $keep = 1 ] && objs+=("$o")
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -fno-honor-nans $extra -c $csrc/$f -o .ab_libs/obj_$name/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o .ab_libs/$name.so "${objs[@]}" .ab_libs/obj_$name/*.o
echo ".ab_libs/$name.so"
