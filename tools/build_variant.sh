#!/bin/bash
# A variant build of the library for same-box A/B runs: recompiles the named sources with extra flags and links them
# with the in-tree objects of the others into .ab_libs/NAME.so (select it with I2V_LIB_PATH; the in-tree library and
# objects are untouched).
# usage: bash tools/build_variant.sh NAME "extra hipcc flags" file.hip [file.hip ...]
#        bash tools/build_variant.sh --variants
#          -> .ab_libs/variants.so: the default library plus the measured-and-rejected kernel forms that the default build
#             leaves out (csrc/variants/{attention32,attention_pipe,gemm_alt,gemm_ws}.hip and the 4-wave gemm_big
#             instantiations, -DI2V_VARIANTS), selected there by I2V_ATTN32 / I2V_ATTN_PIPE / I2V_GEMM_ALT / I2V_GEMM_4W /
#             I2V_GEMM_WS.
set -e
cd "$(dirname "$0")/.."
csrc=i2v-adapter-unofficial_amd/csrc
if [ "$1" = "--variants" ]; then
  name=variants; extra="-DI2V_VARIANTS"; set -- attention.hip gemm_big.hip gemm.hip variants/attention32.hip variants/attention_pipe.hip variants/gemm_alt.hip variants/gemm_ws.hip
else
  name=$1; extra=$2; shift 2
fi
python __graft_entry__.py > /dev/null          # in-tree objects up to date
mkdir -p .ab_libs/obj_$name
rm -f .ab_libs/obj_$name/*.o
objs=()
for o in $csrc/build/*.o; do
  b=$(basename "$o" .o); keep=1
  for f in "$@"; do [ "$f" = "$b.hip" ] && keep=0; done
  [ $keep = 1 ] && objs+=("$o")
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -fno-honor-nans $extra -c $csrc/$f -o .ab_libs/obj_$name/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o .ab_libs/$name.so "${objs[@]}" .ab_libs/obj_$name/*.o
# the source stamp of the tree this library was built from (tests/test_kernels_gpu.py `_variants_env` checks it)
python - "$name" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
headers = [os.path.join(ge.CSRC, f) for f in os.listdir(ge.CSRC) if f.endswith(".h")] + [os.path.join(ge.ROOT, "include", "i2v_hip.h")]
open(os.path.join(".ab_libs", sys.argv[1] + ".so.stamp"), "w").write(ge._stamp(headers) + "\n")
PY
echo ".ab_libs/$name.so"
