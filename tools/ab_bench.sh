#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by +-2 %): the variant is selected through I2V_LIB_PATH, the
# in-tree libi2v_hip.so is never touched.
# usage (GPU box, repo root): bash tools/ab_bench.sh path/to/libA.so path/to/libB.so [rounds] [extra bench.py args]
set -e
cd "${GRAFT_REPO_ROOT:?}"
A=$1; B=$2; rounds=${3:-2}; shift $(( $# < 3 ? $# : 3 ))
mkdir -p gpurun_out
for r in $(seq 1 "$rounds"); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    I2V_LIB_PATH="$(realpath "$lib")" python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],3), round(d['ms_per_step'],3), {k: v['ms'] for k, v in d['kernel_classes'].items() if v['ms'] > 1})"
  done
done
