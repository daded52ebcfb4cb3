#!/bin/bash
# Same-box A/B of two builds of the library: i2v-adapter-unofficial_amd/libi2v_hip_{A,B}.so (boxes differ by +-2 %).
# usage (GPU box, repo root): bash tools/ab_bench.sh [rounds]
P=i2v-adapter-unofficial_amd
for r in $(seq 1 ${1:-2}); do
  for v in A B; do
    cp $P/libi2v_hip_$v.so $P/libi2v_hip.so
    python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],3), round(d['ms_per_step'],3), {k: v['ms'] for k, v in d['kernel_classes'].items() if v['ms'] > 1})"
  done
done
