#!/bin/bash
# Same-box A/B of one environment switch: bash tools/ab_env.sh "I2V_ATTN_32=0" [rounds [bench.py arguments ...]]
# (A = with the setting, B = default)
sw=$1; rounds=${2:-2}; shift; shift
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then export $sw; else unset ${sw%%=*}; fi
    python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],3), round(d['ms_per_step'],3), {k: v['ms'] for k, v in d['kernel_classes'].items() if v['ms'] > 1})"
  done
done
