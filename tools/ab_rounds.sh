#!/bin/bash
# Same-box A/B of two whole TREES (python + library): A = a checked-out earlier round under .ab_old/NAME (git worktree add
# .ab_old/NAME <rev>; python __graft_entry__.py there), B = this tree.   usage (GPU box): bash tools/ab_rounds.sh NAME [rounds]
cd "${GRAFT_REPO_ROOT:?}"
name=$1
for r in $(seq 1 ${2:-2}); do
  for v in A B; do
    dir=.; [ $v = A ] && dir=.ab_old/$name
    (cd $dir && python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],3), round(d['ms_per_step'],3), {k: v['ms'] for k, v in d['kernel_classes'].items() if v['ms'] > 1})"
  done
done
