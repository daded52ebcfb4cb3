#!/bin/bash
# HBM traffic of one eager denoising step by kernel class, from rocprofv3 PMC counters collected as
# /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one), no trace domains besides --kernel-trace, FETCH_SIZE doubled on gfx950.
# Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh   ->  gpurun_out/traffic.json
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/pmc_$c.log 2>&1
done
python tools/summarize_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/traffic.json
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
