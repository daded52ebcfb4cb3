#!/bin/bash
# Per-kernel PMC summary of ONE eager denoising step of bench.py's default workload (16 f x 512^2): matrix-pipe busy,
# VALU / LDS activity, wait shares and LDS bank conflicts of every kernel, plus rocprof's own kernel durations.
# Counters are collected in passes of <= 8 SQ counters with --kernel-trace only (no other trace domains), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.   usage (GPU box): bash tools/pmc_step.sh [tag]
# -> gpurun_out/<tag>_{a,b}/ and gpurun_out/<tag>_summary.txt (copy the summary to profiles/)
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
tag=${1:-pmc_step}
export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_a -- $CMD > gpurun_out/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/${tag}_b -- $CMD > gpurun_out/${tag}_b.log 2>&1
python tools/summarize_pmc.py gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_summary.txt
# the raw per-dispatch CSVs are tens of MB (gpurun merges at most 64 MiB back): keep the summary only
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b
