#!/bin/bash
# PMC summary of the K = 320 GEMM flavours of tools/ws_probe.py (weight-stationary kernel, or the 8-wave tile kernel under
# I2V_GEMM_WS=0): matrix-pipe busy, VALU / LDS activity, wait shares, LDS bank conflicts.   usage: bash tools/pmc_ws.sh [tag]
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
tag=${1:-pmc_ws}
export TMPDIR=/tmp
CMD="python tools/ws_probe.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_a -- $CMD > gpurun_out/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/${tag}_b -- $CMD > gpurun_out/${tag}_b.log 2>&1
python tools/summarize_pmc.py gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_summary.txt
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b
