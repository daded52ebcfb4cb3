#!/bin/bash
# PMC counters of the 8-wave GEMM kernel in isolation (tools/gemm_only.py), two passes of 8 SQ counters.
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_gemm2 -- python tools/gemm_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc_gemm3 -- python tools/gemm_only.py > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_gemm2 gpurun_out/pmc_gemm3 /dev/stdout gemm_big_kernel
