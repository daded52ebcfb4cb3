#!/bin/bash
# PMC counters of the head_dim 40 attention kernel in isolation (tools/attn_only.py): busy / wait / instruction mix and
# LDS bank conflicts, in two passes (8 SQ counters each).  Output: gpurun_out/<tag>_{a,b}/ + a printed summary.
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
tag=${1:-pmc_attn}
export TMPDIR=/tmp
GRP=${GRP:-16} ITERS=3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${tag}_a -- python tools/attn_only.py > /dev/null 2>&1
GRP=${GRP:-16} ITERS=3 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/${tag}_b -- python tools/attn_only.py > /dev/null 2>&1
python - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in (tag + "_a", tag + "_b"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("none", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "attn_kernel" not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:70], r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in v.items():
            print("   ", c, x / cnt[(k, c)])
PY
