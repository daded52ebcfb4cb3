cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_gemm2 -- python tools/gemm_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc_gemm3 -- python tools/gemm_only.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmc_gemm2","pmc_gemm3"):
    f=glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv",recursive=True)
    if not f: print("none", d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "gemm_big_kernel" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:60],r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(k)
        for c,x in v.items(): print("   ",c, x/cnt[(k,c)])
PY
