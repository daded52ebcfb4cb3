#!/bin/bash
# L1 (TCP) / L2 (TCC) counters of one GEMM shape in isolation (MNK=..., EPI=geglu|none): where the K loop's operand
# delivery is held up.  Few counters per pass (8 TCP/TCC counters in one pass hang rocprofv3 on this pool), each pass
# under its own timeout.
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/pmc_l2
rm -rf $out; mkdir -p $out
i=0
for set in "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" "TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1" "TCC_BUSY TCC_CYCLE TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/gemm_only.py > /dev/null 2>&1; echo "pass $i ($set) rc=$?"
done
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4"):
    f = glob.glob(f"gpurun_out/pmc_l2/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if "gemm_big" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:90]].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(p, k, "dispatches", len(n[k]))
        for c, x in sorted(v.items()): print(f"    {c:36s} {x / len(n[k]):16.0f} per dispatch")
PY
rm -rf $out
