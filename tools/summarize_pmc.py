"""Fold two rocprofv3 --pmc passes (tools/pmc_step.sh) into one per-kernel table: average duration, matrix-pipe busy,
VALU / LDS activity, wait shares, LDS bank conflicts.

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (16 per v_mfma_f32_16x16x32_f16); GRBM_GUI_ACTIVE is
summed over the 8 XCDs, so chip SIMD-cycles of a dispatch = 1024 SIMDs x GRBM_GUI_ACTIVE / 8.
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)        fraction of all SIMD-cycles with an MFMA executing
  valu/lds/wait shares = counter / SQ_WAVE_CYCLES                          per resident wave
  conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                    fraction of LDS-array cycles lost to conflicts
usage: python tools/summarize_pmc.py <pass A dir> <pass B dir> <out.txt> [kernel-name filter]"""
import collections, csv, glob, re, sys

a_dir, b_dir, out_path = sys.argv[1:4]
flt = sys.argv[4] if len(sys.argv) > 4 else ""


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:64]


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if f:
        for r in csv.DictReader(open(f[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(float)
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if t:
        for r in csv.DictReader(open(t[0])):
            dur[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    return agg, {k: len(v) for k, v in disp.items()}, dur


A, na, dur = load(a_dir)
B, nb, _ = load(b_dir)
rows = []
for k in A:
    if flt not in k or "spin_kernel" in k:
        continue
    a, b = A[k], B.get(k, {})
    wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    gui = max(a.get("GRBM_GUI_ACTIVE", 0.0), 1.0)
    rows.append(dict(
        k=k, calls=na[k], us=dur[k] / max(na[k], 1), tot_ms=dur[k] / 1e3,
        mfma=a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * gui) if "GRBM_GUI_ACTIVE" in a else float("nan"),
        valu=a.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, lds=a.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
        wait=a.get("SQ_WAIT_ANY", 0.0) / wc, stall=a.get("SQ_WAIT_INST_ANY", 0.0) / wc,
        conf=b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(b.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
        vpm=b.get("SQ_INSTS_VALU", 0.0) / max(b.get("SQ_INSTS_MFMA", 0.0), 1.0) if b.get("SQ_INSTS_MFMA") else float("nan")))
rows.sort(key=lambda r: -r["tot_ms"])
with open(out_path, "w") as o:
    o.write("# rocprofv3 --pmc summary per kernel (tools/pmc_step.sh; one eager step + warm-ups of bench.py's default workload)\n")
    o.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE); valu / lds / wait / stall = share of SQ_WAVE_CYCLES\n")
    o.write("# (ACTIVE_INST_VALU, ACTIVE_INST_LDS, WAIT_ANY = s_waitcnt + barrier, WAIT_INST_ANY = issue stall); lds_conflict =\n")
    o.write("# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; valu/mfma = VALU instructions per MFMA instruction\n")
    o.write(f"{'kernel':64s} {'calls':>6s} {'avg_us':>8s} {'total_ms':>9s} {'mfma_busy':>9s} {'valu':>6s} {'lds':>6s} "
            f"{'wait':>6s} {'stall':>6s} {'lds_conflict':>12s} {'valu/mfma':>9s}\n")
    for r in rows[:28]:
        o.write(f"{r['k']:64s} {r['calls']:6d} {r['us']:8.1f} {r['tot_ms']:9.2f} {r['mfma']:9.3f} {r['valu']:6.3f} "
                f"{r['lds']:6.3f} {r['wait']:6.3f} {r['stall']:6.3f} {r['conf']:12.4f} {r['vpm']:9.2f}\n")
if out_path != "/dev/stdout":
    print(open(out_path).read())
