"""Idle time between consecutive kernels of the hipGraph replays, from a rocprofv3 --kernel-trace CSV: histogram of the
start[i + 1] - end[i] gaps over the last N dispatches and the (previous kernel -> next kernel) pairs that own the long ones.
usage: python tools/gap_trace.py path/to/*_kernel_trace.csv [N]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -6000:]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n))[:60]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), short(a["Kernel_Name"]), short(b["Kernel_Name"]),
         a["Queue_Id"] + ">" + b["Queue_Id"]) for a, b in zip(tail[:-1], tail[1:])]
dur = sum(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) for a in tail)
g = [x for x in gaps if x[0] < 100000]
print(f"{len(g)} boundaries, kernel time {dur / 1e6:.2f} ms, gaps {sum(x[0] for x in g) / 1e6:.2f} ms "
      f"({sum(1 for x in g if x[0] > 3000)} over 3 us: {sum(x[0] for x in g if x[0] > 3000) / 1e6:.2f} ms)")
pairs = collections.Counter(); tot = collections.Counter()
for d, a, b, q in g:
    if d > 3000:
        pairs[(a, b, q)] += 1; tot[(a, b, q)] += d
for (a, b, q), n in pairs.most_common(40):
    print(f"{n:5d} x {tot[(a, b, q)] / n / 1e3:6.1f} us  [{q}]  {a}  ->  {b}")
pos = [i for i, x in enumerate(gaps) if 3000 < x[0] < 100000]
print("distance between consecutive long gaps (dispatches):", sorted(collections.Counter(b - a for a, b in zip(pos[:-1], pos[1:])).items()))
# the last run of consecutive long gaps: where in the step does it begin and end?
runs = []; start = None
for i in pos:
    if start is None: start = prev = i
    elif i - prev <= 3: prev = i
    else: runs.append((start, prev)); start = prev = i
if start is not None: runs.append((start, prev))
for a, b in runs[-3:]:
    print(f"run of long gaps: dispatches {a}..{b} ({b - a + 1} boundaries)")
    for i in list(range(max(a - 3, 0), a + 3)) + [-1] + list(range(b - 2, min(b + 4, len(gaps)))):
        if i < 0: print("      ..."); continue
        print(f"   {i:6d} gap after {gaps[i][0] / 1e3:6.1f} us  {gaps[i][1]}   grid {tail[i]['Grid_Size_X']} lds {tail[i]['LDS_Block_Size']} scratch {tail[i]['Scratch_Size']}")
