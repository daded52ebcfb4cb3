"""Summarise a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) into a short tracked text file."""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary ({f.split('/')[-1]})\n# {note}\n")
    o.write(f"# total kernel time {tot / 1e6:.2f} ms over all dispatches of the command\n")
    try:          # the kernel sources this profile was taken on (bench.py attaches `dominant_kernel` only to a matching library)
        import bench
        o.write(f"# library_source_stamp {bench.source_stamp()}\n")
    except Exception as e:
        o.write(f"# library_source_stamp unknown ({e})\n")
    o.write(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
    for r in rows[:45]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:90]
        o.write(f"{name:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.2f} "
                f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):7.2f}\n")
print(open(dst).read()[:3000])
