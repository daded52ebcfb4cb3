"""Per-kernel resource usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, spills, scratch,
occupancy, LDS.  Usage: python tools/kernel_resources.py attention.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, FLAGS, HIPCC
src = os.path.join(CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line or ("warning" in line and "remark" not in line):
        print(line)
print(f"{'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'occ':>4} {'LDS':>7}  kernel")
for c in rows:
    if flt in c["name"]:
        print(f"{c.get('VGPRs', 0):5d} {c.get('AGPRs', 0):5d} {c.get('VGPRs Spill', 0):6d} {c.get('SGPRs Spill', 0):6d} "
              f"{c.get('ScratchSize [bytes/lane]', 0):7d} {c.get('Occupancy [waves/SIMD]', 0):4d} {c.get('LDS Size [bytes/block]', 0):7d}  {c['name'][:110]}")
