"""Kernels of csrc/*.hip that use scratch memory (register spills or locals the compiler could not keep in registers):
a scratch access inside a K loop waits on vmcnt, i.e. on the DMA it should overlap (round 2: a by-reference conv tap
state in scratch cost the conv class 40 %).  Usage: python tools/check_scratch.py [file.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, FLAGS, HIPCC
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
bad = 0
for f in files:
    r = subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) > 0:
            bad += 1
            print(f"{f}: {int(m.group(1)):5d} B/lane  {name[:140]}")
print(f"{bad} kernel(s) with scratch")
