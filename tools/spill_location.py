"""Where do a kernel's register spills sit relative to its main loop?  Compiles one csrc/*.hip to assembly and prints, for
every kernel whose mangled name contains the filter, the line numbers of its loop headers, barriers and scratch
(spill) instructions.  Usage: python tools/spill_location.py attention.hip attn_kernelILi2ELi48ELi2ELi64ELb1"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, FLAGS, HIPCC
src, flt = os.path.join(CSRC, sys.argv[1]), sys.argv[2]
out = "/tmp/_spill_loc.s"
subprocess.run([HIPCC] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", src, "-o", out], check=True,
               capture_output=True)
s = open(out).read()
for line in s.splitlines():
    if line.startswith("_Z") and ":" in line and flt in line.split(":")[0]:
        name = line.split(":")[0]
        body = s[s.index(name + ":"): s.index("s_endpgm", s.index(name + ":"))].splitlines()
        print(name, f"({len(body)} lines)")
        for i, l in enumerate(body):
            if "Loop Header" in l:
                print(f"  {i:5d}  loop header   {l.strip()[:60]}")
            elif "s_barrier" in l:
                print(f"  {i:5d}  s_barrier")
            elif "scratch_" in l:
                print(f"  {i:5d}  {l.strip()[:90]}")
        loops = [i for i, l in enumerate(body) if "Loop Header" in l]
        last_in_loop = max([i for i, l in enumerate(body) if "in Loop: Header" in l] + loops + [0])
        spills = [i for i, l in enumerate(body) if "scratch_" in l]
        inside = [i for i in spills if loops and loops[0] <= i <= last_in_loop]
        print(f"  => {len(spills)} scratch instructions, {len(inside)} inside the loop body (lines {loops[0] if loops else '-'}..{last_in_loop})")
