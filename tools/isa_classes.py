#!/usr/bin/env python3
"""Instruction classes of one kernel in the assembly tools/isa_count.py left in /tmp: tools/isa_classes.py file.hip mangled_name [--loops]"""
import re, sys, os, tempfile
from collections import Counter
src, name = sys.argv[1], sys.argv[2]
lines = open(os.path.join(tempfile.gettempdir(), "isa_%s.s" % os.path.basename(src))).read().split("\n")
on = False; tot = Counter(); loop = Counter(); inloop = False
for ln in lines:
    if ln.startswith(name + ":"): on = True; continue
    if not on: continue
    if ln.startswith("\t.end_amdhsa_kernel"): break
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
    if m: inloop = "Loop" in m.group(2); continue
    t = ln.strip()
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    cls = "acc" if "accvgpr" in op else "f64" if "_f64" in op else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch") else "vmem" if op.startswith(("global", "buffer")) else "salu" if op.startswith("s_") else "valu_other"
    tot[cls] += 1
    if inloop: loop[cls] += 1
print("kernel", sum(tot.values()), dict(tot)); print("in loops", sum(loop.values()), dict(loop))
