#!/usr/bin/env python3
"""Static instruction count of a kernel: tools/isa_count.py file.hip kernel_substring [extra hipcc flags]
Compiles the device code to assembly (gfx950) and counts instructions of every function whose mangled name contains the substring:
total, fp64 VALU (v_*_f64), MFMA, LDS (ds_*), global / scratch memory, DPP; plus the register figures of the metadata.
(Loops count once: multiply by the trip count by hand.)"""
import re, subprocess, sys, os, tempfile
src, pat = sys.argv[1], sys.argv[2]
out = os.path.join(tempfile.gettempdir(), "isa_%s.s" % os.path.basename(src))
if not os.environ.get("ISA_REUSE") or not os.path.exists(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out] + sys.argv[3:])
cur = None; counts = {}
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur: counts.setdefault(cur, dict(total=0, f64=0, mfma=0, lds=0, vmem=0, scratch=0, dpp=0, salu=0))
        continue
    if cur is None: continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"): cur = None; continue
    t = line.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    c = counts[cur]; c["total"] += 1
    if op.startswith("v_") and "_f64" in op: c["f64"] += 1
    if op.startswith("v_mfma"): c["mfma"] += 1
    if op.startswith("ds_"): c["lds"] += 1
    if op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
    if "dpp" in t: c["dpp"] += 1
    if op.startswith("s_"): c["salu"] += 1
txt = open(out).read()
for k, c in counts.items():
    meta = re.search(r"\.name:\s+%s\b.*?\.vgpr_count:\s+(\d+)" % re.escape(k), txt, re.S)
    sg = re.search(r"\.name:\s+%s\b.*?\.sgpr_count:\s+(\d+)" % re.escape(k), txt, re.S)
    print(k, c, "vgpr", meta.group(1) if meta else "?", "sgpr", sg.group(1) if sg else "?")
