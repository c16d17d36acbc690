#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_stats.py conv.s conv_rwb_fwd_kernel"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\w*%s\w*:" % name, l))
end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm"))
ins = [l.split()[0] for l in txt[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
grp = collections.Counter()
for k, v in c.items():
    g = ("mfma" if "mfma" in k else "ds_read" if k.startswith("ds_read") else "ds_write" if k.startswith("ds_write")
         else "vmem_load" if k.startswith(("buffer_load", "global_load")) else "vmem_store" if k.startswith(("buffer_store", "global_store"))
         else "scratch" if "scratch" in k else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "other")
    grp[g] += v
print(name, "instructions:", len(ins), dict(grp))
print("  top VALU:", [(k, v) for k, v in c.most_common(60) if k.startswith("v_") and "mfma" not in k][:14])
for l in txt[end:end + 400]:
    if re.search(r"\.set .*%s.*\.(num_vgpr|num_agpr|private_seg_size)" % name, l):
        print("  ", l.split(".")[-1].strip())
