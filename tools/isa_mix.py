#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing: tools/isa_mix.py file.s [name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z[\w]+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if pat not in name:
        continue
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        grp = ("pk" if k.startswith("v_pk") else "lds" if k.startswith("ds_") else
               "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else
               "salu" if k.startswith("s_") else "mov" if k.startswith(("v_mov", "v_accvgpr")) else "valu")
        g[grp] += v
    print(name[-60:], "total", sum(c.values()), dict(g))
    print("   ", c.most_common(30))
