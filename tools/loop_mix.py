#!/usr/bin/env python3
"""Instruction mix of the innermost big loop (the frame-pair loop) of one kernel in a hipcc -S listing:
tools/loop_mix.py file.s <mangled-name-substring> [min-instrs]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
lo_lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400


def cls(k):
    return ("pk" if k.startswith("v_pk") else "lds" if k.startswith("ds_") else
            "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else
            "wait" if k.startswith(("s_waitcnt", "s_nop")) else
            "salu" if k.startswith("s_") else "mov" if k.startswith(("v_mov", "v_accvgpr")) else "valu")


parts = re.split(r"\n(_Z[\w]+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if pat not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = n
    loops = []
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            a = labels[m.group(1)]
            ins = [x.strip().split()[0] for x in lines[a:n + 1] if x.startswith("\t") and x.strip() and x.strip()[0] not in ".;"]
            if len(ins) >= lo_lim:
                loops.append((len(ins), a, n, ins))
    loops.sort()
    if loops:
        cnt, a, n, ins = loops[0]
        g = collections.Counter(cls(k) for k in ins)
        print(name[-50:], "loop lines", a, n, "instrs", cnt, dict(g))
