"""Compares two dumps of tools/f0_dump.py frame for frame (NaN = unvoiced): python tools/f0_dump_cmp.py a.npy b.npy"""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
same = (a == b) | (np.isnan(a) & np.isnan(b))
print(f"frames {a.size}, voiced {int((~np.isnan(a)).sum())} / {int((~np.isnan(b)).sum())}, differing {int((~same).sum())}")
sys.exit(0 if same.all() and a.size == b.size else 1)
