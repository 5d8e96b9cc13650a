#!/usr/bin/env python3
"""Filler instructions between consecutive MFMAs of a kernel's main loop, from a hipcc -S listing: one wave per SIMD hides ~5 single-issue instructions under
a 32-cycle MFMA, every gap with more stalls the matrix pipe.  Usage: python tools/isa_gaps.py <file.s> <mangled kernel name substring>"""
import re, sys
from collections import Counter

s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s+; @', s, re.M) if sys.argv[2] in m.group(1)]
for nm in names:
    i = s.index(nm + ':'); j = s.index('s_endpgm', i)
    lines = [l.strip() for l in s[i:j].split('\n')]
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    loops = [(labels[m.group(1)], n) for n, l in enumerate(lines) for m in [re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)] if m and m.group(1) in labels and labels[m.group(1)] < n]
    if not loops:
        continue
    a, b = max(loops, key=lambda t: t[1] - t[0])
    body = [l for l in lines[a:b + 1] if l and not l.startswith((';', '.'))]
    gaps, cur = [], []
    for l in body:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            gaps.append(cur); cur = []
        else:
            cur.append(op)
    sizes = [len(g) for g in gaps[1:]]
    if not sizes:
        continue
    over = sum(max(0, n - 5) for n in sizes)
    print(nm)
    print(f"  {len(sizes) + 1} MFMAs, {len(body)} instructions in the loop, {sum(sizes) / len(sizes):.2f} fillers per gap, gaps over 5: {sum(1 for n in sizes if n > 5)}, excess fillers {over}")
    print("  histogram:", dict(sorted(Counter(sizes).items())))
    print("  sequence:", sizes)
