#!/usr/bin/env python3
"""Refactor safety for the hot kernels, no GPU: compile one translation unit to assembly at two revisions and compare every kernel's instruction stream, function by
function -- a change that adds template instantiations (a new row count, a lab variant) must leave the existing ones byte for byte what they were.
Basic-block labels (.LBB<function>_<n>) are renumbered when functions are added; comments and the kernel-descriptor / metadata sections are ignored.

Usage:  python tools/isa_same.py <before.s> <after.s> [--rename OLD=NEW ...]
  produce the inputs with
    cd autogptq_amd/csrc && hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -fno-strict-aliasing --cuda-device-only -S gemv_tiled.hip -o /tmp/after.s
  (the same command in a checkout of the old revision for before.s); --rename maps mangled-name fragments when a template gained a parameter, e.g.
  --rename 'EEEvNS0_10WideParamsE=ELb0EEEvNS0_10WideParamsE'.
Round 4 uses: the 5..8-row form of gemv_tiled_kernel (all 96 plain kernels identical; 64 of the 72 act-order ones, the other 8 -- bf16, one row -- differ by one
moved instruction or a renamed scalar register), the lab tail of gemm_wide_kernel (7 identical)."""
import difflib
import re
import sys


def functions(path):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[cur] = buf
            cur = None
            continue
        l = re.sub(r"\s*;.*$", "", line.strip())
        if not l or l.startswith((".amdhsa_", ".section", ".p2align", ".end_amdhsa", ".text", ".protected", ".globl", ".type", ".weak")):
            continue
        buf.append(re.sub(r"\.LBB\d+_", ".LBB_", l))
    return out


def main():
    args, ren, it = [], [], iter(sys.argv[1:])
    for a in it:
        if a == "--rename":
            ren.append(next(it).split("=", 1))
        else:
            args.append(a)
    if len(args) != 2:
        sys.exit(__doc__)
    a, b = functions(args[0]), functions(args[1])
    same = diff = gone = 0
    for name, body in a.items():
        new = name
        for o, n in ren:
            new = new.replace(o, n)
        if new not in b:
            gone += 1
            print("missing :", name)
        elif [x.replace(name, "@") for x in body] == [x.replace(new, "@") for x in b[new]]:
            same += 1
        else:
            diff += 1
            d = [x for x in difflib.unified_diff(body, b[new], lineterm="", n=0) if not x.startswith(("---", "+++", "@@"))]
            print(f"differs : {name}  ({len(d)} changed lines; first: {d[:4]})")
    print(f"{same} identical, {diff} different, {gone} missing, {len(b) - same - diff} new")
    return 1 if (diff or gone) else 0


if __name__ == "__main__":
    sys.exit(main())
