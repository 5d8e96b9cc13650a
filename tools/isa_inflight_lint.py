#!/usr/bin/env python3
"""Lint for kernels whose loads are inline asm with hand-counted s_waitcnt (gemm_panel_kernel.cuh, gemm_rows_kernel.cuh): in a `hipcc -S` listing (or an
llvm-objdump disassembly) no instruction may READ a VGPR that a global_load wrote until an `s_waitcnt vmcnt(...)` has been passed -- the compiler does not know the
asm is a load and is free to copy its destination (a v_mov at a control-flow join, a live-range split): the copy reads a register the load has not landed in.
Straight-line scan per kernel in listing order (conditional branches ignored, the state is dropped behind an unconditional branch: optimistic there,
conservative elsewhere -- good enough to catch the copies that bit round 6).
Usage: python tools/isa_inflight_lint.py file.s [--match gemm_panel_kernel]    exit status 1 if anything is flagged."""
import re
import sys


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def lint(text, match):
    bad = []
    kernel, pending = None, {}
    for ln, line in enumerate(text.splitlines(), 1):
        t = line.strip()
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", t)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if kernel is None or match not in kernel or not t or t.startswith((";", ".", "//")):
            continue
        t = t.split(";")[0].split("//")[0].strip()
        parts = t.replace(",", " ").split()
        if not parts:
            continue
        op, ops = parts[0], parts[1:]
        if op == "s_waitcnt" and "vmcnt" in t:
            pending = {}                                   # conservative the other way: any vmcnt wait clears (the hand counts are checked by the parity tests)
            continue
        if op == "s_endpgm":
            kernel = None
            continue
        if op in ("s_branch", "s_setpc_b64"):
            pending = {}                                   # what follows in the listing is not reached by falling through: its predecessors are elsewhere (optimistic)
            continue
        if op.startswith("global_load") and "lds" not in op and ops:
            reads = set().union(*[regs_of(o) for o in ops[1:]]) if len(ops) > 1 else set()
            hit = reads & set(pending)
            if hit:
                bad.append((kernel, ln, t, sorted(hit)))
            for r in regs_of(ops[0]):
                pending[r] = ln
            continue
        if op.startswith(("v_", "ds_", "global_", "buffer_", "flat_")):
            # destination = first operand for v_* (not a read, except it would be a WAW on an in-flight register: also wrong); everything else is read
            srcs = ops[1:] if op.startswith("v_") and not op.startswith("v_cmp") else ops
            reads = set().union(*[regs_of(o) for o in srcs]) if srcs else set()
            writes = regs_of(ops[0]) if op.startswith("v_") and ops else set()
            hit = (reads | writes) & set(pending)
            if hit:
                bad.append((kernel, ln, t, sorted(hit)))
    return bad


if __name__ == "__main__":
    path = sys.argv[1]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "gemm_panel_kernel"
    bad = lint(open(path).read(), match)
    for k, ln, t, regs in bad[:40]:
        print(f"{k}: line {ln}: `{t}` touches in-flight v{regs}")
    print(f"{len(bad)} finding(s) in kernels matching {match!r}")
    sys.exit(1 if bad else 0)
