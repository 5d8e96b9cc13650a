#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`*_results.db`, the default output of
`rocprofv3 --kernel-trace --stats`) into the compact per-kernel table committed under profiles/.

Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--match gptq] > profiles/rNN_xxx.txt
Columns: calls, total ms, avg us, min us, max us, grid, workgroup, LDS B, VGPRs, AGPRs, SGPRs, name.
If the database holds PMC samples (a --pmc run), the per-kernel mean of every counter is appended.
"""
import argparse
import re
import sqlite3
import sys

def demangle(name: str) -> str:
    """rocprofv3 leaves names it cannot demangle itself mangled (the Itanium demanglers at hand do not know the _Float16 /
    __bf16 template arguments DF16_ / DF16b).  This handles exactly the shape of this library's kernels:
    _ZN4gptq<len><name>I<template args>EEv... with integer, bool and scalar-type arguments."""
    m = re.match(r"^_ZN4gptq(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    m2 = re.match(r"^(\d+)", rest)                     # nested namespace (gptq::midk::gemm_mid_kernel, gptq::mlpk::...)
    if m2:
        n2 = int(m2.group(1))
        base = base + "::" + rest[m2.end():m2.end() + n2]
        rest = rest[m2.end() + n2:]
    if not rest.startswith("I"):
        return "gptq::" + base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        for pat, fn in ((r"Li(\d+)E", lambda g: g.group(1)), (r"Lin(\d+)E", lambda g: "-" + g.group(1)),
                        (r"Lb([01])E", lambda g: "true" if g.group(1) == "1" else "false"),
                        (r"DF16_", lambda g: "f16"), (r"DF16b", lambda g: "bf16"), (r"f", lambda g: "float"),
                        (r"t", lambda g: "unsigned short")):
            g = re.match(pat, rest[i:])
            if g:
                args.append(fn(g))
                i += g.end()
                break
        else:
            return name
    return f"gptq::{base}<{', '.join(args)}>"


def short(name: str, width: int = 150) -> str:
    name = demangle(name)
    name = re.sub(r"\(gptq::Ge[a-z]+Params\)$", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= width else name[: width - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="", help="only kernels whose name contains this substring")
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    cur = c.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "grid_x, grid_y, grid_z, workgroup_x, max(lds_size), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count) from kernels group by name, grid_x, grid_y, grid_z, workgroup_x, lds_size "
        "order by sum(duration) desc").fetchall()      # one row per (kernel, launch geometry, LDS bytes): layer shapes stay apart
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {args.db}")
    print(f"# {'calls':>6} {'total_ms':>9} {'%':>5} {'avg_us':>8} {'min_us':>8} {'max_us':>8}  grid(x,y,z)/wg  lds  vgpr agpr sgpr  kernel")
    n = 0
    for r in rows:
        if args.match and args.match not in r[0]:
            continue
        n += 1
        if n > args.top:
            break
        print(f"  {r[1]:6d} {r[2] / 1e6:9.3f} {100 * r[2] / total:5.1f} {r[3] / 1e3:8.2f} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f}  "
              f"({r[6]},{r[7]},{r[8]})/{r[9]} = {r[6] // max(1, r[9]) * r[7] * r[8]}  {r[10]}  {r[11]} {r[12]} {r[13]}  {short(r[0])}")
    try:
        pmc = cur.execute(
            "select k.name, p.counter_name, avg(p.counter_value), count(*), k.grid_x / k.workgroup_x from pmc_events p join kernels k "
            "on p.dispatch_id = k.dispatch_id group by k.name, k.grid_x, k.workgroup_x, p.counter_name order by k.name, k.grid_x").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("# PMC (mean per dispatch)")
        for name, ctr, val, cnt, blocks in pmc:
            if args.match and args.match not in name:
                continue
            print(f"  {ctr:28s} {val:16.1f}  n={cnt:5d}  blocks={blocks:5d}  {short(name, 80)}")


if __name__ == "__main__":
    sys.exit(main())
