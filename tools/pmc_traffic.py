#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the gptq kernels from rocprofv3 PMC databases.

  python tools/pmc_traffic.py --fetch <FETCH_SIZE db> [--write <WRITE_SIZE db>] --out profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, so the read side is doubled (MI355X_MICROARCH.md, section HBM).  Dispatches are keyed by kernel
name + grid so the three Llama-7B layer shapes of bench.py are told apart:
  grid.x (threads) / workgroup = column strips  ->  N = strips * 16 for the 16-column-strip decode kernels.
"""
import argparse
import re
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import demangle


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select k.name, k.grid_x, k.grid_y, k.workgroup_x, avg(p.counter_value), count(*), k.lds_size from pmc_events p join kernels k "
        "on p.dispatch_id = k.dispatch_id where p.counter_name = ? and k.name like '%gptq%' group by k.name, k.grid_x, k.grid_y, k.workgroup_x, k.lds_size",
        (counter,)).fetchall()
    return {(r[0], r[1], r[2], r[3], r[6]): (r[4], r[5]) for r in rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write")
    ap.add_argument("--out", required=True)
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096", help="KxN list used to label decode dispatches")
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--prefill-m", type=int, default=0, help="rows per step of the prefill bench (labels gemm_kernel dispatches)")
    ap.add_argument("--label-gemm", default="", help="K,N,M: EVERY gptq gemm* dispatch of these databases gets this label (a single-shape run: tools/prefill_one.py)")
    ap.add_argument("--append", default="", help="an existing pmc_traffic.json: its rows are kept, rows of this run with the same (kernel, K, N, M) replace them")
    args = ap.parse_args()
    fetch = per_kernel(args.fetch, "FETCH_SIZE")
    write = per_kernel(args.write, "WRITE_SIZE") if args.write else {}
    shapes = [tuple(map(int, s.split("x"))) for s in args.shapes.split(",")]
    out = []
    for key, (kib, n) in sorted(fetch.items()):
        name, gx, gy, wg, lds = key
        name = demangle(name)
        blocks = gx // wg
        ent = {"kernel": name.split("(")[0].replace("void ", ""), "grid_blocks": [blocks, gy], "workgroup": wg, "dispatches": n,
               "FETCH_SIZE_KiB": round(kib, 1), "fetch_bytes_x2": int(kib * 1024 * 2)}
        w = write.get(key)
        ent["WRITE_SIZE_KiB"] = round(w[0], 1) if w else None
        ent["hbm_bytes_per_launch"] = ent["fetch_bytes_x2"] + (int(w[0] * 1024) if w else 0)
        # label with (K, N, M): the launch geometries bench.py produces (decode: M = 1; prefill: --prefill-m rows)
        ent["K"] = ent["N"] = ent["M"] = None
        ent["lds_bytes"] = lds
        if "gemv_tiled_kernel" in name:
            # decode-copy kernel (round 4): one 16-column strip per workgroup -> N = strips * 16 (a multi-layer launch: the summed width); the staged x
            # (K * 2 bytes of LDS) tells the 4096-deep layers from the 11008-deep one
            # (two / four strips per workgroup -- x mode 6 / 5 -- and the [gate | up] pair form -- 4: a workgroup covers 32 / 64 columns)
            xm = re.search(r"gemv_tiled_kernel<\s*\d+,\s*\d+,\s*\d+,\s*\w+,\s*\d+,\s*(\d+)", name)
            tiled_mult = {"6": 2, "5": 4, "4": 2}.get(xm.group(1), 1) if xm else 1
            ent["N"], ent["M"] = blocks * 16 * tiled_mult, args.m
            ent["K"] = 11008 if lds and lds > 20000 else 4096
        if "gemv_q4_stream_kernel" in name:
            # streamed GEMV: (workgroups, threads) -> launch; multi-layer launches are labelled with the summed width
            geo = {(192, 1024): (4096, 12288), (688, 512): (4096, 22016), (688, 256): (4096, 22016), (172, 1024): (4096, 11008)}
            if (blocks, wg) in geo:
                ent["K"], ent["N"] = geo[(blocks, wg)]
                ent["M"] = args.m
        for K, N in shapes:
            if blocks * 16 == N and "gemv" in name and "stream" not in name and not ("gemv_tiled_kernel" in name and ent["N"] != N):
                ent["N"], ent["M"] = N, args.m
                cands = [k for k, n2 in shapes if n2 == N]
                ent["K_candidates"] = cands
                if len(cands) == 1:
                    ent["K"] = cands[0]
                elif "mfma_kernel<4, 1, 1," in name:      # rows per lane U = 1 <=> K/8 >= 1024 (plan_gemv): the 11008-row layer
                    ent["K"] = max(cands)
                elif "mfma_kernel<4, 1, 2," in name:
                    ent["K"] = min(cands)
            if "gemm_wide_kernel" in name and args.prefill_m:   # 128 x 512 workgroup tiles (bench: M = 4096 on 4096 x 4096)
                if blocks == -(-4096 // 128) * -(-N // 512) and N == 4096:
                    ent["N"], ent["M"], ent["K"] = N, 4096, 4096
            if "gemm_kernel" in name and args.prefill_m:        # 128 x 256 workgroup tiles
                if blocks == -(-args.prefill_m // 128) * -(-N // 256):
                    ent["N"], ent["M"] = N, args.prefill_m
                    cands = [k for k, n2 in shapes if n2 == N]
                    ent["K_candidates"] = cands
                    if len(cands) == 1:
                        ent["K"] = cands[0]
        if args.label_gemm and "gemm" in name:
            ent["K"], ent["N"], ent["M"] = map(int, args.label_gemm.split(","))
            ent.pop("K_candidates", None)
        out.append(ent)
    if args.label_gemm:                                        # a single-shape run: only its GEMM rows are of interest
        out = [e for e in out if "gemm" in e["kernel"]]
    if args.append and os.path.exists(args.append):
        old = json.load(open(args.append)).get("kernels", [])
        key = lambda e: (e["kernel"].replace(" ", ""), e.get("K"), e.get("N"), e.get("M"), tuple(e.get("grid_blocks") or ()))
        new = {key(e) for e in out}
        out = [e for e in old if key(e) not in new] + out
    with open(args.out, "w") as f:
        json.dump({"source": {"fetch": args.fetch, "write": args.write}, "note": "FETCH_SIZE doubled (gfx950 correction)",
                   "kernels": out}, f, indent=1)
    for e in out:
        print(e)


if __name__ == "__main__":
    main()
