#!/usr/bin/env python3
"""Merge the decode + prefill pmc_traffic.json files (tools/prof_bench.sh) into profiles/pmc_traffic.json, which bench.py reads
for roofline.traffic.  Dispatches are labelled with (K, N, M): 16-column strips give N = 16 * blocks for the decode kernel; the
two N = 4096 shapes are told apart by the kernel's rows-per-lane template argument U (plan_gemv: U = 1 iff K/8 >= 1024)."""
import json
import re
import sys

dec, pre, out = sys.argv[1], sys.argv[2], sys.argv[3]
ks = []
for e in json.load(open(dec))["kernels"]:
    m = re.search(r"gemv_q4_f16_mfma_kernel<(\d+), (\d+), (\d+)", e["kernel"])
    if m and e["grid_blocks"][1] == 1:
        ln, mt, u = map(int, m.groups())
        N = e["grid_blocks"][0] * 4 * ln
        K = 4096 if N == 11008 else (11008 if u == 1 else 4096)
        e.update(K=K, N=N, M=1)
        e["kernel"] = "gptq::gemv_q4_f16_mfma_kernel"
    ks.append(e)
for e in json.load(open(pre))["kernels"]:
    if "gemm_kernel" in e["kernel"]:
        e["kernel"] = "gptq::gemm_kernel<4, f16, 4, 64>"
        if e["grid_blocks"][0] == 688:
            e.update(K=4096, N=11008, M=2048)
        elif e["grid_blocks"][0] == 256:
            e.update(K=None, N=4096, M=2048)
    ks.append(e)
json.dump({"note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/prof_bench.sh); FETCH_SIZE KiB x 1024 x 2 "
                   "(gfx950 correction) + WRITE_SIZE KiB x 1024 = hbm_bytes_per_launch", "kernels": ks}, open(out, "w"), indent=1)
for e in ks:
    print(e["kernel"], e["grid_blocks"], e.get("K"), e.get("N"), e.get("M"), e["hbm_bytes_per_launch"])
