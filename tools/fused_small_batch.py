#!/usr/bin/env python3
"""[gate|up] layer with the SiLU*mul epilogue (4096 -> 22016) at M = 1..16: the default plan (fused in the GEMV up to 8 rows)
against the unfused form (tuning.path = 3: GEMM-side kernel + elementwise pass).  Usage: python tools/fused_small_batch.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_fused_block
from tools.gemv_sweep import run
from autogptq_amd import _lib

dev = torch.device("cuda:0")
ls = [make_fused_block(dev, 10 * i)[2][3] for i in range(8)]
for M in (1, 2, 4, 5, 8, 9, 16):
    x = (torch.rand(M, 4096, device=dev) - 0.5).half()
    t3 = _lib.GptqTuning(); t3.path = 3
    d = _lib.describe_plan(ls[0]._layer, M)
    print(f"M={M:2d} default[{d.get('kernel')} epi={d.get('epilogue')}]={run(ls, x, None) * 1e6:.2f} us  unfused[{_lib.describe_plan(ls[0]._layer, M, t3).get('kernel')}]={run(ls, x, t3) * 1e6:.2f} us", flush=True)
