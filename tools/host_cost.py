#!/usr/bin/env python3
"""Host cost of an eager decode call, piece by piece (us per call over a few thousand calls; the GPU work per call is shorter than the host work, so the
queue never fills): torch.empty, the raw C-ABI call through the METH_FASTCALL trampoline (plan + hipLaunchKernel), QuantLinear.__call__, forward_multi."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import forward_multi, _raw_stream

dev = torch.device("cuda:0")
q = make_layer(4096, 4096, dev, seed=1)
k = make_layer(4096, 4096, dev, seed=2)
v = make_layer(4096, 4096, dev, seed=3)
x = (torch.rand(1, 4096, device=dev) - 0.5).half()
out = torch.empty(1, 4096, dtype=torch.float16, device=dev)
N = 4000


def t(fn, n=N):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


with torch.no_grad():
    print(f"torch.empty((1, 4096))            {t(lambda: torch.empty((1, 4096), dtype=torch.float16, device=dev)):6.2f} us")
    fast = _lib.fast
    if fast is not None:
        la, xp, op, st = q._layer_addr, x.data_ptr(), out.data_ptr(), _raw_stream(0)
        print(f"fast.forward (plan + launch)      {t(lambda: fast.forward(la, xp, op, 1, 0, 0, st, 0)):6.2f} us")
        print(f"_raw_stream(0)                    {t(lambda: _raw_stream(0)):6.2f} us")
    print(f"x.data_ptr()                      {t(lambda: x.data_ptr()):6.2f} us")
    print(f"torch.cuda.current_device()       {t(lambda: torch.cuda.current_device()):6.2f} us")
    print(f"QuantLinear.forward(x)            {t(lambda: q.forward(x)):6.2f} us")
    print(f"QuantLinear.__call__(x)           {t(lambda: q(x)):6.2f} us")
    grp = [q, k, v]
    print(f"forward_multi([q, k, v], x)       {t(lambda: forward_multi(grp, x)):6.2f} us")
    print(f"nn.Linear-style baseline: x @ W   {t(lambda: torch.matmul(x, x.t())):6.2f} us   (a torch op of the same launch count, for scale)")
