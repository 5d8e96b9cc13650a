import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import bench
from autogptq_amd import _lib
dev = torch.device("cuda:0")
def tune_s64():
    t = _lib.GptqTuning(); t.path = 3; t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_STREAM64; return t
def tune_path(p):
    t = _lib.GptqTuning(); t.path = p; return t
def timeit(ls, x, t):
    def call(): return [q(x, tuning=t) if t is not None else q(x) for _,_,_,q in ls]
    try:
        with torch.no_grad(): call()
    except Exception as e: return None
    torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad(): outs = call()
    bench.settle(g, dev); _, evt = bench.time_graph(g, 6, dev); del g, outs
    return evt / (6 * len(ls)) * 1e6
for K, N in ((13824,5120),(17920,6656),(28672,8192),(11008,4096),(14336,4096),(8192,8192)):
    n = max(4, -(-(320 << 20) // (K * N // 2)))
    ls = [("b", K, N, bench.make_layer(K, N, dev, seed=9900 + i)) for i in range(n)]
    for M in (2, 3, 4, 5):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        timeit(ls, x, None)
        row = []
        for nm, t in (("default", None), ("stream64", tune_s64()), ("gemv(path5)", tune_path(5)), ("stream(path6)", tune_path(6)), ("gemm(path3)", tune_path(3))):
            us = timeit(ls, x, t); row.append(f"{nm} {us:6.2f}" if us else f"{nm} refused")
        print(f"{K}x{N} M={M} [{bench._plan_dict(ls,K,N,M).get('kernel')} ks={bench._plan_dict(ls,K,N,M).get('ksplit')}]: " + " | ".join(row), flush=True)
    del ls; torch.cuda.empty_cache()
