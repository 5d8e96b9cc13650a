import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from autogptq_amd import _lib
dev = torch.device("cuda:0")
def tune(waves, u, ks):
    t = _lib.GptqTuning(); t.path = 8; t.waves = waves; t.ksplit = ks; t.reserved[_lib.LAB.DEPTH] = u; return t
def timeit(ls, x, t):
    def call(): return [q(x, tuning=t) if t is not None else q(x) for _,_,_,q in ls]
    try:
        with torch.no_grad(): call()
    except Exception as e: return None
    torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad(): outs = call()
    bench.settle(g, dev); _, evt = bench.time_graph(g, 6, dev); del g, outs
    return evt / (6 * len(ls)) * 1e6
for K, N in ((8192,1024),(8192,128),(8192,3584),(28672,1024),(8192,1280),(4096,512),(4096,1376)):
    n = max(8, min(64, -(-(320 << 20) // (K * N // 2))))
    ls = [("b", K, N, bench.make_layer(K, N, dev, seed=9990 + i)) for i in range(n)]
    for M in (1, 4):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        timeit(ls, x, None)
        p = bench._plan_dict(ls, K, N, M)
        row = [f"default[{p.get('kernel')} w={p.get('waves')} u={p.get('u')} ks={p.get('ksplit')}] {timeit(ls, x, None):6.2f}"]
        for w,u in ((16,2),(8,4),(8,2),(4,4)):
            for ks in (1,2,4,8):
                us = timeit(ls, x, tune(w,u,ks)); row.append(f"{w}x{u}k{ks} {us:5.2f}" if us else f"{w}x{u}k{ks} -")
        print(f"{K}x{N} M={M}: " + " | ".join(row), flush=True)
    del ls; torch.cuda.empty_cache()
