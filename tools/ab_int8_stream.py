import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
dev = torch.device("cuda:0")
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
GS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for K, N in ((4096, 11008), (11008, 4096)):
    ls = [make_layer(K, N, dev, bits=BITS, gs=GS, seed=i) for i in range(10)]
    for M in (1, 2, 3, 4):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        t = _lib.GptqTuning(); t.path = 5
        a = run(ls, x, None); b = run(ls, x, t)
        print(f"int{BITS} g{GS} {K}x{N} M={M}: default [{_lib.describe_plan(ls[0]._layer, M).get('kernel')}] {a*1e6:.2f} us | register kernel (path 5) {b*1e6:.2f} us", flush=True)
    del ls
