import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench as B
dev = torch.device("cuda:0")
def cap(layers, xs, keep):
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        for _, K, N, q in layers: B.call(q, xs[K])
    torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph(); outs = []
    with torch.cuda.graph(g), torch.no_grad():
        for _, K, N, q in layers:
            o = B.call(q, xs[K])
            if keep: outs.append(o)
            del o
    return g, outs
for M, K, N, act, n in ((4096, 4096, 4096, False, 8), (2048, 4096, 11008, True, 8), (2048, 4096, 4096, True, 8)):
    ls = [("q", K, N, B.make_layer(K, N, dev, act_order=act, seed=900 + i)) for i in range(n)]
    xs = {K: (torch.rand(M, K, device=dev) - 0.5).half()}
    for keep in (True, False, True, False):
        g, outs = cap(ls, xs, keep)
        for _ in range(2): g.replay()
        _, ev = B.time_graph(g, 5, dev)
        per = ev / (5 * n)
        print(f"M={M} {K}x{N} act={act} keep_outputs={keep}: {per*1e6:.1f} us  {2*M*K*N/per/1e12:.0f} TFLOP/s", flush=True)
        del g, outs
    del ls
    torch.cuda.empty_cache()
