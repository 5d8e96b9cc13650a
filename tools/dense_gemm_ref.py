#!/usr/bin/env python
"""Context number for DESIGN.md: what the vendor dense GEMM (torch.matmul -> hipBLASLt) reaches on this GPU for the prefill
shapes, with fp16/bf16 weights already dequantised (i.e. the path "dequantise once, then dense GEMM", 4x the weight bytes).
Weights are rotated through > 256 MiB so they do not sit in the Infinity Cache between launches."""
import sys
import torch

def main():
    dev = torch.device("cuda:0")
    shapes = [(2048, 4096, 4096), (2048, 4096, 11008), (2048, 11008, 4096), (4096, 4096, 4096), (512, 4096, 4096)]
    for dtype in ((torch.float16,) if "--fp16" in sys.argv else (torch.float16, torch.bfloat16)):
        for M, K, N in shapes:
            nl = max(2, (300 << 20) // (K * N * 2))
            W = [(torch.rand(K, N, device=dev) - 0.5).to(dtype) for _ in range(nl)]
            x = (torch.rand(M, K, device=dev) - 0.5).to(dtype)
            out = torch.empty(M, N, device=dev, dtype=dtype)
            import time
            t_end = time.perf_counter() + 0.25              # settled clocks, as bench.py does before every timed graph
            while time.perf_counter() < t_end:
                for w in W:
                    torch.matmul(x, w, out=out)
                torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in W:
                    torch.matmul(x, w, out=out)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / nl)
            print(f"{str(dtype):16s} M={M} K={K} N={N}: {best * 1e3:8.1f} us  {2 * M * K * N / best / 1e9:7.1f} TFLOP/s", flush=True)

def decode():
    """--decode: M = 1 on the Llama-7B shapes, fp16 weights (4x the int4 bytes), rotating HBM-cold layers inside ONE hipGraph like bench.py's decode stack."""
    import time
    dev = torch.device("cuda:0")
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
        nl = max(8, (600 << 20) // (K * N * 2))
        W = [(torch.rand(K, N, device=dev) - 0.5).half() for _ in range(nl)]
        x = (torch.rand(1, K, device=dev) - 0.5).half()
        outs = [torch.empty(1, N, device=dev, dtype=torch.float16) for _ in range(nl)]
        for w, o in zip(W, outs):
            torch.matmul(x, w, out=o)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w, o in zip(W, outs):
                torch.matmul(x, w, out=o)
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            g.replay()
            torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (4 * nl))
        b = K * N * 2 + 2 * K + 2 * N
        print(f"dense fp16 GEMV M=1 K={K} N={N}: {best * 1e3:7.2f} us per launch  {b / best / 1e6:7.0f} GB/s of {b} B", flush=True)


if __name__ == "__main__":
    if "--decode" in sys.argv:
        sys.exit(decode())
    sys.exit(main())
