#!/usr/bin/env python3
"""Context numbers for DESIGN.md 5: what this box's HBM delivers to plain torch kernels in the same session and clock protocol as bench.py --
a device-to-device copy (read + write) and a read-only reduction over 2 GiB, HIP events, settled clocks."""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 30                                            # 2 GiB of fp16
a = torch.ones(n, dtype=torch.float16, device=dev)
b = torch.empty_like(a)
ai = a.view(torch.int32)


def timed(fn, reps=5):
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


t = timed(lambda: b.copy_(a))
print(f"copy 2 GiB -> 2 GiB      {t * 1e3:7.3f} ms   {2 * 2 * n / t / 1e12:5.2f} TB/s (read + write)")
t = timed(lambda: torch.sum(ai))
print(f"read-only sum over 2 GiB {t * 1e3:7.3f} ms   {2 * n / t / 1e12:5.2f} TB/s")
t = timed(lambda: torch.bitwise_and(ai, 1, out=b.view(torch.int32)))
print(f"elementwise 2 GiB -> 2 GiB {t * 1e3:7.3f} ms   {2 * 2 * n / t / 1e12:5.2f} TB/s (read + write)")
