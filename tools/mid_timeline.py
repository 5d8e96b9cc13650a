#!/usr/bin/env python3
"""Per-wave s_memtime timeline of gemm_mid_kernel (lab build: tools/ab_build.sh on a copy of csrc/gemm_mid.hip that starts with
`#define GPTQ_MID_TL 1`; GPTQ_MI355X_LIB selects it).  Stamps per wave: entry, table DMAs issued, prologue stages issued, then per K-step
"data landed" (after s_waitcnt vmcnt) and "consumed + next stage issued", then "loop done".  1 tick = 10 ns.
Usage: GPTQ_MI355X_LIB=tools/libgptq_midtl.so python tools/mid_timeline.py [--shape 4096x11008] [--m 64] [--cw 1]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import reserve_workspace


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--cw", type=int, default=1)
    ap.add_argument("--stages", type=int, default=0)
    ap.add_argument("--ksplit", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K, N = map(int, a.shape.split("x"))
    ls = [make_layer(K, N, dev, seed=i) for i in range(12)]          # > 256 MiB would be better; 12 x 22 MB rotates past the L2s at least
    x = (torch.rand(a.m, K, device=dev) - 0.5).half()
    t = _lib.GptqTuning()
    t.path, t.ksplit = 3, a.ksplit
    t.reserved[0], t.reserved[2], t.reserved[3] = a.stages, 5, a.cw
    plan = _lib.describe_plan(ls[0]._layer, a.m, t)
    print("plan:", plan, "   (stamps below: units of 100 shader cycles, ~0.05 us)")
    with torch.no_grad():
        for q in ls:
            q(x, tuning=t)
    torch.cuda.synchronize()
    buf = reserve_workspace(dev, 1)
    nwg = 1024
    tl = torch.zeros(nwg * 8 * 128, dtype=torch.int64, device=dev)
    buf[65536 - 64 + 16:65536 - 64 + 24] = torch.tensor([tl.data_ptr()], dtype=torch.int64).view(torch.uint8).to(dev)
    torch.cuda.synchronize()
    with torch.no_grad():
        y = ls[5](x, tuning=t)
    torch.cuda.synchronize()
    buf[65536 - 64 + 16:65536 - 64 + 24] = 0
    h = tl.cpu().numpy().reshape(nwg, 8, 128)
    used = h[:, :, 0] != 0
    nw = int(used.sum())
    cnt = (h[:, :, 0] & 0xffffffff).astype(np.int64)
    ksteps = (h[:, :, 0] >> 32).astype(np.int64)
    st = h[:, :, 2:].astype(np.int64)
    g0 = st[used][:, 0].min()
    print(f"{nw} waves with stamps in {int(used.any(axis=1).sum())} workgroups; K-steps per wave: min {ksteps[used].min()} max {ksteps[used].max()}")
    ent = st[used][:, 0] - g0
    print(f"wave entry rel. to the first wave of the chip: median {np.median(ent) / 100:.2f} us, max {ent.max() / 100:.2f} us")
    rows = []
    waits, works = [], []
    for wg, wv in zip(*np.nonzero(used)):
        s, n, k = st[wg, wv], int(cnt[wg, wv]), int(ksteps[wg, wv])
        if k == 0:
            continue
        # stamps: 0 entry, 1 tables issued, 2 prologue issued, then (landed, consumed) x k, last = loop done
        e = s[0]
        landed = s[3:3 + 2 * k:2]
        done = s[4:4 + 2 * k:2]
        prev = np.concatenate(([s[2]], done[:-1]))
        waits.append(landed - prev)
        works.append(done - landed)
        rows.append((s[1] - e, s[2] - e, landed[0] - e, done[-1] - e, s[2 + 2 * k] - e, (s[2 + 2 * k] - g0)))
    r = np.array(rows, dtype=np.float64) / 100
    names = ["tables issued", "prologue issued", "first stage landed", "last K-step consumed", "loop done", "loop done rel. to chip start"]
    for i, nme in enumerate(names):
        print(f"  {nme:32s} min {r[:, i].min():7.2f}  median {np.median(r[:, i]):7.2f}  max {r[:, i].max():7.2f} us")
    kmax = max(len(w) for w in waits)
    W = np.full((len(waits), kmax), np.nan); X = np.full((len(works), kmax), np.nan)
    for i, (w, c) in enumerate(zip(waits, works)):
        W[i, :len(w)] = w; X[i, :len(c)] = c
    print("  per K-step (median over waves, us):  wait for the stage | consume + issue the next")
    for k in range(kmax):
        print(f"    step {k:2d}: wait {np.nanmedian(W[:, k]) / 100:6.2f}   work {np.nanmedian(X[:, k]) / 100:6.2f}   (waves {int(np.sum(~np.isnan(W[:, k])))})")
    # epilogue stamps (entries 100..): E0 first barrier, [owner: flags seen], all chunks done, end -- relative to the wave's loop-done stamp
    ep = h[:, :, 100:106].astype(np.int64)
    ksid = np.arange(nwg) % max(1, int(plan.get("ksplit", 1)))
    for who, sel in (("owner slices (ks = 0)", ksid == 0), ("other slices", ksid != 0)):
        rows2 = []
        for wg, wv in zip(*np.nonzero(used)):
            if not sel[wg] or ksteps[wg, wv] == 0:
                continue
            k = int(ksteps[wg, wv]); ld = st[wg, wv][2 + 2 * k]
            e = ep[wg, wv]; e = e[e != 0]
            rows2.append([(v - ld) / 2000.0 for v in e] + [np.nan] * (4 - len(e)))
        if rows2:
            r2 = np.array(rows2)
            print(f"  epilogue stamps after loop done, {who} (us at 2 GHz; median / max): " + "  ".join(f"{np.nanmedian(r2[:, i]):.2f}/{np.nanmax(r2[:, i]):.2f}" for i in range(r2.shape[1]) if not np.all(np.isnan(r2[:, i]))))
    print(f"  sum over K-steps (median wave): wait {np.nanmedian(np.nansum(W, axis=1)) / 100:.2f} us, work {np.nanmedian(np.nansum(X, axis=1)) / 100:.2f} us")


if __name__ == "__main__":
    main()
