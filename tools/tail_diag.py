#!/usr/bin/env python3
"""Diagnostic for the balanced tail: after a launch, read the published accumulator slabs back from the workspace and compare every slab with the
fp32 product of its K slice (torch matmul on the dequantised weights): says whether a wrong output comes from the published data or from the
last arrival's reads.  No act-order (x is used as given)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from autogptq_amd import qlinear_mi355x as QM


def tun(v):
    t = _lib.GptqTuning(); t.path = 3; t.reserved[3] = v; return t


dev = torch.device("cuda:0")
for (K, N, M) in ((4096, 11008, 1024), (4096, 4096, 4224)):
    q = make_layer(K, N, dev, act_order=False, seed=3)
    x = (torch.rand(M, K, device=dev) - 0.5).half()
    with torch.no_grad():
        y0 = q(x, tuning=tun(41)).float()
        W = q.dequantize().float()
        plan = _lib.describe_plan(q._layer, M, tun(40))
        nbm, nbn = map(int, plan["tiles"].split("x"))
        tail, s = plan["tail"], plan["tail_slices"]
        whole = nbm * nbn - tail
        for rep in range(2):
            y1 = q(x, tuning=tun(40)).float()
            torch.cuda.synchronize()
            ent = QM._WORKSPACE[(0, int(torch.cuda.current_stream(dev).cuda_stream))]
            slabs = ent[0][_lib.WS_HEADER_BYTES:_lib.WS_HEADER_BYTES + tail * s * 131072].view(torch.float32).view(tail, s, 4, 2, 4, 4, 2, 32, 4).clone()
            # dims: t, slice, mt, nt, q, wave, half, l31, e   ->  rows mt*32 + e + 8*q + 4*half ; cols wave*64 + 2*l31 + nt
            bad_out = ((y1 - y0).abs() > 0.01) | ~torch.isfinite(y1)
            print(f"{K}x{N} M={M} tail={tail}x{s} rep {rep}: bad outputs {int(bad_out.sum())}")
            nshown = 0
            for t in range(tail):
                L = whole + t
                full, per = nbn >> 3, nbm * 8
                if L < full * per:
                    cb, r = divmod(L, per); bm, bn = r >> 3, cb * 8 + (r & 7)
                else:
                    w = nbn & 7; r = L - full * per; bm = r // w; bn = full * 8 + (r - bm * w)
                tile_bad = int(bad_out[bm * 128:(bm + 1) * 128, bn * 256:(bn + 1) * 256].sum())
                rep_s = []
                for sl in range(s):
                    k0, k1 = sl * K // s, (sl + 1) * K // s
                    exp = x[bm * 128:(bm + 1) * 128, k0:k1].float() @ W[k0:k1, bn * 256:(bn + 1) * 256]          # [128, 256]
                    # exp[row, col] -> slab layout
                    e5 = exp.view(4, 4, 2, 4, 4, 32, 2)            # mt, q, half, e, wave, l31, nt   (row = mt*32 + q*8 + half*4 + e ; col = wave*64 + l31*2 + nt)
                    e5 = e5.permute(0, 6, 1, 4, 2, 5, 3)            # mt, nt, q, wave, half, l31, e
                    got = slabs[t, sl]
                    d = (got - e5).abs()
                    bad = (d > 2e-2) | ~torch.isfinite(got)
                    rep_s.append(int(bad.sum()))
                    if 0 < int(bad.sum()) < 20000 and nshown < 4:
                        idx = bad.nonzero()
                        print(f"   tile t={t} (bm {bm}, bn {bn}) slice {sl}: {idx.shape[0]} wrong of 32768; mt {sorted(set(idx[:,0].tolist()))} nt {sorted(set(idx[:,1].tolist()))} q {sorted(set(idx[:,2].tolist()))} "
                              f"wave {sorted(set(idx[:,3].tolist()))} half {sorted(set(idx[:,4].tolist()))} l31 {sorted(set(idx[:,5].tolist()))} e {sorted(set(idx[:,6].tolist()))}")
                        a = idx[0].tolist()
                        print("      e.g.", a, float(got[tuple(a)]), "expected", float(e5[tuple(a)]))
                        nshown += 1
                if tile_bad or any(0 < v < 20000 for v in rep_s):
                    print(f"  t={t} (bm {bm}, bn {bn}): bad outputs {tile_bad}; wrong elements per slab {rep_s}  (one slab per tile is not written: the last arrival's)")
