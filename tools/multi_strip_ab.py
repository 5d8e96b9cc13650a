#!/usr/bin/env python3
"""Round 5: the decode-copy kernel with 1 / 2 / 4 strips per workgroup behind one staged x (tuning.path = 8, reserved[1] = strips per workgroup) against the
planner's default, 3..8 rows, single layers and the q|k|v / gate|up launches; rotating HBM-cold layers in a hipGraph.
Usage: python tools/multi_strip_ab.py [--ms 3,4,5,8]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.tiled_sweep import timed
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import forward_multi

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="3,4,5,8")
a = ap.parse_args()
dev = torch.device("cuda:0")


def tune(nstr):
    t = _lib.GptqTuning()
    t.path = 8
    t.reserved[1] = nstr
    return t


GROUPS = [("4096x4096", [(4096, 4096)]), ("4096x11008", [(4096, 11008)]), ("11008x4096", [(11008, 4096)]), ("q|k|v", [(4096, 4096)] * 3), ("gate|up", [(4096, 11008)] * 2)]
for name, shapes in GROUPS:
    per = sum(k * n // 2 for k, n in shapes)
    nl = max(3, min(24, (400 << 20) // per))
    sets = [[make_layer(k, n, dev, seed=i * 8 + j) for j, (k, n) in enumerate(shapes)] for i in range(nl)]
    K = shapes[0][0]
    for M in map(int, a.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        res = {}
        for label, t in (("default", None), ("1 strip", tune(1)), ("2 strips", tune(2)), ("4 strips", tune(4))):
            def fn():
                return [forward_multi(s, x, t) if len(s) > 1 else s[0](x, tuning=t) for s in sets]
            try:
                best = min(timed(fn)[0] for _ in range(2)) / len(sets)
                d = _lib.describe_plan(sets[0][0]._layer, M, t)
                res[label] = f"{best * 1e6:6.2f} us [{d.get('kernel')} w={d.get('waves')} u={d.get('u')}]"
            except Exception as e:
                res[label] = "n/a (" + str(e)[:30] + ")"
        print(f"{name:11s} M={M}: " + "   ".join(f"{k}: {v}" for k, v in res.items()), flush=True)
    del sets
