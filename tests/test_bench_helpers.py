"""Host-side pieces of bench.py that the driver's multi-GPU run depends on (no GPU needed)."""
import importlib.util
import os
import threading
import time

import pytest


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tp_deadline_runs_the_fallback_exactly_once_or_never():
    """The tensor-parallel block of `bench.py --gpus N` runs under _deadline(): either the block finishes first (finish() is True and the
    fallback never runs) or the fallback runs once -- it prints the headline without `tp` and ends the process -- and finish() is False."""
    B = _bench()
    fired = []
    finish = B._deadline(0.3, lambda: fired.append(1))
    assert finish() is True
    time.sleep(0.5)
    assert fired == [] and finish() is False            # a second finish() cannot win again

    fired2, gate = [], threading.Event()
    finish2 = B._deadline(0.05, lambda: (fired2.append(1), gate.set()))
    assert gate.wait(2.0)
    assert fired2 == [1] and finish2() is False
    time.sleep(0.2)
    assert fired2 == [1]


def test_algorithmic_bytes_formula():
    """packed weights + zero-points + scales (+ g_idx for act-order) + x + y per launch."""
    B = _bench()
    K, N, M = 4096, 11008, 1
    assert B.algorithmic_bytes(K, N, M) == K * N // 2 + (K // 128) * (N // 8) * 4 + (K // 128) * N * 2 + M * K * 2 + M * N * 2
    assert B.algorithmic_bytes(K, N, M, act_order=True) - B.algorithmic_bytes(K, N, M) == K * 4
