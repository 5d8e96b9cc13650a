import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout; ignored when the plugin is absent)")


def pytest_collection_modifyitems(config, items):
    """GPU tests get a per-test limit (pytest-timeout, thread method: the process is ended even if the main thread is stuck
    inside a HIP call) so that a wedged kernel fails one test loudly instead of hanging the whole run on the GPU box."""
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


def _torch_dtype(name):
    return {"float16": torch.float16, "float32": torch.float32, "bfloat16": torch.bfloat16}[str(name)]


class RefCase:
    """One reference-generated fixture (tests/golden/ref_*.npz), tensors restored to their dtypes."""

    def __init__(self, path):
        z = np.load(path)
        self.name = os.path.basename(path)[4:-4]
        self.bits = int(z["bits"]); self.group_size = int(z["group_size"])
        self.K = int(z["K"]); self.N = int(z["N"]); self.M = int(z["M"])
        self.act_order = bool(int(z["act_order"]))
        self.dtype = _torch_dtype(z["dtype"]); self.qparams_dtype = _torch_dtype(z["qparams_dtype"])
        self.ref_class = str(z["ref_class"]); self.zero_policy = str(z["zero_policy"])
        self.W = torch.from_numpy(z["W"]); self.scale = torch.from_numpy(z["scale"]); self.zero = torch.from_numpy(z["zero"])
        self.g_idx = torch.from_numpy(z["g_idx"].astype(np.int32))
        self.qweight = torch.from_numpy(z["qweight"]); self.qzeros = torch.from_numpy(z["qzeros"])
        self.scales = torch.from_numpy(z["scales"]).to(self.dtype)
        self.bias = torch.from_numpy(z["bias"]).to(self.dtype) if z["bias"].size else None
        self.lin_bias = torch.from_numpy(z["lin_bias"]).to(self.dtype) if z["lin_bias"].size else None
        self.x = torch.from_numpy(z["x"]).to(self.dtype)
        self.y = torch.from_numpy(z["y"]).to(self.dtype)
        self.Wdq = torch.from_numpy(z["Wdq"]).to(self.dtype)

    @property
    def desc_act_class(self):
        return self.ref_class == "cuda"

    def __repr__(self):
        return self.name


def ref_case_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "ref_*.npz")))


def pytest_generate_tests(metafunc):
    if "ref_case" in metafunc.fixturenames:
        paths = ref_case_paths()
        metafunc.parametrize("ref_case", [RefCase(p) for p in paths], ids=[os.path.basename(p)[4:-4] for p in paths])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Run count of the every-output full-size parity checks (tests/test_gpu_baseline_configs.py), for the round's log."""
    mod = sys.modules.get("test_gpu_baseline_configs") or sys.modules.get("tests.test_gpu_baseline_configs")
    if mod is not None and getattr(mod, "CHECKED", {}).get("cases"):
        terminalreporter.write_line("full-size parity: %d (layer, M) cases, %d outputs compared one by one with the fp64 oracle product"
                                    % (mod.CHECKED["cases"], mod.CHECKED["outputs"]))
