import json, sys, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device("cuda:0")
from autogptq_amd import _lib
_lib.load()
print(json.dumps(bench.bench_mlp_prefill(dev, 12)))
print(json.dumps(bench.bench_mlp_prefill(dev, 12, M=512)))
