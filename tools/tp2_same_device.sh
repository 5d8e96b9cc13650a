HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --same-device --backend gloo --tp-exchange peer_store --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_tp2_same_device.json 2> gpurun_out/r04_bench_tp2.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_tp2_same_device.json").read().strip().splitlines()[-1])
r=d["roofline"]
print({k:v for k,v in r.items() if k.startswith("tp") and "peer" in k or k.endswith("local_only_graph")})
PY
timeout 300 python -m pytest tests/test_peer_exchange.py -x -q -m gpu -k "forward_scatter or forward_gather or two_processes" 2>&1 | tail -3
