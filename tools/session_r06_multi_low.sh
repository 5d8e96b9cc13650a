#!/bin/bash
# round 6: two strips per workgroup behind one staged x at 1 - 2 rows of 4-bit layers (launches of 1024+ strips): decode tests, then tools/multi_strip_ab.py
# (default plan = 2 strips on gate|up against the forced 1-strip form), then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -5
timeout 600 python tools/multi_strip_ab.py --ms 1,2 2>&1 | grep -v amdgpu.ids | cut -c1-230 > gpurun_out/r06_multi_low_product.log; cat gpurun_out/r06_multi_low_product.log
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch_by_shape'), d['roofline'].get('fused_ms_per_step'))
PY
