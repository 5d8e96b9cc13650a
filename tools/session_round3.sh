#!/bin/bash
# One gpurun call that reproduces the round-3 evidence (run on the GPU box from the repo root; ~12 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/session_round3.sh'
# Everything lands under gpurun_out/r03s/ ; the summaries judged are the copies under profiles/r03_*.
set -u
O=gpurun_out/r03s; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
bash tools/prof_bench.sh r03s_prof > $O/prof.log 2>&1
timeout 600 python tools/mid_sweep.py --quick --ms 33,64,96,128,192,256 --shapes 4096x4096,4096x11008,11008x4096,5120x5120 > $O/mid_sweep_defaults.log 2>&1
timeout 300 python tools/mid_multi_ab.py --ms 33,64,128 > $O/mid_multi_ab.log 2>&1
timeout 300 python tools/ab_int8_stream.py > $O/int8_stream_ab.log 2>&1
timeout 300 python tools/hot_cold.py > $O/hot_cold.log 2>&1
hipcc --offload-arch=gfx950 -O2 tools/xfetch_lab.hip -o /tmp/xfetch_lab && timeout 120 /tmp/xfetch_lab > $O/xfetch_lab.log 2>&1
hipcc --offload-arch=gfx950 -O2 --cuda-device-only --no-gpu-bundle-output tools/aql_lab_kernels.hip -o /tmp/aql_lab.hsaco && hipcc -O2 tools/aql_lab.cpp -o /tmp/aql_lab -lhsa-runtime64 2>/dev/null \
  && (timeout 120 /tmp/aql_lab /tmp/aql_lab.hsaco > $O/aql_lab_fenced.log 2>&1; timeout 120 /tmp/aql_lab /tmp/aql_lab.hsaco nofence > $O/aql_lab_nofence.log 2>&1)
tail -3 $O/gpu_tests.log; head -c 300 $O/bench.json; echo
