#!/bin/bash
# First GPU call of the next round (≈ 6 GPU-minutes): the state the round ended in, then the two prefill experiments that were queued unrun.
#   /usr/local/graft/bin/gpurun --timeout 700 -- 'bash tools/session_round4.sh'
# Everything lands under gpurun_out/r04s/ ; copy what is to be judged into profiles/r04_*.
set -u
O=gpurun_out/r04s; mkdir -p $O
timeout 330 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/gpu_tests.log
timeout 100 python bench.py > $O/bench.json 2> $O/bench.err
# (1) K-split launches of the tiled kernel combined INSIDE the launch (tuning.reserved[3] = 43: every tile a tail tile, 4-wave workgroups) against
#     the default (slabs + reduce launch, 8-wave K-group form): the 129..1024-row band on layers with fewer than 192 tiles.  Never run so far.
timeout 150 python tools/tail_ab.py --on 43 --rounds 3 --cases 4096x4096x512,4096x4096x768a,4096x4096x1024a,11008x4096x512,11008x4096x1024a,5120x5120x640,8192x8192x384 > $O/ksplit_in_launch_ab.log 2>&1
# (2) the balanced tail above its 1024-tile limit, more rounds (round 3: +5..7 % in this tool, -4 % inside bench.py)
timeout 150 python tools/tail_ab.py --on 42 --rounds 5 --cases 4096x11008x3072a,4096x11008x3840,4096x11008x4096a,4096x4096x8320,4096x4096x16512 > $O/tail_above_limit_ab.log 2>&1
tail -3 $O/gpu_tests.log; head -c 300 $O/bench.json; echo; grep -v amdgpu $O/ksplit_in_launch_ab.log $O/tail_above_limit_ab.log
