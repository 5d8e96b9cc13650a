#!/bin/bash
# round 6: the 3- / 8-bit and 32-wide-group forms of the panel kernel -- forced-geometry parity, then the A/B against the default plan (HBM-cold rotating layers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_panel.py -x -q -k "3bit or forced" 2>&1 | tail -6
G="--geoms 0,21,22,23,24 --check 1 --rounds 2"
for SPEC in "3 32" "8 32" "4 32" "3 128" "8 128"; do
  set -- $SPEC
  timeout 900 python tools/panel_ab.py --bits $1 --gs $2 --ms 64,96,128,192,256,384,512,768 --shapes 4096x4096,4096x11008,11008x4096 $G 2>&1 | grep -v amdgpu.ids
done
