#!/bin/bash
# round 6: the panel kernel against the planner's choice without it over row counts and model shapes, HBM-cold rotating layers (planner rule: panel_pays / plan_panel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export GPTQ_LAB_NO_PANEL=1
G="--geoms 0,21,22,23,24 --check 0 --rounds 2"
timeout 1200 python tools/panel_ab.py --ms 64,96,128,160,192,256,320,384,448,512,640,768 --shapes 4096x4096,4096x11008,11008x4096 $G 2>&1 | grep -v amdgpu.ids
timeout 1200 python tools/panel_ab.py --ms 128,256,512,768 --shapes 2048x2048,5120x5120,8192x8192,5120x13824,13824x5120,8192x1024,1024x8192,8192x3584,28672x1024 $G 2>&1 | grep -v amdgpu.ids
