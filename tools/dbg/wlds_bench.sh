#!/bin/bash
for X in 0 1; do echo "WLDS=$X"; MV_LINEAR_TUNE=1 MV_LINEAR_WLDS=$X python tools/bench_linear.py --ms 1,2 --only gate_up,lm_head 2>/dev/null | grep -v "^model\|amdgpu" | cut -c1-120; done
