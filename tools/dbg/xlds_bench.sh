#!/bin/bash
# A/B of the x-through-LDS form of linear_small_m (MV_LINEAR_TUNE lets the override through)
for X in -1 1; do echo "XLDS=$X"; MV_LINEAR_TUNE=1 MV_LINEAR_XLDS=$X python tools/bench_linear.py --ms 4,8,16,32 2>/dev/null | grep -v "^model\|amdgpu"; done
