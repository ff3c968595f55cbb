#!/bin/bash
# A/B of the x-through-LDS form of linear_small_m (explicit launch-shape argument)
for X in -1 1; do echo "XLDS=$X"; python tools/bench_linear.py --xlds $X --ms 4,8,16,32 2>/dev/null | grep -v "^model\|amdgpu"; done
