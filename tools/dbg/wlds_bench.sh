#!/bin/bash
for X in -1 1; do echo "WLDS=$X"; python tools/bench_linear.py --wlds $X --ms 1,2 --only gate_up,lm_head 2>/dev/null | grep -v "^model\|amdgpu" | cut -c1-120; done
