#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc FETCH_SIZE pass over the weight-streaming linear kernel: LM head
(151936 x 2048, 622 MB) at M = 1 and 16, gate_up+SiluAndMul (2 x 6144 x 2048, 50 MB) at M = 1 and 16, each on
3 different weight copies.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_probe_linear.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd import hip_ops as H

DEV = "cuda:0"
lm = [(torch.randn(151936, 2048, device=DEV) * 0.02).bfloat16() for _ in range(3)]
gu = [(torch.randn(12288, 2048, device=DEV) * 0.02).bfloat16() for _ in range(3)]
for m in (1, 16):
    x = torch.randn(m, 2048, device=DEV).bfloat16()
    for w in lm:
        H.linear_small_m(x, w)
    for w in gu:
        H.gate_up_silu_small_m(x, w)
torch.cuda.synchronize()
print("done")
