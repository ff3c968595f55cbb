#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic of the decode kernel).  Launches, in order:
a 1 GiB device copy (calibration: exactly 2^30 bytes read and written by a wide coalesced stream),
then the paged decode kernel on the Qwen3-1.7B and Qwen2-7B head layouts at B=256, ctx=1024.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd import hip_ops as H

DEV = "cuda:0"
src = torch.randn(1 << 29, device=DEV, dtype=torch.float32).view(torch.bfloat16)  # 2 GiB? no: 2^29 f32 = 2 GiB
src = src[: 1 << 29]                                                              # 2^29 bf16 = 1 GiB
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()


def decode(B, Hq, Hkv, ctx, D=128, bs=64, reps=3):
    nblk = ctx // bs
    nb = B * nblk
    caches = [(torch.randn(nb, bs, Hkv, D, device=DEV).bfloat16(), torch.randn(nb, bs, Hkv, D, device=DEV).bfloat16())
              for _ in range(reps)]
    q = torch.randn(B, Hq, D, device=DEV).bfloat16()
    table = torch.randperm(nb, device=DEV).to(torch.int32).view(B, nblk)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    for k, v in caches:
        H.paged_decode(q, k, v, table, cl, D ** -0.5)
    torch.cuda.synchronize()


decode(256, 16, 8, 1024)
decode(256, 28, 4, 1024)
print("done")
