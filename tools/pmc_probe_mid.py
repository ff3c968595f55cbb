#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over the decode-batch projection kernel (mv_linear_mid_m): a 1 GiB
device copy (calibration), then the gate_up + SiluAndMul launch of Qwen3-1.7B and of Qwen2-7B at M = 256 and the
K-split down projection + its norm, each on several weight copies (cold weights, as in a decode step).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_probe_mid.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd import hip_ops as H
from minivllm_amd.layers.linear import mid_plan

DEV = "cuda:0"
src = torch.randn(1 << 28, device=DEV, dtype=torch.float32).view(torch.bfloat16)   # 2^29 bf16 = 1 GiB
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst


def run(kind, m, n, k, copies=4):
    rows = 2 * n if kind == "gated" else n
    ws = [H.pack_weight((torch.randn(rows, k, device=DEV) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(m, k, device=DEV).bfloat16()
    res = torch.randn(m, n, device=DEV).bfloat16()
    nw = torch.ones(n, device=DEV, dtype=torch.bfloat16)
    cfg, splits = mid_plan(kind, m, n, k)
    for w in ws:
        if kind == "gated":
            H.gate_up_silu_mid_m(x, w, n, cfg)
        elif splits > 1:
            H.add_rmsnorm_partial(H.linear_mid_m_partial(x, w, n, splits, cfg), res, nw, 1e-6)
        else:
            H.linear_mid_m(x, w, n, None, cfg)
    torch.cuda.synchronize()
    print(kind, m, n, k, "cfg", cfg, "splits", splits, "weight bytes", rows * k * 2)


run("gated", 256, 6144, 2048)
run("row", 256, 2048, 6144)
run("gated", 256, 18944, 3584)
run("row", 256, 3584, 18944)
print("done")
