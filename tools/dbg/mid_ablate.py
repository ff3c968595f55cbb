#!/usr/bin/env python3
"""Timing ablations of linear_mid_m_kernel (cfg + 100 * flags; flags: 1 ring only, 4 no x DMA, 8 no W DMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench_mid import timed
from minivllm_amd import hip_ops

dev = torch.device("cuda")
shapes = [("gate_up(plain)", 12288, 2048, [22, 26, 27, 28, 29])]
m = 256
for name, n, k, cfgs in shapes:
    wbytes = n * k * 2
    copies = max(2, min(48, (600 << 20) // wbytes + 1))
    wps = [hip_ops.pack_weight((torch.randn(n, k, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ncall = max(copies, 16)
    print(f"== {name} M={m} N={n} K={k} {wbytes/1e6:.1f} MB", flush=True)
    for c in cfgs:
        row = []
        for fl, label in ((0, "full"), (4, "no x"), (8, "no W"), (12, "compute only"), (1, "ring only"), (5, "ring no x"), (9, "ring no W"), (13, "barriers only")):
            t = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, None, cfg=c + 100 * fl, out=out), ncall)
            row.append(f"{label} {t:6.2f}")
        print(f"   cfg {c:2d}: " + " | ".join(row), flush=True)
    del wps
