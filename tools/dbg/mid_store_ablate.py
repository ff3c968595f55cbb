#!/usr/bin/env python3
"""How much of linear_mid_m_kernel is its output store?  (cfg + 3200: every lane but one skips the store; timing only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench_mid import timed
from minivllm_amd import hip_ops
from minivllm_amd.layers import linear as L

dev = torch.device("cuda")
m = 256
for name, n, k in (("qkv 1.7B", 4096, 2048), ("gate_up(plain) 1.7B", 12288, 2048), ("qkv 7B", 4608, 3584), ("lm_head 1.7B", 151936, 2048)):
    wbytes = n * k * 2
    copies = max(2, min(48, (600 << 20) // wbytes + 1))
    wps = [hip_ops.pack_weight((torch.randn(n, k, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ncall = max(copies, 16)
    cfgs = [L.mid_plan("col", m, n, k)[0]]
    for c in cfgs:
        t0 = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, None, cfg=c, out=out), ncall)
        t1 = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, None, cfg=c + 3200, out=out), ncall)
        print(f"{name:22s} cfg {c:2d}: full {t0:7.2f} us | no stores {t1:7.2f} us | store share {100 * (t0 - t1) / t0:5.1f} %  (output {m * n * 2 / 1e6:.1f} MB)", flush=True)
    del wps
