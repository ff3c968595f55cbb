#!/usr/bin/env python3
"""In-kernel clock of linear_mid_m_kernel's ablation variants (flags + 16: stamps around the K loop of wave 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from minivllm_amd import hip_ops

dev = torch.device("cuda")
m, n, k = 256, 12288, 2048
copies = 12
wps = [hip_ops.pack_weight((torch.randn(n, k, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
x = torch.randn(m, k, device=dev).bfloat16()
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
dbg = torch.zeros(n, device=dev, dtype=torch.bfloat16)
for c in (22, 14):
    for fl, label in ((0, "full"), (4, "no x"), (8, "no W"), (12, "compute only"), (1, "ring only"), (5, "ring no x"), (9, "ring no W")):
        # keep the chip busy for a while with the same variant, then stamp
        for rep in range(3):
            for i in range(200):
                hip_ops.linear_mid_m(x, wps[i % copies], n, None, cfg=c + 100 * fl, out=out)
        dbg.zero_()
        hip_ops.linear_mid_m(x, wps[0], n, dbg, cfg=c + 100 * (fl + 16), out=out)
        torch.cuda.synchronize()
        v = dbg.view(torch.int64)[:512].view(256, 2).cpu()
        cyc, real = v[:, 0].float(), v[:, 1].float()
        ok = real > 0
        clk = (cyc[ok] / real[ok] * 100).median().item()
        print(f"cfg {c} {label:13s}: K loop {real[ok].median().item() / 100:7.2f} us ({cyc[ok].median().item():8.0f} cycles) clock {clk:6.0f} MHz", flush=True)
