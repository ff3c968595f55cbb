"""Which rows / columns of the 64-rows-per-wave prefill kernel (variant 5) differ from variant 2, on a few small shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from minivllm_amd import hip_ops as H

dev = torch.device("cuda", 0)
for (S, hq, hkv) in ((257, 8, 1), (256, 2, 1), (512, 2, 1), (64, 1, 1), (128, 1, 1), (1000, 4, 2), (4096, 2, 1)):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(S, hq, 128, generator=g).bfloat16().to(dev)
    k = torch.randn(S, hkv, 128, generator=g).bfloat16().to(dev)
    v = torch.randn(S, hkv, 128, generator=g).bfloat16().to(dev)
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    a = H.prefill_varlen(q, k, v, cu, cu, S, S, 128 ** -0.5, flags=2 << 4).float()
    for var in (5,):
        b = H.prefill_varlen(q, k, v, cu, cu, S, S, 128 ** -0.5, flags=var << 4).float()
        err = (a - b).abs()
        print("variant", var, end=": ")
        bad_rows = (err.amax(dim=(1, 2)) > 0.05).nonzero().flatten().tolist()
        print(f"S={S} heads {hq}/{hkv}: max err {err.max().item():.3g}; bad rows {len(bad_rows)}: {bad_rows[:40]}")
        if bad_rows:
            r0 = bad_rows[0]
            bad_cols = (err[r0].amax(dim=0) > 0.05).nonzero().flatten().tolist()
            bad_heads = (err[r0].amax(dim=1) > 0.05).nonzero().flatten().tolist()
            print(f"   row {r0}: bad heads {bad_heads}, bad cols {bad_cols[:64]}; nan {torch.isnan(b).sum().item()}")
