"""Sampler timing on the generic radix paths (fp32 logits; 16-bit rows whose thresholds fall below the window)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from minivllm_amd import hip_ops as H

dev = torch.device("cuda", 0)
B, V = 256, 151936
g = torch.Generator().manual_seed(0)
base = torch.randn(B, V, generator=g)
temps = torch.full((B,), 0.8, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1e3, 1)


out = {}
for name, logits in (("f32_peaked", (base * 2.5).to(dev)), ("bf16_peaked", (base * 2.5).bfloat16().to(dev)),
                     ("bf16_flat", (base * 0.05).bfloat16().to(dev)), ("f16_flat", (base * 0.05).half().to(dev))):
    for label, k, p in (("k50_p0.9", 50, 0.9), ("p0.9", -1, 0.9), ("k5000", 5000, 1.0), ("plain", -1, 1.0)):
        ks = torch.full((B,), k, dtype=torch.int64, device=dev)
        ps = torch.full((B,), p, device=dev)
        out[f"{name}/{label}"] = timed(lambda: H.sample(logits, temps, ks, ps, None, seed=1, offset=1))
print(os.environ.get("MVHIP_LIB", "default"), json.dumps(out, indent=1))
