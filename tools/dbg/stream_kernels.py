"""Streaming kernels of a 16384-token prefill step (Qwen2-7B / Qwen3-1.7B dims): achieved HBM bandwidth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from minivllm_amd import hip_ops as H
DEV = "cuda:0"

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

T = 16384
for name, hid, inter in (("qwen2-7b", 3584, 18944), ("qwen3-1.7b", 2048, 6144)):
    x = torch.randn(T, 2 * inter, device=DEV).bfloat16()
    t = timeit(lambda: H.silu_and_mul(x))
    print(f"{name} silu_and_mul [{T}, 2x{inter}]: {t*1e6:7.1f} us  {(3 * T * inter * 2) / t / 1e12:5.2f} TB/s")
    h = torch.randn(T, hid, device=DEV).bfloat16(); r = torch.randn(T, hid, device=DEV).bfloat16()
    w = torch.ones(hid, device=DEV).bfloat16()
    t = timeit(lambda: H.add_rmsnorm(h, r, w, 1e-6))
    print(f"{name} add_rmsnorm [{T}, {hid}]: {t*1e6:7.1f} us  {(4 * T * hid * 2) / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: H.rmsnorm(h, w, 1e-6))
    print(f"{name} rmsnorm [{T}, {hid}]: {t*1e6:7.1f} us  {(2 * T * hid * 2) / t / 1e12:5.2f} TB/s")
    del x, h, r
