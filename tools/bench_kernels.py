#!/usr/bin/env python3
"""Micro-benchmarks of the libmvhip.so kernels (HIP-event timed, algorithmic GB/s)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd import hip_ops as H

DEV = "cuda:0"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bench_decode(B, Hq, Hkv, D, ctx, bs=64, dtype=torch.bfloat16, layers=4, flags=None):
    nblk = (ctx + bs - 1) // bs
    nb = B * nblk
    # several "layers" of cache so that successive launches do not hit in the 256 MiB Infinity Cache
    caches = [(torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.float32).to(dtype),
               torch.randn(nb, bs, Hkv, D, device=DEV, dtype=torch.float32).to(dtype)) for _ in range(layers)]
    q = torch.randn(B, Hq, D, device=DEV).to(dtype)
    table = torch.randperm(nb, device=DEV).to(torch.int32).view(B, nblk)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    out = torch.empty_like(q)
    state = {"i": 0}

    def fn():
        k, v = caches[state["i"] % layers]
        state["i"] += 1
        H.paged_decode(q, k, v, table, cl, D ** -0.5, out=out, flags=flags)
    t = timeit(fn)
    esz = 2 if dtype != torch.float32 else 4
    bytes_ = B * ctx * 2 * Hkv * D * esz + 2 * B * Hq * D * esz + B * nblk * 4
    print(f"paged_decode B={B} H={Hq}/{Hkv} D={D} ctx={ctx} flags={flags}: {t*1e6:8.1f} us  {bytes_/t/1e9:8.1f} GB/s  ({bytes_/1e6:.1f} MB)")
    return t, bytes_


def bench_prefill(n_seqs, seqlen, Hq, Hkv, D=128, dtype=torch.bfloat16, variant=0):
    T = n_seqs * seqlen
    q = torch.randn(T, Hq, D, device=DEV).to(dtype)
    k = torch.randn(T, Hkv, D, device=DEV).to(dtype)
    v = torch.randn(T, Hkv, D, device=DEV).to(dtype)
    cu = (torch.arange(n_seqs + 1, device=DEV) * seqlen).to(torch.int32)
    out = torch.empty_like(q)
    t = timeit(lambda: H.prefill_varlen(q, k, v, cu, cu, seqlen, seqlen, D ** -0.5, out=out, flags=variant << 4), iters=10)
    flops = n_seqs * 4 * Hq * D * (seqlen * seqlen - seqlen * (seqlen - 1) / 2)
    print(f"prefill {n_seqs}x{seqlen} H={Hq}/{Hkv} variant {variant}: {t*1e3:8.3f} ms  {flops/t/1e12:7.1f} TFLOP/s  {T/t/1e6:6.2f} Mtok/s")


def bench_sampler(B, V, temp, top_k, top_p, min_p=0.0, dtype=torch.bfloat16):
    logits = (torch.randn(B, V, device=DEV) * 3).to(dtype)
    t_ = torch.full((B,), temp, device=DEV)
    k_ = torch.full((B,), top_k, dtype=torch.int64, device=DEV)
    p_ = torch.full((B,), top_p, device=DEV)
    m_ = torch.full((B,), min_p, device=DEV)
    t = timeit(lambda: H.sample(logits, t_, k_, p_, m_, seed=1, offset=2), iters=20)
    bytes_ = B * V * logits.element_size()
    print(f"sampler B={B} V={V} T={temp} k={top_k} p={top_p} minp={min_p}: {t*1e6:8.1f} us  ({bytes_/t/1e9:7.1f} GB/s per single pass)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--prefill-only", action="store_true")
    ap.add_argument("--decode-ab", action="store_true", help="paged decode shapes, warmed up and repeated")
    ap.add_argument("--variants", default="0", help="prefill kernel bench variants (flags >> 4), comma separated")
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0))
    if a.decode_ab:
        for _ in range(2):
            bench_decode(256, 16, 8, 128, 1024)
        print("--- (warm-up above) ---")
        for rep in range(3):
            # flag 16 = one kv head per workgroup (the round-1 mapping); bits 8-15 = context-split target / 16
            # (round 2: an XCD-aware grid order was tried here: 172 -> 173 us, dropped)
            for fl in (16, 0):
                bench_decode(256, 16, 8, 128, 1024, flags=fl)
                bench_decode(64, 16, 8, 128, 4096, layers=2, flags=fl)
                bench_decode(8, 16, 8, 128, 4096, layers=2, flags=fl)
            bench_decode(256, 28, 4, 128, 1024, flags=0)
            bench_decode(256, 8, 1, 128, 1024, flags=0)
        raise SystemExit(0)
    if a.prefill_only:
        # the chip's clock ramps over the first second of load: warm up, then interleave the variants and repeat
        for _ in range(3):
            bench_prefill(4, 4096, 28, 4)
        print("--- (warm-up above) ---")
        for rep in range(3):
            for var in [int(v) for v in a.variants.split(",")]:
                bench_prefill(16, 1024, 16, 8, variant=var)
                bench_prefill(16, 1024, 28, 4, variant=var)
                bench_prefill(4, 4096, 28, 4, variant=var)
                bench_prefill(64, 256, 28, 4, variant=var)
        raise SystemExit(0)
    bench_decode(256, 16, 8, 128, 1024)          # Qwen3-1.7B headline
    bench_prefill(16, 1024, 16, 8)               # Qwen3-1.7B, one 16384-token batch
    bench_prefill(16, 1024, 28, 4)               # Qwen2-7B config 4
    bench_prefill(4, 4096, 28, 4)
    if not a.quick:
        bench_sampler(256, 151936, 0.0, -1, 1.0)           # greedy
        bench_sampler(256, 151936, 1.0, -1, 1.0)           # plain multinomial
        bench_sampler(256, 151936, 0.6, 40, 0.95)          # the reference's cpu_inference_opt.py preset
        bench_sampler(256, 151936, 0.8, -1, 0.9, 0.05)     # nucleus over the full vocabulary + min-p
        bench_sampler(8, 151936, 0.6, 40, 0.95)
        bench_decode(256, 16, 8, 128, 4096, layers=2)
        bench_decode(256, 28, 4, 128, 1024)      # Qwen2-7B
        bench_decode(256, 8, 1, 128, 1024)       # 72B @ TP8
        bench_decode(64, 16, 8, 128, 1024)
        bench_decode(8, 16, 8, 128, 1024)
        bench_decode(1, 16, 8, 128, 4096)
        bench_decode(8, 28, 4, 128, 4096)
