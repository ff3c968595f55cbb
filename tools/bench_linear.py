#!/usr/bin/env python3
"""Decode-sized linear layers: mv_linear_small_m vs the library GEMM (F.linear -> hipBLASLt), per
projection shape of Qwen3-1.7B / Qwen2-7B and per batch size.  Every timed call streams a different
weight copy (a pool larger than the 256 MB Infinity Cache), as consecutive layers of a decode step do.

    python tools/bench_linear.py [--models qwen3-1.7b,qwen2-7b] [--ms 1,8,16,32,64]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd import hip_ops  # noqa: E402

SHAPES = {  # name -> (N, K, bias)
    "qwen3-1.7b": {"qkv": (4096, 2048, False), "o": (2048, 2048, False), "gate_up": (12288, 2048, False),
                   "down": (2048, 6144, False), "lm_head": (151936, 2048, False)},
    "qwen2-7b": {"qkv": (4608, 3584, True), "o": (3584, 3584, False), "gate_up": (37888, 3584, False),
                 "down": (3584, 18944, False), "lm_head": (152064, 3584, False)},
}


def timed(fn, n_calls):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(n_calls):
            fn(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(n_calls):
                fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_calls)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="qwen3-1.7b,qwen2-7b")
    ap.add_argument("--ms", default="1,8,16,32,64")
    ap.add_argument("--no-sweep-packed", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--xlds", type=int, default=0, help="1 / -1: force x through LDS on / off")
    ap.add_argument("--wlds", type=int, default=0, help="1 / -1: force W through LDS on / off")
    ap.add_argument("--packed", action="store_true", help="also time the kernel on the fragment-ordered weight copy (and sweep its launch shape)")
    ap.add_argument("--sweep", action="store_true", help="sweep row tiles per wave x waves per shape (explicit launch-shape argument)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    print(f"{'model':11s} {'proj':8s} {'M':>3s} {'N':>7s} {'K':>6s} {'MB':>7s} | {'lib us':>8s} {'TB/s':>6s} | {'mv us':>8s} {'TB/s':>6s} | {'x':>5s} maxerr")
    for model in a.models.split(","):
        for name, (n, k, has_bias) in SHAPES[model].items():
            if a.only and name not in a.only.split(","):
                continue
            wbytes = n * k * 2
            copies = max(2, min(64, (600 << 20) // wbytes + 1))
            ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(copies)]
            bias = (torch.randn(n, device=dev) * 0.1).to(torch.bfloat16) if has_bias else None
            for m in [int(v) for v in a.ms.split(",")]:
                x = torch.randn(m, k, device=dev).to(torch.bfloat16)
                ref = (x.double() @ ws[0].double().t())
                if bias is not None:
                    ref = ref + bias.double()
                got = hip_ops.linear_small_m(x, ws[0], bias).double()
                lib = F.linear(x, ws[0], bias).double()
                err, err_lib = (got - ref).abs().max().item(), (lib - ref).abs().max().item()
                ncall = max(copies, 16)
                t_lib = timed(lambda i: F.linear(x, ws[i % copies], bias), ncall)
                if a.sweep:
                    res = []
                    for rpw in (1, 2, 4):
                        for waves in (2, 4, 8, 16):
                            if waves * rpw * ((m + 15) // 16) > 64:
                                continue
                            shape = hip_ops.small_m_shape(rpw, waves)
                            res.append((timed(lambda i: hip_ops.linear_small_m(x, ws[i % copies], bias, shape), ncall), rpw, waves))
                    res.sort()
                    print(f"   sweep {name} M={m}: " + "  ".join(f"r{r}w{w}={t:.1f}" for t, r, w in res[:6]), flush=True)
                forced = hip_ops.small_m_shape(0, 0, a.xlds, a.wlds)
                t_mv = timed(lambda i: hip_ops.linear_small_m(x, ws[i % copies], bias, forced), ncall)
                tail = ""
                if a.packed:
                    wps = [hip_ops.pack_weight(w) for w in ws]
                    same = torch.equal(hip_ops.linear_small_m(x, ws[0], bias, packed=wps[0]), hip_ops.linear_small_m(x, ws[0], bias))
                    t_pk = timed(lambda i: hip_ops.linear_small_m(x, ws[i % copies], bias, 0, wps[i % copies]), ncall)
                    res = []
                    for rpw in (() if a.no_sweep_packed else (1, 2)):
                        for waves in (2, 4, 6, 7, 8, 12, 14, 16):
                            if waves * rpw * ((m + 15) // 16) > 64:
                                continue
                            shape = hip_ops.small_m_shape(rpw, waves)
                            res.append((timed(lambda i: hip_ops.linear_small_m(x, ws[i % copies], bias, shape, wps[i % copies]), ncall), rpw, waves))
                    res.sort()
                    tail = (f" | packed {t_pk:7.2f} us {wbytes/t_pk/1e6:5.2f} TB/s same={same} best " +
                            " ".join(f"r{r}w{w}={t:.1f}" for t, r, w in res[:7]))
                    del wps
                print(f"{model:11s} {name:8s} {m:3d} {n:7d} {k:6d} {wbytes/1e6:7.1f} | {t_lib:8.2f} {wbytes/t_lib/1e6:6.2f} | "
                      f"{t_mv:8.2f} {wbytes/t_mv/1e6:6.2f} | {t_lib/t_mv:5.2f} {err:.3g} (lib {err_lib:.3g}){tail}", flush=True)
                if a.packed and name == "gate_up":
                    # the gated form the models actually run (gate_up + SiluAndMul): row-major vs packed, launch shapes
                    wps = [hip_ops.pack_weight(w) for w in ws]
                    t_rm = timed(lambda i: hip_ops.gate_up_silu_small_m(x, ws[i % copies]), ncall)
                    t_pk = timed(lambda i: hip_ops.gate_up_silu_small_m(x, ws[i % copies], 0, wps[i % copies]), ncall)
                    res = []
                    for rpw in (1, 2):
                        for waves in (2, 4, 7, 8, 14, 16):
                            shape = hip_ops.small_m_shape(rpw, waves)
                            res.append((timed(lambda i: hip_ops.gate_up_silu_small_m(x, ws[i % copies], shape, wps[i % copies]), ncall), rpw, waves))
                    res.sort()
                    print(f"{model:11s} gated    {m:3d} {n // 2:7d} {k:6d} {wbytes/1e6:7.1f} | row-major {t_rm:7.2f} us | packed {t_pk:7.2f} us "
                          f"{wbytes/t_pk/1e6:5.2f} TB/s best " + " ".join(f"r{r}w{w}={t:.1f}" for t, r, w in res[:7]), flush=True)
                    del wps
            del ws


if __name__ == "__main__":
    main()
