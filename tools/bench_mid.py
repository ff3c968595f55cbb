#!/usr/bin/env python3
"""Decode-batch projections (M = 64..256): mv_linear_mid_m per workgroup tile vs the library GEMM (F.linear ->
hipBLASLt), per projection shape.  Every timed call streams a different weight copy (a pool larger than the
256 MB Infinity Cache), as consecutive layers of a decode step do.

    python tools/bench_mid.py [--models qwen3-1.7b,qwen2-7b] [--ms 256] [--cfgs 1,2,3] [--ingest] [--splits 2,4]
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
    "qwen2-72b-tp8": {"qkv": (1280, 8192, True), "o": (8192, 1024, False), "gate_up": (7392, 8192, False),
                      "lm_head": (19008, 8192, False)},
}
from minivllm_amd.layers.linear import MID_CFGS  # noqa: E402
CFGS = {c: (bm, bn) for c, (bm, bn, _) in MID_CFGS.items()}
GATED_OK = tuple(c for c, (_, _, g) in MID_CFGS.items() if g)


def timed(fn, n_calls, reps=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(min(n_calls, 3)):
            fn(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for i in range(n_calls):
                fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_calls)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="qwen3-1.7b,qwen2-7b")
    ap.add_argument("--ms", default="256")
    ap.add_argument("--only", default="")
    ap.add_argument("--cfgs", default="1,2,3,4,5,6,7,8,9,10")
    ap.add_argument("--norot", action="store_true", help="also time WITH the K-loop rotation (cfg + 200)")
    ap.add_argument("--ingest", action="store_true", help="also time the DMA ring alone (cfg + 100)")
    ap.add_argument("--splits", default="", help="also time the K-split partial-sum form with these split counts")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cfgs = [int(c) for c in a.cfgs.split(",")]
    for model in a.models.split(","):
        for name, (n, k, has_bias) in SHAPES[model].items():
            if a.only and name not in a.only.split(","):
                continue
            wbytes = n * k * 2
            copies = max(2, min(64, (600 << 20) // wbytes + 1))
            ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(copies)]
            wps = [hip_ops.pack_weight(w) for w in ws]
            assert torch.equal(wps[0].view(-1, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(-1, k)[:n], ws[0])
            bias = (torch.randn(n, device=dev) * 0.1).to(torch.bfloat16) if has_bias else None
            ncall = max(copies, 16)
            for m in [int(v) for v in a.ms.split(",")]:
                x = torch.randn(m, k, device=dev).to(torch.bfloat16)
                ref = x.double() @ ws[0].double().t()
                if bias is not None:
                    ref = ref + bias.double()
                lib = F.linear(x, ws[0], bias)
                err_lib = (lib.double() - ref).abs().max().item()
                t_lib = timed(lambda i: F.linear(x, ws[i % copies], bias), ncall)
                print(f"{model} {name} M={m} N={n} K={k} {wbytes/1e6:.1f} MB | lib {t_lib:.2f} us "
                      f"({wbytes/t_lib/1e6:.2f} TB/s, {2*m*n*k/t_lib/1e6:.0f} TF) err {err_lib:.3g}", flush=True)
                res = []
                for c in cfgs:
                    bm, bn = CFGS[c]
                    wgs = -(-m // bm) * -(-n // bn)
                    if wgs > 600 or wgs < 32:
                        continue
                    got = hip_ops.linear_mid_m(x, wps[0], n, bias, cfg=c)
                    err = (got.double() - ref).abs().max().item()
                    same = (got == lib).float().mean().item()
                    t = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, bias, cfg=c), ncall)
                    line = f"   cfg {c} {bm}x{bn} wgs {wgs:4d}: {t:7.2f} us  x{t_lib/t:.2f}  err {err:.3g} same-as-lib {same:.4f}"
                    if a.ingest:
                        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
                        ti = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, None, cfg=c + 100, out=out), ncall)
                        line += f"  | ring alone {ti:7.2f} us = {(bm + bn) * k * 2 / ti / 1e3:.0f} GB/s/WG"
                    if a.norot:
                        tn = timed(lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, bias, cfg=c + 200), ncall)
                        line += f"  | rotated K loop {tn:7.2f} us"
                    print(line, flush=True)
                    res.append((t, c))
                for sp in [int(v) for v in a.splits.split(",") if v]:
                    for c in cfgs:
                        bm, bn = CFGS[c]
                        wgs = -(-m // bm) * -(-n // bn) * sp
                        if wgs > 1024 or wgs < 32 or (k // 64) % sp:
                            continue
                        part = hip_ops.linear_mid_m_partial(x, wps[0], n, sp, cfg=c)
                        err = (part.double().sum(0) + (bias.double() if bias is not None else 0) - ref).abs().max().item()
                        t = timed(lambda i: hip_ops.linear_mid_m_partial(x, wps[i % copies], n, sp, cfg=c), ncall)
                        print(f"   split {sp} cfg {c} {bm}x{bn} wgs {wgs:4d}: {t:7.2f} us  x{t_lib/t:.2f}  err {err:.3g}", flush=True)
                if name == "gate_up":
                    ni = n // 2
                    act_ref = F.silu(lib[:, :ni].float()).to(torch.bfloat16).float() * lib[:, ni:].float()
                    t_unf = timed(lambda i: hip_ops.silu_and_mul(F.linear(x, ws[i % copies])), ncall)
                    print(f"   gate_up + silu, library + mv_silu_and_mul: {t_unf:.2f} us", flush=True)
                    for c in cfgs:
                        if c not in GATED_OK:
                            continue
                        bm, bn = CFGS[c]
                        wgs = -(-m // bm) * -(-ni // (bn // 2))
                        if wgs > 1024 or wgs < 32:
                            continue
                        got = hip_ops.gate_up_silu_mid_m(x, wps[0], ni, cfg=c)
                        unf = hip_ops.silu_and_mul(hip_ops.linear_mid_m(x, wps[0], n, cfg=c))
                        t = timed(lambda i: hip_ops.gate_up_silu_mid_m(x, wps[i % copies], ni, cfg=c), ncall)
                        print(f"   gated cfg {c} {bm}x{bn} wgs {wgs:4d}: {t:7.2f} us  x{t_unf/t:.2f}  "
                              f"bit-equal to unfused mid_m+silu: {bool((got == unf).all())}  "
                              f"max |d| vs lib path {(got.float() - act_ref).abs().max().item():.3g}", flush=True)
            del ws, wps


if __name__ == "__main__":
    main()
