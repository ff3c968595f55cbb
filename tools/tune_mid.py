#!/usr/bin/env python3
"""Measure, per projection shape and decode batch size, which form runs fastest on this GPU: the library GEMM
(F.linear -> hipBLASLt) or mv_linear_mid_m with one of its workgroup tiles / K splits — and write the table
layers/linear.py::mid_plan looks up (mini-vllm_amd/tuned/mid_m_gfx950.json).

Kinds: 'col' = plain projection; 'gated' = gate_up + SiluAndMul (fused epilogue vs library + mv_silu_and_mul);
'row' = projection + the residual-add + RMSNorm behind it (a K split hands fp32 partial sums to the norm kernel).
Every timed call streams a different weight copy (a pool larger than the Infinity Cache), inside a hipGraph.

    python tools/tune_mid.py --out gpurun_out/mid_m_gfx950.json [--models qwen3-1.7b,qwen2-7b] [--ms 64,128,192,256]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd import hip_ops  # noqa: E402
from minivllm_amd.layers.linear import MID_CFGS  # noqa: E402

# model -> [(kind, N (gated: intermediate size), K, bias)]
SHAPES = {
    "qwen3-0.6b": [("col", 4096, 1024, False), ("row", 1024, 2048, False), ("gated", 3072, 1024, False),
                   ("row", 1024, 3072, False)],
    "qwen3-1.7b": [("col", 4096, 2048, False), ("row", 2048, 2048, False), ("gated", 6144, 2048, False),
                   ("row", 2048, 6144, False)],
    "qwen2-7b": [("col", 4608, 3584, True), ("row", 3584, 3584, False), ("gated", 18944, 3584, False),
                 ("row", 3584, 18944, False)],
    "lm-heads": [("col", 151936, 1024, False), ("col", 151936, 2048, False), ("col", 152064, 3584, False),
                 ("col", 19008, 8192, False)],
    # the per-rank shards of Qwen3-1.7B under TP = 2 / 4 / 8 (bench.py --gpus N)
    "qwen3-1.7b-tp": [("col", 2048, 2048, False), ("row", 2048, 1024, False), ("gated", 3072, 2048, False), ("row", 2048, 3072, False),
                      ("col", 1024, 2048, False), ("row", 2048, 512, False), ("gated", 1536, 2048, False), ("row", 2048, 1536, False),
                      ("col", 512, 2048, False), ("row", 2048, 256, False), ("gated", 768, 2048, False), ("row", 2048, 768, False),
                      ("col", 75968, 2048, False), ("col", 37984, 2048, False), ("col", 18992, 2048, False),
                      # (under TP the row-parallel outputs go to the all-reduce, not to a K-split norm: plain form)
                      ("col", 2048, 1024, False), ("col", 2048, 3072, False), ("col", 2048, 512, False),
                      ("col", 2048, 1536, False), ("col", 2048, 256, False), ("col", 2048, 768, False)],
    "deep-k": [("row", 3584, 18944, False), ("row", 8192, 29568, False), ("row", 8192, 8192, False)],
    "qwen2-72b-tp8": [("col", 1280, 8192, True), ("row", 8192, 1024, False), ("gated", 3696, 8192, False)],
    "qwen2-72b": [("col", 10240, 8192, True), ("row", 8192, 8192, False), ("gated", 29568, 8192, False),
                  ("row", 8192, 29568, False)],
}


def timed(fn, n_calls, reps=4):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(0)
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
    ap.add_argument("--ms", default="64,128,192,256")
    ap.add_argument("--out", default="gpurun_out/mid_m_gfx950.json")
    ap.add_argument("--merge", default="", help="existing table to extend")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    plans = {}
    if a.merge and os.path.exists(a.merge):
        plans = json.load(open(a.merge)).get("plans", {})
    t_start = time.time()
    for model in a.models.split(","):
        for kind, n, k, has_bias in SHAPES[model]:
            n_rows = 2 * n if kind == "gated" else n
            if k % 64 or (kind == "gated" and n % 16):
                print(f"skip {model} {kind} N={n} K={k}: not a mid-M shape", flush=True)
                continue
            wbytes = n_rows * k * 2
            copies = max(2, min(48, (400 << 20) // wbytes + 1))   # (>= 2 copies: 1.2 - 2.2 GB for an LM head)
            ws = [(torch.randn(n_rows, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(copies)]
            wps = [hip_ops.pack_weight(w) for w in ws]
            bias = (torch.randn(n_rows, device=dev) * 0.1).to(torch.bfloat16) if has_bias else None
            nw = torch.ones(n, device=dev, dtype=torch.bfloat16)
            ncall = max(copies, 12)
            for m in [int(v) for v in a.ms.split(",")]:
                key = f"{kind}:{m}:{n}:{k}"
                x = torch.randn(m, k, device=dev).to(torch.bfloat16)
                res = torch.randn(m, n, device=dev).to(torch.bfloat16)
                if kind == "col":
                    t_lib = timed(lambda i: F.linear(x, ws[i % copies], bias), ncall)
                elif kind == "gated":
                    t_lib = timed(lambda i: hip_ops.silu_and_mul(F.linear(x, ws[i % copies])), ncall)
                else:
                    t_lib = timed(lambda i: hip_ops.add_rmsnorm(F.linear(x, ws[i % copies]), res, nw, 1e-6), ncall)
                cands = [(t_lib, 0, 1)]
                for c, (bm, bn, gated_ok) in MID_CFGS.items():
                    if kind == "gated" and not gated_ok:
                        continue
                    bno = bn // 2 if kind == "gated" else bn
                    for sp in (((1, 2, 4, 8) if k >= 8192 else (1, 2, 4)) if kind == "row" else (1,)):
                        wgs = -(-m // bm) * -(-n // bno) * sp
                        if wgs > (640 if n < 65536 else 4096) or wgs < 48 or (k // 64) % sp:
                            continue
                        if kind == "col":
                            fn = lambda i: hip_ops.linear_mid_m(x, wps[i % copies], n, bias, c)       # noqa: E731
                        elif kind == "gated":
                            fn = lambda i: hip_ops.gate_up_silu_mid_m(x, wps[i % copies], n, c)       # noqa: E731
                        elif sp == 1:
                            fn = lambda i: hip_ops.add_rmsnorm(hip_ops.linear_mid_m(x, wps[i % copies], n, None, c), res, nw, 1e-6)  # noqa: E731
                        else:
                            fn = lambda i: hip_ops.add_rmsnorm_partial(hip_ops.linear_mid_m_partial(x, wps[i % copies], n, sp, c), res, nw, 1e-6)  # noqa: E731
                        cands.append((timed(fn, ncall), c, sp))
                cands.sort()
                t, c, sp = cands[0]
                plans[key] = {"cfg": c, "splits": sp, "us": round(t, 2), "lib_us": round(t_lib, 2)}
                top = "  ".join(f"c{cc}s{ss}={tt:.1f}" for tt, cc, ss in cands[:5])
                print(f"{model} {key}: lib {t_lib:.1f} us | best cfg {c} splits {sp} {t:.1f} us (x{t_lib / t:.2f}) | {top}"
                      f"  [{time.time() - t_start:.0f}s]", flush=True)
            del ws, wps
            torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "note": "tools/tune_mid.py; us = microseconds per call "
                   "(row: projection + residual-add + RMSNorm; gated: gate_up + SiluAndMul); cfg 0 = library GEMM",
                   "plans": plans}, f, indent=1, sort_keys=True)
    print(f"wrote {len(plans)} plans to {a.out}")


if __name__ == "__main__":
    main()
