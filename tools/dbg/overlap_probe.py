#!/usr/bin/env python3
"""Would two half-batches on two streams hide the projections of one under the attention of the other?

The kernel sequence of a Qwen3-1.7B decode layer (qkv -> paged attention -> o (K split) + add/norm -> gate_up+silu ->
down (K split) + add/norm; RoPE left out), 28 layers with their own weights and K/V pools, ctx 1024:
    A   one chain of B rows on one stream                                   (what the engine does)
    B   two chains of B/2 rows on two streams, captured as ONE hipGraph with a fork and a join;
        the second chain starts when the first has issued its first qkv (so that their attention phases alternate)
Timed as graph replays.  Results are not checked (the two chains share the split-KV workspace).

    python tools/dbg/overlap_probe.py [--batch 256] [--layers 28]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivllm_amd import hip_ops as H  # noqa: E402
from minivllm_amd.layers.linear import mid_plan  # noqa: E402

DEV = "cuda:0"
HID, NQ, NKV, D, INTER, BS = 2048, 16, 8, 128, 6144, 64


class Layer:
    def __init__(self, batch, ctx):
        def w(n, k):
            return H.pack_weight((torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16))
        self.wqkv, self.wo = w((NQ + 2 * NKV) * D, HID), w(HID, NQ * D)
        self.wgu, self.wd = w(2 * INTER, HID), w(HID, INTER)
        self.nw = torch.ones(HID, device=DEV, dtype=torch.bfloat16)
        nblk = ctx // BS
        self.kc = torch.randn(batch * nblk, BS, NKV, D, device=DEV, dtype=torch.bfloat16)
        self.vc = torch.randn(batch * nblk, BS, NKV, D, device=DEV, dtype=torch.bfloat16)


def chain(layers, x, res, table, ctxl):
    m = x.shape[0]
    for L in layers:
        cfg, _ = mid_plan("col", m, (NQ + 2 * NKV) * D, HID)
        qkv = H.linear_mid_m(x, L.wqkv, (NQ + 2 * NKV) * D, None, cfg)
        q = qkv[:, :NQ * D].view(m, NQ, D)
        o = H.paged_decode(q, L.kc, L.vc, table, ctxl, D ** -0.5)
        cfg, sp = mid_plan("row", m, HID, NQ * D)
        if sp > 1:
            x, res = H.add_rmsnorm_partial(H.linear_mid_m_partial(o.view(m, NQ * D), L.wo, HID, sp, cfg), res, L.nw, 1e-6)
        else:
            x, res = H.add_rmsnorm(H.linear_mid_m(o.view(m, NQ * D), L.wo, HID, None, cfg), res, L.nw, 1e-6)
        cfg, _ = mid_plan("gated", m, INTER, HID)
        act = H.gate_up_silu_mid_m(x, L.wgu, INTER, cfg)
        cfg, sp = mid_plan("row", m, HID, INTER)
        if sp > 1:
            x, res = H.add_rmsnorm_partial(H.linear_mid_m_partial(act, L.wd, HID, sp, cfg), res, L.nw, 1e-6)
        else:
            x, res = H.add_rmsnorm(H.linear_mid_m(act, L.wd, HID, None, cfg), res, L.nw, 1e-6)
    return x


def timed(graph, reps=10):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--ctx", type=int, default=1024)
    a = ap.parse_args()
    B, ctx = a.batch, a.ctx
    torch.manual_seed(0)
    layers = [Layer(B, ctx) for _ in range(a.layers)]
    nblk = ctx // BS
    table = torch.arange(B * nblk, device=DEV, dtype=torch.int32).view(B, nblk)
    ctxl = torch.full((B,), ctx, device=DEV, dtype=torch.int32)
    x = torch.randn(B, HID, device=DEV).to(torch.bfloat16)
    res = torch.randn(B, HID, device=DEV).to(torch.bfloat16)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    h = B // 2

    with torch.cuda.stream(s0):
        chain(layers, x, res.clone(), table, ctxl)             # warm-up (workspace, lazy module loads)
        chain(layers, x[:h], res[:h].clone(), table[:h], ctxl[:h])
        torch.cuda.synchronize()
        g_full = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_full, stream=s0):
            chain(layers, x, res.clone(), table, ctxl)
    t_full = timed(g_full)
    print(f"A  one chain of {B} rows:                      {t_full:7.3f} ms / step", flush=True)

    with torch.cuda.stream(s0):
        g_half = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_half, stream=s0):
            chain(layers, x[:h], res[:h].clone(), table[:h], ctxl[:h])
    t_half = timed(g_half)
    print(f"   one chain of {h} rows alone:                  {t_half:7.3f} ms / step  (two in a row: {2 * t_half:.3f})", flush=True)

    for offset in (False, True):
        with torch.cuda.stream(s0):
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=s0):
                fork = torch.cuda.Event()
                if not offset:
                    fork.record(s0)
                    s1.wait_event(fork)
                r0 = res[:h].clone()
                if offset:
                    # the second chain may start once the first has issued its first projection
                    cfg, _ = mid_plan("col", h, (NQ + 2 * NKV) * D, HID)
                    H.linear_mid_m(x[:h], layers[0].wqkv, (NQ + 2 * NKV) * D, None, cfg)
                    fork.record(s0)
                    s1.wait_event(fork)
                with torch.cuda.stream(s1):
                    chain(layers, x[h:], res[h:].clone(), table[h:], ctxl[h:])
                    join = torch.cuda.Event()
                    join.record(s1)
                chain(layers, x[:h], r0, table[:h], ctxl[:h])
                s0.wait_event(join)
        t2 = timed(g2)
        print(f"B  two chains of {h} rows on two streams{' (offset start)' if offset else '               '}: {t2:7.3f} ms / step "
              f"({t_full / t2:.2f}x of A)", flush=True)


if __name__ == "__main__":
    main()
