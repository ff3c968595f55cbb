#!/usr/bin/env python3
"""Does the paged-decode kernel's speed depend on the STRIDES of the KV pool (round-2 verdict item 5)?

The engine's pool is one tensor [2][L][num_blocks][64][Hkv][D]: the K and V rows of one block are a plane stride
(L * num_blocks * block_bytes) apart, successive layers a layer stride (num_blocks * block_bytes) apart.  `--auto-kv`
caps the pool at max_num_seqs * ceil(max_model_len / 64) = 16384 blocks: with 128-KiB blocks (Qwen3-1.7B) that makes the
layer stride 2 GiB and the plane stride 56 GiB — large powers of two — while the explicit pool of bench.py (4608-4864
blocks) has odd strides.  Same blocks in use either way.  This probe times the 28 per-layer launches of one decode
step (B=256, ctx~1040, the blocks a fresh engine hands out: sequence i owns blocks 17i .. 17i+16) on pools carved from
ONE flat allocation with a chosen block count per layer and a chosen extra offset of the V plane.

    python tools/kv_stride_probe.py [--hkv 8 --heads 16] [--cases 4864,16384,16383,16400,16384+4096,...]
    case syntax: <blocks per layer>[+<extra V-plane offset in bytes>][/<extra layer pad in bytes>][@<base offset of the pool in GiB>]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from minivllm_amd import hip_ops as H  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=1040)
    ap.add_argument("--cases", default="4864,16384,16383,16400,16384+4096,16384+65536,16384/4096,16448,17408")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    D, bs, L, B = 128, 64, a.layers, a.batch
    blk_elems = bs * a.hkv * D
    per_seq = -(-(a.ctx + 64) // bs)
    used = B * per_seq
    cases = []
    for c in a.cases.split(","):
        lpad = 0
        base = 0
        if "@" in c:
            c, b_ = c.split("@")
            base = int(float(b_) * (1 << 30))
        if "/" in c:
            c, lp = c.split("/")
            lpad = int(lp)
        voff = 0
        if "+" in c:
            c, vo = c.split("+")
            voff = int(vo)
        cases.append((int(c), voff, lpad, base))
    biggest = max((nb * blk_elems * 2 + lp) * L * 2 + vo + ba for nb, vo, lp, ba in cases) + (1 << 20)
    flat = torch.empty(biggest // 2, dtype=torch.bfloat16, device=DEV)
    # fill what the kernel reads with data once (values do not matter for speed; zeros would raise the clock)
    q = torch.randn(B, a.heads, D, device=DEV).bfloat16()
    table = (torch.arange(B, dtype=torch.int32, device=DEV).view(B, 1) * per_seq
             + torch.arange(per_seq, dtype=torch.int32, device=DEV).view(1, per_seq)).contiguous()
    cl = torch.full((B,), a.ctx, dtype=torch.int32, device=DEV)
    out = torch.empty_like(q)
    print(torch.cuda.get_device_name(0), f"B={B} ctx={a.ctx} heads {a.heads}/{a.hkv} L={L}, blocks in use {used}")
    algo = B * a.ctx * 2 * a.hkv * D * 2 + 2 * B * a.heads * D * 2 + B * per_seq * 4

    def views(nb, voff, lpad, base):
        layer_elems = nb * blk_elems + lpad // 2
        plane_elems = L * layer_elems + voff // 2
        ks, vs = [], []
        for i in range(L):
            ko = base // 2 + i * layer_elems
            vo = base // 2 + plane_elems + i * layer_elems
            ks.append(flat[ko:ko + nb * blk_elems].view(nb, bs, a.hkv, D))
            vs.append(flat[vo:vo + nb * blk_elems].view(nb, bs, a.hkv, D))
        return ks, vs

    for i in range(0, flat.numel(), 1 << 30):            # random data everywhere (zeros would raise the clock)
        flat[i:i + (1 << 30)].normal_()
    results = {c: [] for c in cases}
    for rep in range(a.reps + 1):
        for c in cases:
            ks, vs = views(*c)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(L):
                H.paged_decode(q, ks[i], vs[i], table, cl, D ** -0.5, out=out)
            e.record()
            torch.cuda.synchronize()
            if rep:
                results[c].append(s.elapsed_time(e) / L * 1e3)
    for (nb, voff, lpad, base), t in results.items():
        t = sorted(t)
        med = t[len(t) // 2]
        lay = nb * blk_elems * 2 + lpad
        print(f"blocks/layer {nb:6d} V-plane +{voff:6d} B layer pad {lpad:5d} B base {base / 2**30:5.1f} GiB | layer stride {lay / 2**20:9.3f} MiB "
              f"plane stride {(L * lay + voff) / 2**30:8.4f} GiB | {med:7.1f} us/launch (min {t[0]:.1f}) = {algo / med / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
