#!/usr/bin/env python3
"""Prefill tokens/s of the whole model step (SURVEY §8 d(ii), configs 3/4): batches of n prompts x len tokens
through ``InferenceExecutor.execute_batch(prefill=True)`` -- input build, H2D, every layer (hipBLASLt GEMMs,
fused norm/rope/store kernels, the MFMA flash prefill), LM head on the last tokens and the sampler.
Random-init weights at the true dimensions, fresh random prompts per repetition (no prefix-cache hits),
median of --reps timed runs after one warm-up per shape.

    python tools/bench_prefill.py --model qwen2-7b [--cases 16x1024,4x4096,64x256]
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODELS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2-7b", choices=list(MODELS))
    ap.add_argument("--cases", default="16x1024,4x4096,64x256")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from minivllm_amd import LLM, Config
    from minivllm_amd.engine.sequence import Sequence

    cases = [tuple(int(v) for v in c.split("x")) for c in a.cases.split(",")]
    with tempfile.TemporaryDirectory() as workdir:
        with open(os.path.join(workdir, "config.json"), "w") as f:
            json.dump(MODELS[a.model], f)
        cfg = Config(workdir, dtype="bfloat16", kvcache_block_size=64, max_num_seqs=max(n for n, _ in cases),
                     max_model_len=4096, max_num_batched_tokens=16384, enforce_eager=True, load_format="dummy",
                     seed=0, num_kvcache_blocks=16384 // 64 + 2 * max(n for n, _ in cases))
        engine = LLM(cfg)
        ex, bm = engine.model_runner.inference_executor, engine.scheduler.block_manager
        vocab = MODELS[a.model]["vocab_size"]
        g = torch.Generator().manual_seed(0)
        hf = MODELS[a.model]
        L, H, D = hf["num_hidden_layers"], hf["num_attention_heads"], hf.get("head_dim") or hf["hidden_size"] // hf["num_attention_heads"]
        print(f"| model | prompts x len | ms / batch | prefill tokens/s | attention TFLOP/s share (4·H·D·S²/2 per seq and layer) |")
        print("|---|---|---|---|---|")
        for n, ln in cases:
            times = []
            for rep in range(a.reps + 1):
                seqs = [Sequence(torch.randint(0, vocab, (ln,), generator=g).tolist(), block_size=64) for _ in range(n)]
                for s in seqs:
                    bm.allocate(s)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ex.execute_batch(seqs, prefill=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                for s in seqs:
                    bm.deallocate(s)
                if rep:
                    times.append(dt)
            med = statistics.median(times)
            attn_flop = L * n * 4 * H * D * (ln * ln - ln * (ln - 1) / 2)
            print(f"| {a.model} | {n} x {ln} | {med * 1e3:.1f} | {n * ln / med:,.0f} | {attn_flop / 1e12:.2f} TFLOP per batch |",
                  flush=True)
        engine.exit()


if __name__ == "__main__":
    main()
