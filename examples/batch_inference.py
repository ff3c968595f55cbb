#!/usr/bin/env python3
"""Batch generation through the reference-compatible API (counterpart of the reference's
examples/batch_inference_example.py): ``LLM(Config(...)).generate(prompts, SamplingParams(...))``.

    python examples/batch_inference.py /path/to/Qwen3-1.7B --max-tokens 64
    python examples/batch_inference.py examples/qwen3-0.6b --dummy         # random weights (config.json only), token-id prompts
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivllm_amd import LLM, Config, SamplingParams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", help="local HF model directory (config.json [+ tokenizer + safetensors])")
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--temperature", type=float, default=0.0, help="0 = greedy (reference: batch_inference_example.py:37)")
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--top-k", type=int, default=-1)
    ap.add_argument("--dummy", action="store_true", help="random-initialise the weights (no checkpoint needed)")
    ap.add_argument("--eager", action="store_true")
    a = ap.parse_args()
    cfg = Config(a.model, dtype="bfloat16", tensor_parallel_size=a.tp, enforce_eager=a.eager, max_num_seqs=256,
                 load_format="dummy" if a.dummy else "auto")
    llm = LLM(cfg)
    sp = SamplingParams(temperature=a.temperature, top_p=a.top_p, top_k=a.top_k, max_tokens=a.max_tokens,
                        ignore_eos=a.dummy)
    if llm.tokenizer is not None:
        prompts = ["Introduce yourself.", "List all prime numbers within 100.", "What is the capital of France?",
                   "Explain paged attention in two sentences."]
    else:
        prompts = [[(7 * i + j) % 1000 + 10 for j in range(16 + 8 * i)] for i in range(4)]
    t0 = time.perf_counter()
    outs = llm.generate(prompts, sp)
    dt = time.perf_counter() - t0
    n = sum(len(o["token_ids"]) for o in outs)
    for p, o in zip(prompts, outs):
        print(f"--- {p if isinstance(p, str) else f'<{len(p)} token ids>'}\n{o['text'] or o['token_ids'][:16]}")
    print(f"{n} tokens in {dt:.2f}s ({n / dt:.1f} tok/s end to end)")
    llm.exit()


if __name__ == "__main__":
    main()
