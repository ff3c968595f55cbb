#!/usr/bin/env python3
"""Counterpart of the reference's examples/cpu_inference_opt.py (BASELINE config 1): OPT-125M, five short prompts,
``max_num_seqs=8, max_model_len=1024, enforce_eager=True, dtype=float32``, sampling ``temperature=0.6, top_p=0.95,
top_k=40, max_tokens=50`` (reference lines 56-62 and 143-158), through ``LLM(Config(...)).generate``.

The reference runs that workload on the host CPU; this build's product path is the MI355X, so by default the same
workload runs on the HIP kernels (fp32 tensors: the generic paged-decode / prefill kernels at head_dim 64).  ``--cpu``
prints the CPU timing of the same workload instead — that leg lives in ``bench.py`` (``cpu_baseline.config1``), the
only place outside the tests that may drive the CPU restatement under ``oracle/``.

    python examples/cpu_inference_opt.py                      # OPT-125M dimensions, random weights, token-id prompts
    python examples/cpu_inference_opt.py --model /path/to/opt-125m   # a local HF directory (tokenizer + safetensors)
    python examples/cpu_inference_opt.py --cpu
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROMPTS = ["Hello, who are you?", "What is your name?", "Where are you from?", "Where is the capital of France?",
           "Tell me a joke."]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None, help="local HF model directory; default: OPT-125M dimensions, random weights")
    ap.add_argument("--cpu", action="store_true", help="time the workload on the host CPU (bench.py --cpu-config1-only)")
    a = ap.parse_args()
    if a.cpu:
        raise SystemExit(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-config1-only"]).returncode)
    import torch

    from bench import OPT_125M
    from minivllm_amd import LLM, Config, SamplingParams

    with tempfile.TemporaryDirectory() as workdir:
        model = a.model
        if model is None:
            model = workdir
            with open(os.path.join(workdir, "config.json"), "w") as f:
                json.dump(OPT_125M, f)
        cfg = Config(model, max_num_seqs=8, max_model_len=1024, enforce_eager=True, device_memory_utilization=0.9,
                     dtype="float32", load_format="auto" if a.model else "dummy", num_kvcache_blocks=8 * 16)
        llm = LLM(cfg)
        sp = SamplingParams(temperature=0.6, top_p=0.95, top_k=40, max_tokens=50, ignore_eos=a.model is None)
        if llm.tokenizer is not None:
            prompts = PROMPTS
        else:       # no tokenizer offline: 16 token ids per prompt (what the five prompts come to under the template)
            g = torch.Generator().manual_seed(0)
            prompts = [torch.randint(3, OPT_125M["vocab_size"], (16,), generator=g).tolist() for _ in PROMPTS]
        llm.generate(prompts[:1], SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True), use_tqdm=False)  # warm-up
        t0 = time.perf_counter()
        outs = llm.generate(prompts, sp, use_tqdm=False)
        dt = time.perf_counter() - t0
        n = sum(len(o["token_ids"]) for o in outs)
        for p, o in zip(prompts, outs):
            print(f"--- {p if isinstance(p, str) else f'<{len(p)} token ids>'}\n{o['text'] or o['token_ids'][:16]}")
        print(f"{n} tokens in {dt:.3f}s = {n / dt:.1f} tokens/s (MI355X, fp32, eager)")
        llm.exit()


if __name__ == "__main__":
    main()
