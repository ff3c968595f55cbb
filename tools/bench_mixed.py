#!/usr/bin/env python3
"""What mixed prefill/decode batches buy (Config.enable_mixed_batches): a serving trace on one engine.

`--running` sequences are decoding (context `--ctx`) while `--arrivals` prompts of `--prompt` tokens arrive, one every
`--every` engine steps.  Reported per schedule (separate steps = the reference's: a prefill step stalls every decoder;
mixed = decode rows ride in the prefill steps; with `--chunk` also chunked prefill of the prompts):
  * wall time until everything has finished, decode tokens/s of the whole trace,
  * the inter-token gaps the RUNNING sequences see: median and worst (a prefill step between two of their tokens is a stall).
Random-init weights at the model's true dimensions (throughput is value-independent), greedy, token-id prompts.

    python tools/bench_mixed.py [--model qwen3-1.7b] [--running 64] [--arrivals 16] [--prompt 2048]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(a, mixed, chunk):
    from minivllm_amd import LLM, Config, SamplingParams

    cfgm = bench.MODELS[a.model]
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(cfgm, f)
        total = a.running + a.arrivals
        max_len = max(a.ctx, a.prompt) + a.gen + 64
        blocks = total * (-(-max_len // 64) + 1)
        cfg = Config(d, dtype="bfloat16", kvcache_block_size=64, max_num_seqs=total, max_model_len=max_len,
                     max_num_batched_tokens=chunk if chunk else max(16384, max_len), load_format="dummy", seed=0,
                     num_kvcache_blocks=blocks, enable_chunked_prefill=bool(chunk), enable_mixed_batches=mixed)
        eng = LLM(cfg)
        g = torch.Generator().manual_seed(0)
        sp_run = SamplingParams(temperature=0.0, max_tokens=a.gen, ignore_eos=True)
        sp_new = SamplingParams(temperature=0.0, max_tokens=a.gen // 4, ignore_eos=True)
        first = [eng.add_request(torch.randint(0, cfgm["vocab_size"], (a.ctx,), generator=g).tolist(), sp_run)
                 for _ in range(a.running)]
        # the running set is prefilled before the clock starts, the same way for every schedule (whole prompts,
        # nothing decoding alongside): what is compared is the trace behind it
        sch = eng.scheduler
        keep = (sch.mixed_batches, sch.max_num_batched_tokens)
        sch.mixed_batches, sch.max_num_batched_tokens = False, max(16384, a.ctx)
        while sch.waiting:
            eng.step()
        sch.mixed_batches, sch.max_num_batched_tokens = keep
        watch = {s.seq_id: s for s in eng.scheduler.running}
        last_len = {i: len(s) for i, s in watch.items()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last_t = {i: t0 for i in watch}
        gaps, step, arrived, decode_tokens = [], 0, 0, 0
        while not eng.is_finished() or arrived < a.arrivals:
            if arrived < a.arrivals and step % a.every == 0:
                eng.add_request(torch.randint(0, cfgm["vocab_size"], (a.prompt,), generator=g).tolist(), sp_new)
                arrived += 1
            if eng.is_finished():
                step += 1
                continue
            eng.step()
            torch.cuda.synchronize()
            now = time.perf_counter()
            for i, s in watch.items():
                if len(s) > last_len[i]:
                    gaps.append(now - last_t[i])
                    decode_tokens += len(s) - last_len[i]
                    last_len[i], last_t[i] = len(s), now
            step += 1
        wall = time.perf_counter() - t0
        eng.exit()
    gaps.sort()
    return {"schedule": ("mixed" if mixed else "separate") + (f" + {chunk}-token chunks" if chunk else ""),
            "wall_s": round(wall, 3), "engine_steps": step,
            "running_tokens_per_s": round(decode_tokens / wall, 1),
            "gap_ms_median": round(gaps[len(gaps) // 2] * 1e3, 2), "gap_ms_p99": round(gaps[int(len(gaps) * 0.99)] * 1e3, 2),
            "gap_ms_max": round(gaps[-1] * 1e3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-1.7b")
    ap.add_argument("--running", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--gen", type=int, default=128)
    ap.add_argument("--arrivals", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=4096)
    ap.add_argument("--every", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=1024)
    a = ap.parse_args()
    print(f"# {a.model}: {a.running} sequences decoding (ctx {a.ctx}, {a.gen} tokens each), {a.arrivals} prompts of {a.prompt} "
          f"tokens arriving every {a.every} steps")
    for mixed, chunk in ((False, 0), (True, 0), (False, a.chunk), (True, a.chunk)):
        print(json.dumps(run(a, mixed, chunk)), flush=True)


if __name__ == "__main__":
    main()
