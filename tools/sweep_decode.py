#!/usr/bin/env python3
"""Decode sweep of SURVEY §8(d) config 3: Qwen3-1.7B bf16, hipGraph decode, B x ctx grid, greedy.
One engine (max_num_seqs=256); KV content is synthetic (throughput is value-independent), the
scheduler / block manager / graph path are the real ones.  Prints a markdown table."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-1.7b")
    ap.add_argument("--batches", default="1,8,32,64,128,256")
    ap.add_argument("--ctxs", default="128,1024,4096")
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    from minivllm_amd import SamplingParams
    args = argparse.Namespace(model=a.model, batch=256, ctx=4096, steps=a.steps, warmup=4, gpus=1, eager=False, auto_kv=False)
    cfgm = bench.MODELS[a.model]
    with tempfile.TemporaryDirectory() as d:
        eng = bench.build_engine(args, d)
        ex = eng.model_runner.inference_executor
        ex.kv_cache.normal_(0.0, 1.0)
        g = torch.Generator().manual_seed(0)
        L, nb, bs, hkv, dd = ex.kv_cache.shape[1:]
        w_bytes = sum(p.numel() * p.element_size() for n, p in ex.model.named_parameters() if "embed_tokens" not in n or cfgm["tie_word_embeddings"])
        print(f"| B | ctx | ms/step | decode tok/s | HBM-bound tok/s @8.0 TB/s spec | frac of 8.0 TB/s (of the 6.29 TB/s copy ceiling) |")
        print("|---|---|---|---|---|---|")
        for B in [int(x) for x in a.batches.split(",")]:
            for ctx in [int(x) for x in a.ctxs.split(",")]:
                sp = SamplingParams(temperature=0.0, max_tokens=a.steps + 16, ignore_eos=True)
                for _ in range(B):
                    eng.add_request(torch.randint(0, cfgm["vocab_size"], (ctx,), generator=g).tolist(), sp)
                while eng.scheduler.waiting:
                    batch, is_prefill = eng.scheduler.schedule()
                    eng.scheduler.postprocess(batch, [1] * len(batch))
                for _ in range(4):
                    eng.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    eng.step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps
                kv_bytes = B * (ctx + 4 + a.steps / 2) * 2 * hkv * dd * 2 * L
                bound = B / ((w_bytes + kv_bytes) / 8.0e12)
                print(f"| {B} | {ctx} | {dt*1e3:.3f} | {B/dt:.0f} | {bound:.0f} | {B/dt/bound:.2f} ({B/dt/bound*8.0/6.29:.2f}) |", flush=True)
                for s in list(eng.scheduler.running):
                    eng.scheduler.block_manager.deallocate(s)
                eng.scheduler.running.clear()
                ex._spec = None
                ex.builder._last_batch = None
        eng.exit()


if __name__ == "__main__":
    main()
