#!/usr/bin/env python3
"""Where does the host spend a decode step?  (bench.py workload; perf_counter around the three engine phases)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse()
with tempfile.TemporaryDirectory() as d:
    eng = bench.build_engine(args, d)
    args.synthetic_kv = True
    bench.fill_prompts(eng, args)
    ex = eng.model_runner.inference_executor
    for _ in range(4):
        eng.step()
    torch.cuda.synchronize()
    T = dict(schedule=0.0, run=0.0, post=0.0)
    N = 32
    for _ in range(N):
        t0 = time.perf_counter()
        seqs, is_prefill = eng.scheduler.schedule()
        ex.speculate_next = not eng.scheduler.waiting
        t1 = time.perf_counter()
        toks = eng.model_runner.call("run", seqs, is_prefill)
        t2 = time.perf_counter()
        eng.scheduler.postprocess(seqs, toks)
        t3 = time.perf_counter()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2)):
            T[k] += v
    print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "us per step; total", round(sum(T.values()) / N * 1e3, 3),
          "ms; speculation hits/misses", ex.spec_hits, ex.spec_misses)
    eng.exit()
