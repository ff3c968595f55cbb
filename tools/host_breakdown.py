#!/usr/bin/env python3
"""Where does the host spend a decode step?  (bench.py workload, perf_counter around each phase)"""
import json, os, sys, tempfile, time
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
    T = dict(schedule=0.0, build=0.0, ship_replay=0.0, sync_tolist=0.0, post=0.0)
    N = 32
    orig_replay = ex._replay_graph
    for _ in range(N):
        t0 = time.perf_counter()
        seqs, is_prefill = eng.scheduler.schedule()
        t1 = time.perf_counter()
        bs = ex._graph_bucket(len(seqs))
        inp = ex.builder.decode(seqs, pad_to=bs)
        t2 = time.perf_counter()
        b = inp.num_seqs
        ex.sampler.calls += 1
        ex.builder.p_rng[0] = ex.sampler.calls
        ex._pack_dev.copy_(ex.builder.decode_pack, non_blocking=True)
        w = inp.block_tables.shape[1]
        if w and inp.tables_dirty:
            ex._dev_tables[:b, :w].copy_(inp.block_tables, non_blocking=True)
        ex._ship_sampling(seqs, b_pad=bs, unchanged=not inp.batch_changed)
        ex.graphs[bs].replay()
        gv = ex.graph_vars
        t3 = time.perf_counter()
        toks = gv["tokens"][:b].tolist()
        t4 = time.perf_counter()
        eng.scheduler.postprocess(seqs, toks)
        t5 = time.perf_counter()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[k] += v
    print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "us per step; total", round(sum(T.values()) / N * 1e3, 3), "ms")
    eng.exit()
