#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s of the paged-attention inference path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: one rank per GPU.  Started WITHOUT torch.distributed.run, the script starts it itself as a child
    process — before anything has touched the GPU — relays the child's output and exits with its code.)

Workload (BASELINE.json configs[2]): Qwen3-1.7B, bf16, kvcache_block_size=64,
hipGraph decode, max_num_seqs=256 — B=256 sequences at context 1024, synthetic token ids,
random-initialised weights at the true architecture dimensions (no checkpoints offline; throughput is
value-independent).  One "step" = one engine decode step over the whole batch: scheduler ->
input prep -> H2D -> graph replay (28 layers + LM head + sampler) -> D2H of the B tokens ->
postprocess.  value = B * K / wall time (max over ranks).  With N>1 the same model runs
tensor-parallel over N GPUs (RCCL all-reduce over xGMI): total work fixed => "strong" scaling.

Also reported on the same line:
  roofline     — the dominant kernel (paged-KV decode attention): algorithmic bytes per launch /
                 average launch time, timed with HIP events on the launching stream on the LIVE
                 block tables / context lengths / KV cache of this run
  cpu_baseline — the CPU restatement of the reference's eager path (oracle/, "port") timed on this
                 host's cores on a bounded sample of the same model (one untimed warm-up step, then the
                 median of a fixed number of steps), plus BASELINE config 1 (OPT-125M fp32, the preset of
                 the reference's examples/cpu_inference_opt.py) as `config1`
  secondary    — (N=1) the rest of BASELINE.json's metric under the same clock: prefill tokens/s on the
                 config-4 prompt shapes for the headline model, and Qwen2-7B decode (B=256, ctx 1024, its own
                 roofline object) + prefill tokens/s, each with its step count and milliseconds
  allreduce_probe — (N>1) HIP-event time of the tensor-parallel all-reduce launches of eager probe steps, per rank
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODELS = {
    # public HF config.json values
    "qwen3-1.7b": dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=2048,
                       num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                       intermediate_size=6144, vocab_size=151936, max_position_embeddings=40960,
                       rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True, hidden_act="silu",
                       attention_bias=False, torch_dtype="bfloat16", eos_token_id=151645),
    "qwen3-0.6b": dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=1024,
                       num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                       intermediate_size=3072, vocab_size=151936, max_position_embeddings=40960,
                       rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True, hidden_act="silu",
                       attention_bias=False, torch_dtype="bfloat16", eos_token_id=151645),
    "qwen2-7b": dict(model_type="qwen2", architectures=["Qwen2ForCausalLM"], hidden_size=3584,
                     num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                     intermediate_size=18944, vocab_size=152064, max_position_embeddings=32768,
                     rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False, hidden_act="silu",
                     torch_dtype="bfloat16", eos_token_id=151645),
    "qwen2-72b": dict(model_type="qwen2", architectures=["Qwen2ForCausalLM"], hidden_size=8192,
                      num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8,
                      intermediate_size=29568, vocab_size=152064, max_position_embeddings=32768,
                      rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False, hidden_act="silu",
                      torch_dtype="bfloat16", eos_token_id=151645),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="qwen3-1.7b", choices=list(MODELS))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--eager", action="store_true", help="enforce_eager (no hipGraph)")
    ap.add_argument("--synthetic-kv", action="store_true",
                    help="fill the KV cache with random values instead of running the prompt prefill")
    ap.add_argument("--auto-kv", action="store_true",
                    help="size the KV pool from free HBM (num_kvcache_blocks=-1, the way a user runs it) instead of "
                         "allocating exactly what this workload needs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary records (Qwen2-7B decode, prefill tokens/s on the config-4 shapes)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-baseline steps (median), after one warm-up")
    ap.add_argument("--cpu-config1-only", action="store_true",
                    help="run only the CPU leg of BASELINE config 1 (OPT-125M fp32, reference preset) and print it")
    return ap.parse_args(argv)


def launch_command(args, script=None, port=None):
    """The child a bare ``python bench.py --gpus N`` (N > 1) starts: torch.distributed.run, one rank per GPU, the
    same arguments.  Separate from main() so that a host test can check it."""
    if port is None:
        import socket

        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--model", args.model, "--batch", str(args.batch), "--ctx", str(args.ctx)]
    for flag in ("eager", "synthetic_kv", "auto_kv", "no_cpu_baseline", "no_secondary"):
        if getattr(args, flag):
            cmd.append("--" + flag.replace("_", "-"))
    return cmd


def build_engine(args, workdir, model=None):
    from minivllm_amd import LLM, Config

    model = model or args.model
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "config.json"), "w") as f:
        json.dump(MODELS[model], f)
    max_len = max(4096, args.ctx + args.steps + args.warmup + 64)
    cfg = Config(workdir, dtype="bfloat16", kvcache_block_size=64, max_num_seqs=args.batch,
                 max_model_len=max_len, max_num_batched_tokens=max(16384, max_len),
                 tensor_parallel_size=args.gpus, enforce_eager=args.eager, load_format="dummy", seed=0,
                 num_kvcache_blocks=-1 if args.auto_kv else
                 args.batch * (-(-(args.ctx + args.steps + args.warmup + 8) // 64) + 1))
    return LLM(cfg)


def fill_prompts(engine, args, model=None):
    """B requests of ctx synthetic tokens each -> running sequences with allocated blocks."""
    from minivllm_amd import SamplingParams

    g = torch.Generator().manual_seed(0)
    vocab = MODELS[model or args.model]["vocab_size"]
    sp = SamplingParams(temperature=0.0, max_tokens=args.steps + args.warmup + 16, ignore_eos=True)
    for _ in range(args.batch):
        engine.add_request(torch.randint(0, vocab, (args.ctx,), generator=g).tolist(), sp)
    prefill_tokens, t0 = 0, time.perf_counter()
    if args.synthetic_kv:
        ex = engine.model_runner.inference_executor
        ex.kv_cache.normal_(0.0, 1.0)
        while engine.scheduler.waiting:
            batch, is_prefill = engine.scheduler.schedule()
            assert is_prefill
            engine.scheduler.postprocess(batch, [1] * len(batch))
        return None
    while engine.scheduler.waiting:
        _, n = engine.step()
        prefill_tokens += max(n, 0)
    torch.cuda.synchronize()
    return prefill_tokens / (time.perf_counter() - t0)


class _EventProbe:
    """Brackets calls with HIP event pairs recorded on the stream the call launches on (torch's current stream)."""

    def __init__(self):
        self.pairs: dict[str, list] = {}

    def wrap(self, name, real, keep=lambda out: True):
        def timed(*a, **kw):
            stream = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            out = real(*a, **kw)
            e1.record(stream)
            if keep(out):
                self.pairs.setdefault(name, []).append((e0, e1))
            return out
        return timed

    def times_us(self, name):
        return [a.elapsed_time(b) * 1e3 for a, b in self.pairs.get(name, [])]


def decode_roofline(engine, args, model=None):
    """Average duration of the paged-decode launches of REAL decode steps of this run, measured with
    HIP events on the launching stream: the executor is switched to eager for a few extra steps and
    every mv_paged_decode launch (one per layer, in situ between the GEMMs, on the live block
    tables / context lengths / KV cache) is bracketed by an event pair; the host runs ahead of the GPU (see below).
    With N > 1 the same probe steps also bracket every tensor-parallel all-reduce launch (-> allreduce_probe)."""
    import torch.distributed as dist

    from minivllm_amd import hip_ops, parallel_state
    from minivllm_amd.layers import attention as attn_mod

    model = model or args.model
    ex = engine.model_runner.inference_executor
    seqs = list(engine.scheduler.running)
    lens = [len(s) for s in seqs]
    L, nb, bs, hkv, d = ex.kv_cache.shape[1:]
    heads = MODELS[model]["num_attention_heads"] // args.gpus
    probe = _EventProbe()
    # the launch is mv_paged_decode, or mv_rope_paged_decode where the layer folds RoPE + the K/V store into it
    reals = {"paged_decode": hip_ops.paged_decode, "rope_paged_decode": hip_ops.rope_paged_decode}
    ar = parallel_state.get_custom_all_reduce() if args.gpus > 1 else None
    ar_reals = {}
    if ar is not None:
        ar_reals = {"all_reduce": ar.all_reduce, "all_reduce_add_rmsnorm": ar.all_reduce_add_rmsnorm}
    real_dist_ar = dist.all_reduce
    was_eager = ex.enforce_eager
    ex.enforce_eager = True
    # (Events cannot be recorded inside a captured graph on ROCm — "External events are disallowed in rocm",
    # tools/dbg/ext_event_probe.py — so the probe steps are eager.  Keeping the host ahead of the GPU with a spin
    # kernel in front of each probe step changes nothing (180 vs 178 us): what an event pair adds to the ~168 us
    # rocprofv3 sees for this kernel inside the graph is the two event packets themselves, reported below as
    # `event_pair_floor_us` = the same pair around a one-element fill.)
    probe_steps = 3
    try:
        engine.step()                       # eager warm-up step (kernels already resident)
        for name, real in reals.items():
            # (None: the fused form declined the layout, the plain launch follows)
            setattr(attn_mod.hip_ops, name, probe.wrap("decode", real, keep=lambda out: out is not None))
        for name, real in ar_reals.items():
            setattr(ar, name, probe.wrap("p2p_" + name, real))
        if args.gpus > 1:
            dist.all_reduce = probe.wrap("backend_all_reduce", real_dist_ar)
        for _ in range(probe_steps):
            engine.step()
    finally:
        for name, real in reals.items():
            setattr(attn_mod.hip_ops, name, real)
        for name in ar_reals:
            delattr(ar, name)               # the instance attribute shadowed the method
        dist.all_reduce = real_dist_ar
        ex.enforce_eager = was_eager
    torch.cuda.synchronize()
    times = [t * 1e-3 for t in probe.times_us("decode")]      # ms
    assert len(times) == probe_steps * L
    # what the event pair itself costs: the same bracket around a one-element fill, inside a stream that is busy
    tiny, floor = torch.zeros(1, device=ex.device), []
    busy = torch.empty(256 << 20, dtype=torch.uint8, device=ex.device)
    for _ in range(20):
        busy.zero_()                        # ~50 us of work ahead of the pair: the host is not what it times
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tiny.zero_()
        e1.record()
        floor.append((e0, e1))
    torch.cuda.synchronize()
    del busy
    floor_us = sorted(a.elapsed_time(b) for a, b in floor)[len(floor) // 2] * 1e3
    # an event pair also spans whatever the HOST does between recording the first event and enqueuing the
    # kernel while the GPU is idle (eager probe steps are launch-bound): a rare host stall there (seen: one
    # ~170 ms gap in 84 launches) is not kernel time.  Launches above 3x the median are left out and counted.
    med = sorted(times)[len(times) // 2]
    kept = [t for t in times if t <= 3 * med]
    sec = sum(kept) * 1e-3 / len(kept)
    esz = ex.dtype.itemsize
    lens = [n + 1 + probe_steps // 2 for n in lens]   # context length at the middle probe step
    algo = sum(lens) * 2 * hkv * d * esz + 2 * len(seqs) * heads * d * esz + sum(-(-n // bs) for n in lens) * 4
    gbs = algo / sec / 1e9
    # HBM traffic from the PMC counters (separate rocprofv3 --pmc passes, corrected as the microarch guide
    # prescribes): measured on the same kernel / head layout at B=256, ctx=1024 and scaled by bytes
    traffic = src = None
    for pmc in ("r03_pmc_decode.json", "r02_pmc_decode.json"):
        path = os.path.join(ROOT, "profiles", pmc)
        if os.path.exists(path) and args.gpus == 1 and traffic is None:
            with open(path) as f:
                for name, rec in json.load(f)["kernels"].items():
                    if model in name:
                        traffic, src = int(algo * rec["ratio"]), pmc
    cfgm = MODELS[model]
    kernel = ("paged_decode_mfma_kernel" if cfgm["num_attention_heads"] // cfgm["num_key_value_heads"] >= 4
              else "paged_decode_kernel")
    roof = {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in separate passes on this kernel and head layout at B=256 ctx=1024, per launch, scaled to this run's context; counters cannot be read inside a timed run)" if traffic else None,
            "bytes_per_launch": algo, "us_per_launch": round(sec * 1e6, 2), "event_pair_floor_us": round(floor_us, 2),
            "launches_timed": len(kept),
            "launches_discarded_host_stall": len(times) - len(kept)}
    ar_probe = None
    if args.gpus > 1:
        ar_probe = {"probe_steps": probe_steps, "event_pair_floor_us": round(floor_us, 2)}
        for name in ("p2p_all_reduce", "p2p_all_reduce_add_rmsnorm", "backend_all_reduce"):
            t = probe.times_us(name)
            if t:
                m = statistics.median(t)
                keep = [v for v in t if v <= 3 * m]
                ar_probe[name] = {"launches_per_step": round(len(t) / probe_steps, 1), "median_us": round(m, 2),
                                  "us_per_step": round(sum(keep) / len(keep) * len(t) / probe_steps, 1)}
    return roof, ar_probe


def release_all(engine):
    """Drop every queued / running sequence and give its blocks back (between two records on one engine)."""
    sch = engine.scheduler
    for seq in list(sch.running) + ([sch.prefilling] if sch.prefilling is not None else []):
        sch.block_manager.deallocate(seq)
    sch.running.clear()
    sch.waiting.clear()
    sch.prefilling = None
    ex = engine.model_runner.inference_executor
    ex.speculate_next = False
    drop = getattr(ex, "_drop_speculation", None)
    if drop is not None:
        drop()
    torch.cuda.synchronize()


def prefill_records(engine, model, cases=((16, 1024), (4, 4096)), reps=5):
    """Prefill tokens/s of the whole model step (SURVEY §8 d(ii), config-4 prompt shapes): n prompts x len fresh
    random tokens through ``execute_batch(prefill=True)`` -- input build, H2D, every layer, LM head on the last
    tokens, sampler, D2H of the n tokens.  One warm-up per shape, then the median of ``reps`` timed batches."""
    from minivllm_amd.engine.sequence import Sequence

    ex, bm = engine.model_runner.inference_executor, engine.scheduler.block_manager
    vocab = MODELS[model]["vocab_size"]
    g = torch.Generator().manual_seed(1)
    out = []
    for n, ln in cases:
        times = []
        for rep in range(reps + 1):
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
        out.append({"metric": "prefill tokens/s", "model": model, "prompts": n, "prompt_len": ln,
                    "batches_timed": reps, "warmup_batches": 1, "ms_per_batch": round(med * 1e3, 3),
                    "value": round(n * ln / med, 1), "unit": "tokens/s",
                    "timing": "wall clock around execute_batch(prefill=True) bracketed by torch.cuda.synchronize(), median"})
    return out


def timed_decode(engine, args, dist=None):
    """W untimed + exactly K timed engine decode steps, barrier + synchronize on both sides, max over ranks."""
    def barrier():
        torch.cuda.synchronize()
        if args.gpus > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        engine.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.gpus > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert len(engine.scheduler.running) == args.batch, "a sequence finished inside the timed region"
    return elapsed


def usable_cores() -> int:
    """CPU threads this process may really use: the affinity mask, cut by a cgroup CPU quota where one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_threads() -> int:
    # torch's default is one thread per physical core of the HOST; a GPU box hands a job a share of the host
    # (16 CPUs for one GPU on this pool), and 128 threads time-sliced on that share is what made this baseline
    # swing 3.4x between two boxes in round 2
    n = max(1, min(torch.get_num_threads(), usable_cores()))
    torch.set_num_threads(n)
    return n


def cpu_baseline(args):
    """Oracle (CPU restatement of the reference's eager path) decode tokens/s, bounded sample: one untimed warm-up
    step, then ``--cpu-steps`` timed steps, median."""
    from oracle.model import OracleModel

    threads = _cpu_threads()
    cfg = MODELS[args.model]
    b, ctx = 8, 128
    nblk = -(-(ctx + 64) // 64)
    t_build = time.perf_counter()
    model = OracleModel.random(cfg, dtype=torch.float32, num_blocks=b * nblk, block_size=64)
    for kc, vc in zip(model.k_cache, model.v_cache):
        kc.normal_()
        vc.normal_()
    tables = torch.arange(b * nblk, dtype=torch.int32).view(b, nblk)
    lens = torch.full((b,), ctx, dtype=torch.int32)
    ids = torch.randint(0, cfg["vocab_size"], (b,))
    build_s = time.perf_counter() - t_build
    per_step = []
    for step in range(args.cpu_steps + 1):
        t0 = time.perf_counter()
        pos = (lens - 1).long()
        slots = (tables[torch.arange(b), (pos // 64)] * 64 + (pos % 64).int()).int()
        meta = dict(is_prefill=False, slot_mapping=slots, block_tables=tables, context_lens=lens)
        logits = model.forward_logits(ids, pos, meta)
        ids = logits.float().argmax(-1)
        lens = lens + 1
        if step:                             # step 0 is the warm-up (first touch of 6.9 GB of fp32 weights)
            per_step.append(time.perf_counter() - t0)
    med = statistics.median(per_step)
    return {"value": round(b / med, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{args.model} fp32 (the reference's CPU dtype, examples/cpu_inference_opt.py:143-151) random weights, "
                      f"decode only, B={b}, ctx {ctx}->{ctx + len(per_step) + 1}: 1 untimed warm-up step, then the median of "
                      f"{len(per_step)} steps ({min(per_step):.2f}-{max(per_step):.2f} s per step; oracle/model.py; build "
                      f"{build_s:.0f}s).  A bounded sample of the headline model, NOT the headline batch: the CPU path "
                      f"needs minutes per step at B={args.batch}, ctx {args.ctx}; tokens/s per sequence at B={b} is the "
                      f"CPU's best case (weights are re-read per step whatever B is).  cores = torch intra-op threads = "
                      f"min(torch's default, the CPUs this job may use: {usable_cores()}) of {os.cpu_count()} logical CPUs"}


OPT_125M = dict(model_type="opt", architectures=["OPTForCausalLM"], hidden_size=768, num_hidden_layers=12,
                num_attention_heads=12, ffn_dim=3072, vocab_size=50272, max_position_embeddings=2048,
                word_embed_proj_dim=768, do_layer_norm_before=True, activation_function="relu",
                enable_bias=True, layer_norm_elementwise_affine=True, tie_word_embeddings=True,
                torch_dtype="float16", eos_token_id=2, bos_token_id=2, pad_token_id=1)


def cpu_config1(reps=3):
    """BASELINE config 1 on this host's CPU cores: the workload of the reference's examples/cpu_inference_opt.py
    (:56-62 five prompts, :143-158 max_num_seqs=8, max_model_len=1024, eager, fp32, temperature 0.6 / top_p 0.95 /
    top_k 40 / 50 new tokens) at the TRUE OPT-125M dimensions with random weights (no checkpoint offline), token-id
    prompts of 16 tokens (what those prompts come to under the chat template), through the build's host engine
    (scheduler, block manager, input builder) with the CPU restatement of the eager model as its runner."""
    from minivllm_amd import Config, SamplingParams
    from minivllm_amd.engine.input_builder import InputBuilder
    from minivllm_amd.engine.llm_engine import LLMEngine
    from oracle.model import OracleModel, OracleRunner

    threads = _cpu_threads()
    with tempfile.TemporaryDirectory() as workdir:
        with open(os.path.join(workdir, "config.json"), "w") as f:
            json.dump(OPT_125M, f)
        cfg = Config(workdir, dtype="float32", max_num_seqs=8, max_model_len=1024, max_num_batched_tokens=1024,
                     enforce_eager=True, kvcache_block_size=64, num_kvcache_blocks=8 * 16, load_format="dummy", seed=0)
        model = OracleModel.random(OPT_125M, dtype=torch.float32, num_blocks=cfg.num_kvcache_blocks, block_size=64)
        g = torch.Generator().manual_seed(0)
        prompts = [torch.randint(3, OPT_125M["vocab_size"], (16,), generator=g).tolist() for _ in range(5)]
        sp = SamplingParams(temperature=0.6, top_p=0.95, top_k=40, max_tokens=50, ignore_eos=True)
        runs = []
        for rep in range(reps + 1):
            engine = LLMEngine(cfg, model_runner=OracleRunner(model, InputBuilder(64, 8, 1024, cfg.max_num_batched_tokens), seed=rep))
            t0 = time.perf_counter()
            outs = engine._generate_local(prompts, sp, use_tqdm=False)
            dt = time.perf_counter() - t0
            n = sum(len(o["token_ids"]) for o in outs)
            assert n == 5 * 50
            if rep:
                runs.append(n / dt)
    return {"value": round(statistics.median(runs), 1), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"BASELINE config 1: OPT-125M fp32 at its true dimensions (random weights), 5 prompts x 16 token ids, 50 "
                      f"new tokens each, temperature 0.6 / top_p 0.95 / top_k 40 (reference examples/cpu_inference_opt.py:"
                      f"143-158), generated tokens / wall time of generate() incl. the prefill step; 1 untimed warm-up run, "
                      f"then the median of {len(runs)} runs ({min(runs):.1f}-{max(runs):.1f}); host engine of this build + "
                      f"oracle/model.py as its runner.  The reference itself measured 28.6 tokens/s (BASELINE.md) and the "
                      f"survey container 120.6 tokens/s on 8 threads with the real checkpoint"}


def secondary_records(args, engine, workdir):
    """(N = 1) The rest of BASELINE.json's metric on the driver's line: prefill tokens/s on the config-4 shapes for the
    headline model (on the engine the headline just ran on), then Qwen2-7B: decode B=256 ctx=1024 with its own
    roofline object, and its prefill tokens/s.  Returns (records, engine-to-exit)."""
    recs = []
    release_all(engine)
    recs += prefill_records(engine, args.model)
    if args.model == "qwen2-7b":
        return recs, engine
    engine.exit()
    del engine
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    other = "qwen2-7b"
    a2 = argparse.Namespace(**vars(args))
    a2.model, a2.batch, a2.ctx, a2.steps, a2.warmup = other, 256, 1024, max(20, min(args.steps, 64)), max(5, min(args.warmup, 8))
    eng2 = build_engine(a2, os.path.join(workdir, other), other)
    fill_prompts(eng2, a2, other)
    elapsed = timed_decode(eng2, a2)
    roof, _ = decode_roofline(eng2, a2, other)
    ex = eng2.model_runner.inference_executor
    recs.append({"metric": "decode tokens/s", "model": other, "value": round(a2.batch * a2.steps / elapsed, 1),
                 "unit": "tokens/s", "steps": a2.steps, "warmup": a2.warmup,
                 "ms_per_step": round(elapsed / a2.steps * 1e3, 4), "dtype": "bf16",
                 "config": {"workload": f"{other} decode, B={a2.batch}, ctx={a2.ctx}, kvcache_block_size=64, "
                                        f"max_num_seqs={a2.batch}, {'hipGraph' if (ex.graphs or ex.greedy_graphs) else 'eager'} decode, greedy",
                            "kv_blocks": eng2.config.num_kvcache_blocks},
                 "roofline": roof})
    release_all(eng2)
    recs += prefill_records(eng2, other)
    return recs, eng2


def main():
    args = parse()
    if args.cpu_config1_only:
        print(json.dumps(cpu_config1()))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (the driver's `python3 bench.py --gpus N`): become the launcher.  Nothing in this process has
        # touched the GPU (importing torch does not), and the ranks are CHILD processes, never an exec of this one.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
        proc = subprocess.run(launch_command(args), env=env)
        raise SystemExit(proc.returncode)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCHELASTIC_RUN_ID", "bench")
    import torch.distributed as dist

    secondary = None
    with tempfile.TemporaryDirectory() as workdir:
        engine = build_engine(args, workdir)
        prefill_tps = fill_prompts(engine, args)
        elapsed = timed_decode(engine, args, dist)
        roof, ar_probe = decode_roofline(engine, args)
        ex = engine.model_runner.inference_executor
        graphs = bool(ex.graphs or ex.greedy_graphs)
        allreduce = tp_info = None
        if args.gpus > 1:
            from minivllm_amd import parallel_state
            from minivllm_amd.custom_allreduce import describe_status

            ar = parallel_state.get_custom_all_reduce()
            words = ar.status_words() if ar is not None else None
            allreduce = "rccl" if ar is None else (
                f"p2p two-shot kernel (validated against the backend at start-up, {ar.calls} calls)"
                if words[0] == 0 else f"p2p two-shot kernel: RESULT INVALID, {describe_status(words)}")
            # every rank must have generated the same tokens (SPMD lock-step): compare a digest of all sequences
            import hashlib

            digest = hashlib.sha256(json.dumps([s.completion_token_ids for s in engine.scheduler.running]).encode()).hexdigest()
            digests = [None] * world
            dist.all_gather_object(digests, digest, group=engine.model_runner.distributed_manager.control_group)
            assert len(set(digests)) == 1, f"ranks generated different tokens: {digests}"
            probes = [None] * world
            dist.all_gather_object(probes, ar_probe, group=engine.model_runner.distributed_manager.control_group)
            # what the process group actually is: backend "nccl" = RCCL on ROCm, and how many ranks it spans
            tp_info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                       "activation_all_reduce": "rccl" if ar is None else "p2p",
                       "all_reduces_per_step": 2 * MODELS[args.model]["num_hidden_layers"] + 1,
                       "logit_exchange": "per-rank argmax merge (greedy batch): no [B, V/tp] all-gather",
                       "tokens_identical_on_all_ranks": True,
                       "allreduce_probe_per_rank": probes}
        kv_blocks = engine.config.num_kvcache_blocks
        if args.gpus == 1 and not args.no_secondary and not args.synthetic_kv:
            secondary, engine = secondary_records(args, engine, workdir)
        engine.exit()
    if rank != 0:
        return
    cpu = None
    if not args.no_cpu_baseline and args.gpus == 1:          # rank 0, N=1 only
        cpu = cpu_baseline(args)
        cpu["config1"] = cpu_config1()
    value = args.batch * args.steps / elapsed
    line = {
        "metric": "decode tokens/s", "value": round(value, 1), "unit": "tokens/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} decode, B={args.batch}, ctx={args.ctx}, kvcache_block_size=64, "
                               f"max_num_seqs={args.batch}, {'hipGraph' if graphs else 'eager'} decode, greedy",
                   "model": args.model, "global_batch": args.batch, "seq_len": args.ctx,
                   "parallelism": f"tp{args.gpus}", "allreduce": allreduce, "tensor_parallel": tp_info,
                   "kv_blocks": kv_blocks, "kv_sizing": "from free HBM (auto)" if args.auto_kv else "explicit",
                   "prefill": "synthetic KV" if args.synthetic_kv else "engine prefill"},
        "prefill_tokens_per_s": None if prefill_tps is None else round(prefill_tps, 1),
        "prefill_tokens_per_s_method": "the B x ctx prompt tokens of this run / wall time of the engine's prefill steps (first "
                                       "steps of a fresh engine: includes one-time lazy set-up; the steady-state figures are in `secondary`)",
        "roofline": roof, "cpu_baseline": cpu, "secondary": secondary,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
