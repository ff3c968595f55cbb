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

What else `--gpus N` puts on the line (BASELINE.json's metric is "Qwen3-1.7B TP=1 & Qwen2-7B TP=1/2/4/8", config 5 is
Qwen2-72B TP=8): see ``record_plan`` -- N=1: sampled decode + prefill for the headline model, Qwen2-7B decode + prefill;
N=2/4: Qwen2-7B decode + prefill at TP=N; N=8: the Qwen2-7B TP=8 rejection (28/4 heads, as in the reference) and
Qwen2-72B TP=8 decode with its per-GPU KV pool.  No scaling curve has been measured on hardware by the builder (one-GPU
boxes): the N>1 path is rehearsed with ranks time-slicing one card (profiles/r04_bench_lines.md).

Also reported on the same line:
  roofline     — the dominant kernel (paged-KV decode attention): algorithmic bytes per launch /
                 average launch time, timed with HIP events on the launching stream on the LIVE
                 block tables / context lengths / KV cache of this run
  projection_roofline — the decode-batch projection kernel (linear_mid_m_kernel): weight bytes / HIP-event time of the four
                 projections of every layer replayed from a captured graph (HBM bound), next to the bound that binds it at
                 256 rows: what one CU must take in, (BM + BN) x K x 2 bytes per workgroup, at the guide's 66-73 GB/s per CU
  step_roofline — the whole decode step: weights touched once + the KV bytes of the batch, per GPU, over ms_per_step
  cpu_baseline — the CPU restatement of the reference's eager path (oracle/, "port") timed on this
                 host's cores on a bounded sample of the same model (one untimed warm-up step, then the
                 median of a fixed number of steps), plus BASELINE config 1 (OPT-125M fp32, the preset of
                 the reference's examples/cpu_inference_opt.py) as `config1`
  secondary    — the rest of BASELINE.json's metric under the same clock (``record_plan``); every prefill record carries
                 a roofline object of the flash kernel (HIP-event time of the mv_prefill_varlen launches of one extra
                 eager batch, flops = sum 4 H D (Sq Sk - Sq (Sq - 1) / 2), peak 2.5 PFLOP/s dense bf16, MFMA-busy quoted
                 from the committed PMC profile)
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
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (the 5 PF headline includes 2:1 sparsity)
L2_INGEST_GBS_PER_CU = (66.0, 73.0)  # MI355X_MICROARCH.md "Indexed rows: gather into LDS": rows shared by every workgroup, served by the XCD's L2
PREFILL_CASES = ((16, 1024), (4, 4096), (64, 256))        # SURVEY §8(d) config 4 prompt shapes
SAMPLED = dict(temperature=0.6, top_p=0.95, top_k=40)      # reference examples/cpu_inference_opt.py:153-158
_T0 = time.perf_counter()


def note(msg: str) -> None:
    """Progress on stderr (the JSON line on stdout stays the only stdout output): a long multi-record run must not look hung."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


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
                    help="size the KV pool from free HBM (num_kvcache_blocks=-1: the DEFAULT user path, LLM(Config()); "
                         "1.5 %% slower than the explicit pool on the headline, profiles/r03_kv_stride_probe.md) instead of "
                         "allocating exactly what this workload needs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary records (the rest of BASELINE.json's metric: record_plan)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-baseline steps (median), after one warm-up")
    ap.add_argument("--rehearse-n8", action="store_true",
                    help="(with --gpus 2, ranks sharing one card) run the N = 8 record flow on stand-ins: a Qwen2-7B-like model "
                         "whose heads do not divide by 2 (-> the rejection record) and a 4-layer model at Qwen2-72B's per-layer "
                         "dimensions on an explicit KV pool; the records are marked as a rehearsal")
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
    for flag in ("eager", "synthetic_kv", "auto_kv", "no_cpu_baseline", "no_secondary", "rehearse_n8"):
        if getattr(args, flag):
            cmd.append("--" + flag.replace("_", "-"))
    return cmd


def tp_allowed(model: str, tp: int) -> bool:
    """The reference's divisibility rule (models/qwen_base.py:95-104): q heads and kv heads must divide by the TP degree."""
    c = MODELS[model]
    return c["num_attention_heads"] % tp == 0 and c["num_key_value_heads"] % tp == 0 and c["vocab_size"] % tp == 0


def record_plan(gpus: int, model: str = "qwen3-1.7b", batch: int = 256, ctx: int = 1024) -> list[dict]:
    """What one ``bench.py --gpus N`` run measures, in order (the first entry is the driver's headline, the others become
    ``secondary``).  BASELINE.json: metric = "decode tokens/sec/GPU + prefill tok/s, Qwen3-1.7B TP=1 & Qwen2-7B
    TP=1/2/4/8"; config 5 = Qwen2-72B TP=8, max_model_len=4096, 288 GB KV sizing.

      N=1    headline model: greedy decode (headline), sampled decode (the reference's preset), prefill on the three config-4
             shapes; Qwen2-7B: greedy decode, prefill on the three shapes
      N=2,4  headline model at TP=N (strong scaling point); Qwen2-7B at TP=N: greedy decode, prefill 16x1024 / 4x4096
      N=8    headline model at TP=8; Qwen2-7B TP=8 -> "rejected, as in reference" (28/4 heads); Qwen2-72B at TP=8
             (config 5): greedy decode B=256 ctx=1024 on a KV pool sized from free HBM, max_model_len=4096
    Pure host logic (tests/test_bench_host.py)."""
    plan = [dict(kind="decode", model=model, tp=gpus, batch=batch, ctx=ctx, sampling="greedy", headline=True)]
    if gpus == 1:
        plan.append(dict(kind="decode", model=model, tp=1, batch=batch, ctx=ctx, sampling="sampled"))
        plan += [dict(kind="prefill", model=model, tp=1, prompts=n, prompt_len=ln) for n, ln in PREFILL_CASES]
        if model != "qwen2-7b":
            plan.append(dict(kind="decode", model="qwen2-7b", tp=1, batch=256, ctx=1024, sampling="greedy"))
            plan += [dict(kind="prefill", model="qwen2-7b", tp=1, prompts=n, prompt_len=ln) for n, ln in PREFILL_CASES]
        return plan
    if tp_allowed("qwen2-7b", gpus):
        if model != "qwen2-7b":
            plan.append(dict(kind="decode", model="qwen2-7b", tp=gpus, batch=256, ctx=1024, sampling="greedy"))
        plan += [dict(kind="prefill", model="qwen2-7b", tp=gpus, prompts=n, prompt_len=ln) for n, ln in PREFILL_CASES[:2]]
    else:
        plan.append(dict(kind="rejected", model="qwen2-7b", tp=gpus))
    if gpus == 8:
        plan.append(dict(kind="decode", model="qwen2-72b", tp=8, batch=256, ctx=1024, sampling="greedy", auto_kv=True,
                         max_model_len=4096))
    return plan


def build_engine(args, workdir, model=None, max_model_len=None, auto_kv=None):
    from minivllm_amd import LLM, Config

    model = model or args.model
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "config.json"), "w") as f:
        json.dump(MODELS[model], f)
    max_len = max_model_len or max(4096, args.ctx + args.steps + args.warmup + 64)
    auto_kv = args.auto_kv if auto_kv is None else auto_kv
    cfg = Config(workdir, dtype="bfloat16", kvcache_block_size=64, max_num_seqs=args.batch,
                 max_model_len=max_len, max_num_batched_tokens=max(16384, max_len),
                 tensor_parallel_size=args.gpus, enforce_eager=args.eager, load_format="dummy", seed=0,
                 num_kvcache_blocks=-1 if auto_kv else
                 args.batch * (-(-(args.ctx + args.steps + args.warmup + 8) // 64) + 1))
    return LLM(cfg)


def fill_prompts(engine, args, model=None, sampling=None, seed=0):
    """B requests of ctx synthetic tokens each -> running sequences with allocated blocks."""
    from minivllm_amd import SamplingParams

    g = torch.Generator().manual_seed(seed)
    vocab = MODELS[model or args.model]["vocab_size"]
    sp = SamplingParams(max_tokens=args.steps + args.warmup + 16, ignore_eos=True, **(sampling or dict(temperature=0.0)))
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


def _event_pair_floor_us(device) -> float:
    """What an event pair itself costs: the same bracket around a one-element fill, inside a stream that is busy."""
    tiny, floor = torch.zeros(1, device=device), []
    busy = torch.empty(256 << 20, dtype=torch.uint8, device=device)
    for _ in range(20):
        busy.zero_()                        # ~50 us of work ahead of the pair: the host is not what it times
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tiny.zero_()
        e1.record()
        floor.append((e0, e1))
    torch.cuda.synchronize()
    del busy
    return sorted(a.elapsed_time(b) for a, b in floor)[len(floor) // 2] * 1e3


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
    # (steps of <= 16 rows run all layers in one persistent launch: the probe times the attention kernel on its own, so
    #  its steps take the multi-launch path, which computes the same bits)
    from minivllm_amd.glue import qwen as _qwen
    was_persistent, _qwen.PERSISTENT = _qwen.PERSISTENT, False
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
        _qwen.PERSISTENT = was_persistent
    torch.cuda.synchronize()
    times = [t * 1e-3 for t in probe.times_us("decode")]      # ms
    assert len(times) == probe_steps * L
    floor_us = _event_pair_floor_us(ex.device)
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
    for pmc in ("r04_pmc_decode.json", "r03_pmc_decode.json", "r02_pmc_decode.json"):
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
            "heads_per_rank": [heads, hkv],
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


@torch.inference_mode()
def _graph_time_us(fn, launches: int, reps: int = 5) -> float:
    """Microseconds per launch of ``fn`` (which enqueues ``launches`` kernels on the current stream): captured into a
    hipGraph -- how the decode step itself runs them -- and replayed between one HIP event pair per replay (on the
    replaying stream), median over ``reps``; the event pair and the replay's fixed cost are spread over all launches."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    del graph
    return statistics.median(a.elapsed_time(b) for a, b in pairs) * 1e3 / launches


def projection_plan(kind: str, rows: int, n: int, k: int) -> dict | None:
    """Tile, K split, workgroup count and per-CU ingest of the mv_linear_mid_m launch that ``layers/linear.py`` picks for
    this projection (None: the library GEMM).  ``n`` = output features (gated: the intermediate size).  The ingest
    figure is what the busiest CU must take in: every workgroup reads its BM rows of x and its BN weight rows over its
    K slice, (BM + BN) x K/splits x 2 bytes, and a CU runs ceil(workgroups / 256) of them.  Host logic."""
    from minivllm_amd.layers.linear import MID_CFGS, mid_plan

    cfg, splits = mid_plan(kind, rows, n, k)
    if not cfg:
        return None
    bm, bn, _ = MID_CFGS[cfg]
    bno = bn // 2 if kind == "gated" else bn
    wgs = -(-rows // bm) * -(-n // bno) * splits
    per_cu = -(-wgs // 256) * (bm + bn) * (k // splits) * 2
    lo, hi = L2_INGEST_GBS_PER_CU
    return {"cfg": cfg, "tile": [bm, bn], "k_splits": splits, "workgroups": wgs, "ingest_bytes_per_cu": per_cu,
            "ingest_bound_us": [round(per_cu / hi / 1e3, 2), round(per_cu / lo / 1e3, 2)]}


def projection_roofline(engine, rows: int):
    """The decode-batch projection kernel (csrc/linear_mid_m.hip) on this engine's own weights at ``rows`` tokens: per
    projection kind, one launch per LAYER (28 different weight matrices, so every launch streams its weights from HBM
    like the step does) captured into a graph and replayed; weight bytes / time against the HBM peak, and next to it the
    per-CU ingest bound (``projection_plan``) -- at 256 rows the kernel is bound by how fast a CU takes bytes in, not
    by HBM (profiles/r03_mid_m_ablation.md), and 0.2 of 8 TB/s says little without that second number."""
    from minivllm_amd import hip_ops
    from minivllm_amd.layers.linear import linear, mid_m

    ex = engine.model_runner.inference_executor
    layers = list(ex.model.model.layers)
    dev, dt = ex.device, ex.dtype
    g = torch.Generator(device=dev).manual_seed(7)
    kinds = (("qkv_proj", lambda l: l.self_attn.qkv_proj, "col"), ("o_proj", lambda l: l.self_attn.o_proj, "row"),
             ("gate_up_proj", lambda l: l.mlp.gate_up_proj, "gated"), ("down_proj", lambda l: l.mlp.down_proj, "row"))
    per, tot_b, tot_us, tot_lo, tot_hi = [], 0, 0.0, 0.0, 0.0
    for name, get, kind in kinds:
        mods = [get(l) for l in layers]
        w = mods[0].weight
        n, k = (w.shape[0] // 2 if kind == "gated" else w.shape[0]), w.shape[1]
        x = torch.randn((rows, k), device=dev, dtype=torch.float32, generator=g).mul_(0.05).to(dt)
        # the kind layers/linear.py resolves: a row-parallel projection may K-split into fp32 partials only on a single
        # rank (RowParallelLinear.forward), and a bias forces the plain form (linear())
        if kind == "row":
            eff = "row" if mods[0].tp_size == 1 and mods[0].bias is None else "col"
        else:
            eff = kind
        ours = mid_m(x, mods[0].weight_packed) and hip_ops.linear_mid_m_ok(x, n)
        plan = projection_plan(eff, rows, n, k) if ours else None
        if kind == "gated":
            if plan is None:
                continue          # (library GEMM + SiluAndMul launch: not this kernel)

            def run(mods=mods, x=x, n=n, cfg=plan["cfg"]):
                for m in mods:
                    hip_ops.gate_up_silu_mid_m(x, m.weight_packed, n, cfg)
        else:
            def run(mods=mods, x=x, eff=eff, row=kind == "row"):
                for m in mods:
                    linear(x, m.weight, None if row and m.tp_rank != 0 else m.bias, m.weight_packed, eff)
        us = _graph_time_us(run, len(mods))
        wbytes = w.numel() * w.element_size()
        rec = {"projection": name, "out_features": n, "in_features": k, "weight_bytes": wbytes, "us_per_launch": round(us, 2),
               "GBs": round(wbytes / us / 1e3, 1), "frac_of_hbm_peak": round(wbytes / us / 1e3 / HBM_PEAK_GBS, 4),
               "kernel": "linear_mid_m_kernel" if plan else "library GEMM (hipBLASLt)"}
        if plan:
            rec.update(plan)
            rec["frac_of_ingest_bound"] = [round(b / us, 3) for b in plan["ingest_bound_us"]]
            tot_lo += plan["ingest_bound_us"][0]
            tot_hi += plan["ingest_bound_us"][1]
        per.append(rec)
        tot_b += wbytes
        tot_us += us
    if not per:
        return None
    gbs = tot_b / tot_us / 1e3
    all_ours = all("cfg" in r for r in per)
    return {"bound": "hbm", "kernel": "linear_mid_m_kernel", "rows": rows,
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "weight_bytes_per_layer": tot_b, "us_per_layer": round(tot_us, 2), "launches_per_layer": len(per),
            "second_bound": None if not all_ours else {
                "what": "per-CU L2 ingest: every workgroup takes in (BM + BN) x K/splits x 2 bytes (its rows of x from L2, its weight "
                        "rows), the busiest CU ceil(workgroups / 256) of them, at the guide's 66-73 GB/s per CU for rows served by the "
                        "XCD's L2 (MI355X_MICROARCH.md, 'Indexed rows: gather into LDS'); launch fill / drain not included",
                "GBs_per_cu": list(L2_INGEST_GBS_PER_CU), "bound_us_per_layer": [round(tot_lo, 2), round(tot_hi, 2)],
                "frac": [round(tot_lo / tot_us, 3), round(tot_hi / tot_us, 3)]},
            "timing": f"one launch per layer ({len(layers)} weight matrices) captured into a hipGraph, replayed between one HIP event pair "
                      "on the replaying stream, median of 5 replays / launches",
            "per_projection": per}


def step_roofline(engine, model, batch, ctx_mid, ms_per_step):
    """The whole decode step against HBM (SURVEY §8 d): per GPU, every matrix weight of this rank touched once (embedding
    rows excluded, the tied / untied LM head shard included) + the K/V bytes of the batch at its mean context."""
    ex = engine.model_runner.inference_executor
    seen, wbytes = set(), 0
    for mod in ex.model.modules():
        w = getattr(mod, "weight", None)
        if w is None or w.dim() != 2 or w.data_ptr() in seen:
            continue
        if hasattr(mod, "input_size") or type(mod).__name__ == "ParallelLMHead":
            seen.add(w.data_ptr())
            wbytes += w.numel() * w.element_size()
    L, nb, bs, hkv, d = ex.kv_cache.shape[1:]
    kv = L * batch * ctx_mid * 2 * hkv * d * ex.dtype.itemsize
    gbs = (wbytes + kv) / (ms_per_step * 1e-3) / 1e9
    return {"bound": "hbm", "weight_bytes_per_gpu": wbytes, "kv_bytes_per_step_per_gpu": int(kv),
            "bytes_per_step_per_gpu": int(wbytes + kv), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "mean_context": ctx_mid}


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


def prefill_flops(heads: int, head_dim: int, lens) -> int:
    """Causal flash prefill without a cached prefix, one layer (SURVEY §8 d, K2): sum over sequences of
    4 H D (Sq Sk - Sq (Sq - 1) / 2) with Sq = Sk = len."""
    return sum(4 * heads * head_dim * (n * n - n * (n - 1) // 2) for n in lens)


def _mfma_busy_quote():
    for name in ("r04_pmc_prefill.json", "r03_pmc_prefill.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                for key, rec in json.load(f).items():
                    if isinstance(rec, dict) and "mfma_utilisation" in rec:
                        return rec["mfma_utilisation"], f"profiles/{name}: {key.split(' (')[0]} -- SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), rocprofv3 --pmc in a pass of its own (counters cannot be read inside a timed run)"
    return None, None


def prefill_records(engine, model, gpus=1, cases=PREFILL_CASES, reps=5):
    """Prefill tokens/s of the whole model step (SURVEY §8 d(ii), config-4 prompt shapes): n prompts x len fresh
    random tokens through ``execute_batch(prefill=True)`` -- input build, H2D, every layer, LM head on the last
    tokens, sampler, D2H of the n tokens.  One warm-up per shape, then the median of ``reps`` timed batches; then ONE more
    batch whose mv_prefill_varlen launches (one per layer) are bracketed by HIP event pairs -> ``roofline``."""
    from minivllm_amd import hip_ops
    from minivllm_amd.engine.sequence import Sequence
    from minivllm_amd.layers import attention as attn_mod

    ex, bm = engine.model_runner.inference_executor, engine.scheduler.block_manager
    cfgm = MODELS[model]
    vocab = cfgm["vocab_size"]
    heads = cfgm["num_attention_heads"] // gpus
    head_dim = cfgm.get("head_dim") or cfgm["hidden_size"] // cfgm["num_attention_heads"]
    busy, busy_src = _mfma_busy_quote()
    g = torch.Generator().manual_seed(1)
    out = []
    for n, ln in cases:
        times, probe = [], _EventProbe()
        for rep in range(reps + 2):
            seqs = [Sequence(torch.randint(0, vocab, (ln,), generator=g).tolist(), block_size=64) for _ in range(n)]
            for s in seqs:
                bm.allocate(s)
            probing = rep == reps + 1
            real = hip_ops.prefill_varlen
            if probing:
                attn_mod.hip_ops.prefill_varlen = probe.wrap("prefill", real)
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ex.execute_batch(seqs, prefill=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                attn_mod.hip_ops.prefill_varlen = real
            for s in seqs:
                bm.deallocate(s)
            if rep and not probing:
                times.append(dt)
        med = statistics.median(times)
        t_us = probe.times_us("prefill")
        m = statistics.median(t_us)
        kept = [t for t in t_us if t <= 3 * m]
        us = sum(kept) / len(kept)
        flops = prefill_flops(heads, head_dim, [ln] * n)
        tf = flops / us / 1e6
        out.append({"metric": "prefill tokens/s", "model": model, "tp": gpus, "prompts": n, "prompt_len": ln,
                    "batches_timed": reps, "warmup_batches": 1, "ms_per_batch": round(med * 1e3, 3),
                    "value": round(n * ln / med, 1), "unit": "tokens/s",
                    "timing": "wall clock around execute_batch(prefill=True) bracketed by torch.cuda.synchronize(), median",
                    "roofline": {"bound": "mfma", "kernel": "prefill_mfma_kernel", "achieved": round(tf, 1),
                                 "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                                 "traffic": None, "flops_per_launch": flops, "us_per_launch": round(us, 2),
                                 "launches_timed": len(kept), "heads_per_rank": heads,
                                 "share_of_step": round(us * len(t_us) / (med * 1e6), 4),
                                 "mfma_busy": busy, "mfma_busy_source": busy_src,
                                 "timing": "HIP event pairs around every mv_prefill_varlen launch of one extra (untimed) batch, on the launching stream"}})
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


def _tp_info(engine, args, model, ar_probe, dist):
    """(N > 1) what the process group is, which all-reduce ran, per-rank all-reduce probe, and the proof that every rank
    generated the same tokens (SPMD lock-step): a digest of all sequences compared across ranks."""
    import hashlib

    from minivllm_amd import parallel_state
    from minivllm_amd.custom_allreduce import describe_status

    world = args.gpus
    ar = parallel_state.get_custom_all_reduce()
    words = ar.status_words() if ar is not None else None
    allreduce = "rccl" if ar is None else (
        f"p2p two-shot kernel (validated against the backend at start-up, {ar.calls} calls)"
        if words[0] == 0 else f"p2p two-shot kernel: RESULT INVALID, {describe_status(words)}")
    group = engine.model_runner.distributed_manager.control_group
    digest = hashlib.sha256(json.dumps([s.completion_token_ids for s in engine.scheduler.running]).encode()).hexdigest()
    digests = [None] * world
    dist.all_gather_object(digests, digest, group=group)
    assert len(set(digests)) == 1, f"ranks generated different tokens: {digests}"
    probes = [None] * world
    dist.all_gather_object(probes, ar_probe, group=group)
    # what the process group actually is: backend "nccl" = RCCL on ROCm, and how many ranks it spans
    info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
            "activation_all_reduce": "rccl" if ar is None else "p2p",
            "all_reduces_per_step": 2 * MODELS[model]["num_hidden_layers"] + 1,
            "logit_exchange": "per-rank argmax merge (greedy batch): no [B, V/tp] all-gather",
            "tokens_identical_on_all_ranks": True, "allreduce_probe_per_rank": probes}
    return allreduce, info


def decode_record(engine, args, model, dist, sampling=None, seed=0, probes=True):
    """One decode record on a filled engine: timed steps, attention roofline (+ all-reduce probe), projection roofline,
    whole-step roofline, and under TP the token digest.  ``args`` carries batch / ctx / steps / warmup of THIS record."""
    note(f"{model} tp{args.gpus} B={args.batch} ctx={args.ctx} {'sampled' if sampling else 'greedy'}: prefill of the prompts")
    prefill_tps = fill_prompts(engine, args, model, sampling=sampling, seed=seed)
    note("  timed decode steps")
    elapsed = timed_decode(engine, args, dist)
    ms = elapsed / args.steps * 1e3
    ex = engine.model_runner.inference_executor
    graphs = bool(ex.graphs or ex.greedy_graphs)
    rec = {"metric": "decode tokens/s", "model": model, "tp": args.gpus, "value": round(args.batch * args.steps / elapsed, 1),
           "unit": "tokens/s", "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "dtype": "bf16",
           "config": {"workload": f"{model} decode, B={args.batch}, ctx={args.ctx}, kvcache_block_size=64, max_num_seqs={args.batch}, "
                                  f"{'hipGraph' if graphs and not args.eager else 'eager'} decode, "
                                  + (f"sampled (temperature {sampling['temperature']}, top_p {sampling['top_p']}, top_k {sampling['top_k']}: "
                                     f"reference examples/cpu_inference_opt.py:153-158)" if sampling else "greedy"),
                      "parallelism": f"tp{args.gpus}", "kv_blocks": engine.config.num_kvcache_blocks,
                      "max_model_len": engine.config.max_model_len}}
    if not probes:
        return rec, prefill_tps
    note("  attention roofline probe (eager steps)")
    roof, ar_probe = decode_roofline(engine, args, model)
    rec["roofline"] = roof
    rec["step_roofline"] = step_roofline(engine, model, args.batch, args.ctx + args.warmup + args.steps // 2, ms)
    if args.gpus > 1:
        allreduce, info = _tp_info(engine, args, model, ar_probe, dist)
        rec["config"]["allreduce"], rec["config"]["tensor_parallel"] = allreduce, info
    return rec, prefill_tps


def run_records(args, workdir, dist):
    """Execute ``record_plan``: one engine per model, records in plan order.  Every rank runs the same sequence (SPMD)."""
    import gc

    plan = record_plan(args.gpus, args.model, args.batch, args.ctx)
    if args.rehearse_n8:
        # the N = 8 flow (rejection, then a third engine with the 72B layer shape) on what two ranks on one card can hold
        odd = dict(MODELS["qwen2-7b"], num_attention_heads=7, num_key_value_heads=1, hidden_size=896, num_hidden_layers=2)
        MODELS["qwen2-7b-odd-heads (rehearsal)"] = odd
        MODELS["qwen2-72b-4-layers (rehearsal)"] = dict(MODELS["qwen2-72b"], num_hidden_layers=4)
        plan = [plan[0], dict(kind="rejected", model="qwen2-7b-odd-heads (rehearsal)", tp=args.gpus),
                dict(kind="decode", model="qwen2-72b-4-layers (rehearsal)", tp=args.gpus, batch=256, ctx=1024, sampling="greedy",
                     max_model_len=4096)]
    if args.no_secondary or args.synthetic_kv:
        plan = plan[:1]
    records, engine, engine_model, engine_args, first_on_engine = [], None, None, None, True
    headline_prefill_tps = None
    for i, item in enumerate(plan):
        model = item["model"]
        headline = bool(item.get("headline"))
        if item["kind"] == "rejected":
            if engine is not None:
                engine.exit()
                engine, engine_model = None, None
                gc.collect()
                torch.cuda.empty_cache()
            note(f"{model} tp{item['tp']}: expecting the reference's rejection")
            a2 = argparse.Namespace(**vars(args))
            a2.model, a2.batch, a2.ctx = model, 256, 1024
            try:
                eng = build_engine(a2, os.path.join(workdir, f"{model}-rejected"), model)
            except ValueError as e:
                records.append({"metric": "decode tokens/s", "model": model, "tp": item["tp"], "value": None,
                                "status": "rejected, as in reference", "error": f"ValueError: {e}",
                                "reference": "minivllm/models/qwen_base.py:95-104 (q / kv heads must divide by the TP degree: "
                                             f"{MODELS[model]['num_attention_heads']} / {MODELS[model]['num_key_value_heads']} by "
                                             f"{item['tp']}); the build raises the same error instead of silently replicating KV heads"})
            else:
                eng.exit()
                raise AssertionError(f"{model} at TP={item['tp']} was accepted: the reference rejects it")
            continue
        if engine_model != model:
            if engine is not None:
                engine.exit()
                del engine
                gc.collect()
                torch.cuda.empty_cache()
            a2 = argparse.Namespace(**vars(args))
            if not headline:
                a2.model, a2.batch, a2.ctx = model, item.get("batch", 256), item.get("ctx", 1024)
                a2.steps, a2.warmup = max(20, min(args.steps, 64)), max(5, min(args.warmup, 8))
                if args.gpus > 1:
                    # the secondary TP engines capture only the bucket this workload replays (each bucket x 3 graph
                    # families x up to 80 layers is seconds of start-up the driver's clock would pay 21 times over)
                    os.environ["MV_GRAPH_BATCH_SIZES"] = str(a2.batch)
            note(f"building {model} tp{args.gpus}" + (" (KV pool from free HBM)" if item.get("auto_kv") or args.auto_kv else ""))
            engine = build_engine(a2, os.path.join(workdir, model), model, max_model_len=item.get("max_model_len"),
                                  auto_kv=item.get("auto_kv"))
            os.environ.pop("MV_GRAPH_BATCH_SIZES", None)
            engine_model, engine_args, first_on_engine = model, a2, True
        if item["kind"] == "decode":
            if not first_on_engine:
                release_all(engine)
            sampling = SAMPLED if item["sampling"] == "sampled" else None
            rec, tps = decode_record(engine, engine_args, model, dist, sampling=sampling, seed=0 if first_on_engine else 2,
                                     probes=sampling is None)
            if sampling is None and engine_args.batch > 16:
                note("  projection roofline (graph replays)")
                release_all(engine)
                rec["projection_roofline"] = projection_roofline(engine, engine_args.batch)
            else:
                release_all(engine)
            if item.get("auto_kv"):
                ex = engine.model_runner.inference_executor
                L, nb, bs, hkv, d = ex.kv_cache.shape[1:]
                rec["config"]["kv_pool_per_gpu"] = {
                    "blocks": nb, "block_bytes": 2 * L * bs * hkv * d * ex.dtype.itemsize,
                    "GiB": round(ex.kv_cache.numel() * ex.dtype.itemsize / 2 ** 30, 2), "local_kv_heads": hkv,
                    "sizing": "min((total*0.9 - used - peak + current) // block_bytes, max_num_seqs * ceil(max_model_len / 64)), "
                              "MIN over the ranks (reference inference_executor.py:245-318 with the LOCAL kv-head count)"}
            if headline:
                rec["headline"], headline_prefill_tps = True, tps
            records.append(rec)
            first_on_engine = False
        elif item["kind"] == "prefill":
            if not first_on_engine:
                release_all(engine)
            note(f"{model} tp{args.gpus} prefill {item['prompts']} x {item['prompt_len']}")
            records += prefill_records(engine, model, args.gpus, cases=((item["prompts"], item["prompt_len"]),))
            first_on_engine = False
    if engine is not None:
        engine.exit()
    return records, headline_prefill_tps


def init_process_group(args):
    """(N > 1) ONE process group for the whole run, made here and not by the first engine: engines come and go (one per
    model of the plan) and the DistributedManager only destroys a group it made itself."""
    import torch.distributed as dist

    backend = os.environ.get("MINIVLLM_TP_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    kw = {}
    local = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local)
    from datetime import timedelta

    dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timedelta(minutes=10), **kw)
    return dist


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
    dist = None
    if args.gpus > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCHELASTIC_RUN_ID", "bench")
        dist = init_process_group(args)
    with tempfile.TemporaryDirectory() as workdir:
        records, prefill_tps = run_records(args, workdir, dist)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    cpu = None
    if not args.no_cpu_baseline and args.gpus == 1:          # rank 0, N=1 only
        note("cpu baseline (oracle on the host cores)")
        cpu = cpu_baseline(args)
        cpu["config1"] = cpu_config1()
    head = records[0]
    line = {
        "metric": "decode tokens/s", "value": head["value"], "unit": "tokens/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": dict(head["config"], kv_sizing="from free HBM (auto: the default user path)" if args.auto_kv else "explicit",
                       prefill="synthetic KV" if args.synthetic_kv else "engine prefill"),
        # steady state first (the 16 x 1024 record of `secondary`); the fresh engine's own prefill of the B x ctx prompts,
        # which includes one-time lazy set-up, under its own name
        "prefill_tokens_per_s": next((s["value"] for s in records[1:] if s.get("metric") == "prefill tokens/s"
                                      and s.get("model") == args.model and s.get("prompts") == 16), None),
        "prefill_tokens_per_s_method": "steady state: median of 5 timed batches of 16 x 1024 fresh tokens after a warm-up batch "
                                       "(the matching `secondary` record; null when the secondary records are skipped)",
        "prefill_tokens_per_s_first_run": None if prefill_tps is None else round(prefill_tps, 1),
        "prefill_tokens_per_s_first_run_method": "the B x ctx prompt tokens of this run / wall time of the engine's prefill steps "
                                                 "(first steps of a fresh engine: includes one-time lazy set-up)",
        "roofline": head.get("roofline"), "projection_roofline": head.get("projection_roofline"),
        "step_roofline": head.get("step_roofline"), "cpu_baseline": cpu,
        "secondary": records[1:] or None,
        "multi_gpu_note": None if args.gpus == 1 else
        "strong scaling: the same B x ctx workload tensor-parallel over N GPUs; secondary carries the Qwen2-7B / Qwen2-72B part "
        "of BASELINE.json's metric for this N (record_plan)",
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
