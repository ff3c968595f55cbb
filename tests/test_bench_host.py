"""bench.py host logic that needs no GPU: the N > 1 self-launch command, the usable-core count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_multi_gpu_bench_launches_itself_under_torch_distributed_run():
    """`python3 bench.py --gpus N` (the driver's command) must become `python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N ...` with the same arguments, rendezvous on 127.0.0.1."""
    args = bench.parse(["--gpus", "4", "--steps", "7", "--warmup", "3", "--auto-kv", "--no-cpu-baseline"])
    cmd = bench.launch_command(args, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    tail = cmd[script + 1:]
    again = bench.parse(tail)                                  # the child parses to the same request
    assert (again.gpus, again.steps, again.warmup, again.auto_kv, again.no_cpu_baseline) == (4, 7, 3, True, True)
    assert again.model == args.model and again.batch == args.batch and again.ctx == args.ctx
    # a free port is chosen when none is given
    port = int(bench.launch_command(args)[bench.launch_command(args).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_usable_cores_is_positive_and_bounded():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_record_plan_carries_the_whole_metric_for_every_gpu_count():
    """BASELINE.json: "decode tokens/sec/GPU + prefill tok/s, Qwen3-1.7B TP=1 & Qwen2-7B TP=1/2/4/8", config 5 =
    Qwen2-72B TP=8.  Every `bench.py --gpus N` the driver may run must have a producer for its share: the headline
    first, then (N=1) sampled decode + the three config-4 prefill shapes + Qwen2-7B; (N=2,4) Qwen2-7B at TP=N;
    (N=8) the Qwen2-7B rejection the reference's head rule demands (qwen_base.py:95-104) and Qwen2-72B on an
    HBM-sized KV pool with max_model_len=4096."""
    for n in (1, 2, 4, 8):
        plan = bench.record_plan(n)
        assert plan[0] == dict(kind="decode", model="qwen3-1.7b", tp=n, batch=256, ctx=1024, sampling="greedy", headline=True)
        assert all(p["tp"] == n for p in plan) and sum(bool(p.get("headline")) for p in plan) == 1
    kinds = lambda n: [(p["kind"], p["model"], p.get("sampling") or (p.get("prompts"), p.get("prompt_len"))) for p in bench.record_plan(n)[1:]]
    assert kinds(1) == [("decode", "qwen3-1.7b", "sampled"),
                        ("prefill", "qwen3-1.7b", (16, 1024)), ("prefill", "qwen3-1.7b", (4, 4096)), ("prefill", "qwen3-1.7b", (64, 256)),
                        ("decode", "qwen2-7b", "greedy"),
                        ("prefill", "qwen2-7b", (16, 1024)), ("prefill", "qwen2-7b", (4, 4096)), ("prefill", "qwen2-7b", (64, 256))]
    for n in (2, 4):
        assert kinds(n) == [("decode", "qwen2-7b", "greedy"), ("prefill", "qwen2-7b", (16, 1024)), ("prefill", "qwen2-7b", (4, 4096))]
    assert kinds(8) == [("rejected", "qwen2-7b", (None, None)), ("decode", "qwen2-72b", "greedy")]
    big = bench.record_plan(8)[-1]
    assert big["auto_kv"] and big["max_model_len"] == 4096 and (big["batch"], big["ctx"]) == (256, 1024)
    # the divisibility rule itself, as the reference states it
    assert [bench.tp_allowed("qwen2-7b", n) for n in (1, 2, 4, 8)] == [True, True, True, False]
    assert all(bench.tp_allowed(m, 8) for m in ("qwen3-1.7b", "qwen2-72b"))


def test_projection_plan_ingest_figures_for_the_headline_shapes():
    """The second bound next to the HBM fraction of linear_mid_m_kernel: what the busiest CU takes in.  Qwen3-1.7B at 256
    rows, the tuned tiles (tuned/mid_m_gfx950.json): 2.49 MB per CU and layer -> 34-38 us at 66-73 GB/s per CU."""
    shapes = (("col", 4096, 2048), ("row", 2048, 2048), ("gated", 6144, 2048), ("row", 2048, 6144))
    plans = [bench.projection_plan(kind, 256, n, k) for kind, n, k in shapes]
    assert all(p is not None and p["workgroups"] > 0 for p in plans)
    for p, (kind, n, k) in zip(plans, shapes):
        bm, bn = p["tile"]
        bno = bn // 2 if kind == "gated" else bn
        wgs = -(-256 // bm) * -(-n // bno) * p["k_splits"]
        assert p["workgroups"] == wgs
        assert p["ingest_bytes_per_cu"] == -(-wgs // 256) * (bm + bn) * (k // p["k_splits"]) * 2
        lo, hi = p["ingest_bound_us"]
        assert lo < hi and abs(hi - p["ingest_bytes_per_cu"] / 66e3) < 0.01
    total = sum(p["ingest_bytes_per_cu"] for p in plans)
    assert 2.0e6 < total < 3.0e6


def test_prefill_flops_formula():
    # 4 H D (Sq Sk - Sq (Sq - 1) / 2); S = 4096, H = 28, D = 128 -> 1.20e11 per sequence and layer
    assert abs(bench.prefill_flops(28, 128, [4096]) / 1.20e11 - 1) < 0.01
    assert bench.prefill_flops(16, 128, [1024] * 16) == 16 * 4 * 16 * 128 * (1024 * 1024 - 1024 * 1023 // 2)


def test_run_records_executes_the_n8_plan_in_order(monkeypatch, tmp_path):
    """The N = 8 run end to end on stand-ins (no GPU): headline engine, then the Qwen2-7B engine construction that the head
    rule rejects -> a record with the error text (every rank raises the same error, so the ranks stay in step), then ONE
    Qwen2-72B engine with a KV pool sized from free HBM and max_model_len 4096, capturing only the bucket it replays;
    engines are exited before the next one is built."""
    events = []

    class FakeEngine:
        def __init__(self, model):
            self.model = model
            ex = type("Ex", (), {})()
            ex.kv_cache = type("KV", (), {"shape": (2, 80, 16384, 64, 1, 128), "numel": lambda self: 2 * 80 * 16384 * 64 * 128})()
            ex.dtype = type("DT", (), {"itemsize": 2})()
            self.model_runner = type("MR", (), {"inference_executor": ex})()

        def exit(self):
            events.append(("exit", self.model))

    def fake_build(args, workdir, model=None, max_model_len=None, auto_kv=None):
        events.append(("build", model, args.gpus, max_model_len, auto_kv, os.environ.get("MV_GRAPH_BATCH_SIZES")))
        if model == "qwen2-7b" and args.gpus == 8:
            raise ValueError("total_num_heads (28) must be divisible by tp_size (8)")
        return FakeEngine(model)

    def fake_decode(engine, args, model, dist, sampling=None, seed=0, probes=True):
        events.append(("decode", model, args.batch, args.ctx, args.steps))
        return {"metric": "decode tokens/s", "model": model, "tp": args.gpus, "value": 1.0, "ms_per_step": 1.0,
                "config": {"workload": model}}, 123.0

    monkeypatch.setattr(bench, "build_engine", fake_build)
    monkeypatch.setattr(bench, "decode_record", fake_decode)
    monkeypatch.setattr(bench, "release_all", lambda e: events.append(("release", e.model)))
    monkeypatch.setattr(bench, "projection_roofline", lambda e, rows: {"rows": rows})
    monkeypatch.delenv("MV_GRAPH_BATCH_SIZES", raising=False)
    args = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    records, tps = bench.run_records(args, str(tmp_path), dist=None)
    assert [r["model"] for r in records] == ["qwen3-1.7b", "qwen2-7b", "qwen2-72b"] and tps == 123.0
    assert records[0]["headline"] and records[0]["projection_roofline"] == {"rows": 256}
    rej = records[1]
    assert rej["status"] == "rejected, as in reference" and rej["value"] is None and "divisible by tp_size (8)" in rej["error"]
    big = records[2]
    pool = big["config"]["kv_pool_per_gpu"]
    assert pool["blocks"] == 16384 and pool["local_kv_heads"] == 1 and pool["block_bytes"] == 2 * 80 * 64 * 1 * 128 * 2
    builds = [e for e in events if e[0] == "build"]
    assert builds[0][1:5] == ("qwen3-1.7b", 8, None, None) and builds[0][5] is None          # the headline captures every bucket
    assert builds[1][1] == "qwen2-7b"
    assert builds[2][1:5] == ("qwen2-72b", 8, 4096, True) and builds[2][5] == "256"          # one bucket, HBM-sized pool
    assert "MV_GRAPH_BATCH_SIZES" not in os.environ
    # the headline engine is gone before the rejected construction, the last engine at the end
    order = [e[:2] for e in events if e[0] in ("build", "exit")]
    assert order == [("build", "qwen3-1.7b"), ("exit", "qwen3-1.7b"), ("build", "qwen2-7b"), ("build", "qwen2-72b"), ("exit", "qwen2-72b")]
