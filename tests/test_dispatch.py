"""Tensor-parallel dispatch seam on the host (no GPU): the driver of SPAWNED workers relays every public call --
``add_request`` / ``step`` / ``generate`` -- over the gloo control group (reference: ``ModelRunner.call`` broadcasts
each "run" to the workers, minivllm/engine/model_runner.py:173-198, under the ``add_request``/``step`` loop of
engine/llm_engine.py:171-251), and a dead worker is an exception at the next call, not a wait.

Two real processes, world size 2, gloo on 127.0.0.1; the model runner is a deterministic stand-in (token =
f(sequence)) so that only the dispatch is under test."""
import json
import os
import socket
import sys
import time
from types import SimpleNamespace

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TINY = dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=64, num_hidden_layers=1,
            num_attention_heads=2, num_key_value_heads=1, head_dim=32, intermediate_size=128, vocab_size=320,
            max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=10000.0, tie_word_embeddings=True,
            hidden_act="silu", torch_dtype="float32", eos_token_id=319)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FakeRunner:
    """Stand-in for ModelRunner: a real DistributedManager (gloo), a token rule that depends on the whole sequence,
    a log of the batches it ran, and an optional step at which it raises."""

    def __init__(self, dm, log_path=None, fail_at=None):
        self.distributed_manager = dm
        self.model_manager = SimpleNamespace(tokenizer=None)
        self.log_path, self.fail_at, self.steps = log_path, fail_at, 0

    def call(self, name, *args):
        return getattr(self, name)(*args)

    def run(self, sequences, is_prefill):
        self.steps += 1
        if self.fail_at is not None and self.steps == self.fail_at:
            raise RuntimeError("injected fault")
        toks = [(sum(s.token_ids) * 31 + len(s)) % 300 for s in sequences]
        if self.log_path:
            with open(self.log_path, "a") as f:
                f.write(json.dumps([is_prefill, [s.seq_id for s in sequences], toks]) + "\n")
        return toks

    def exit(self):
        self.distributed_manager.cleanup()


def _config(model_dir):
    from minivllm_amd import Config

    return Config(model_dir, tensor_parallel_size=2, max_num_seqs=4, max_model_len=256, max_num_batched_tokens=256,
                  num_kvcache_blocks=16, dtype="float32", enforce_eager=True)


def _env(port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MINIVLLM_TP_BACKEND="gloo")
    os.environ.pop("TORCHELASTIC_RUN_ID", None)


def _worker(model_dir, port, log_path, fail_at):
    sys.path.insert(0, ROOT)
    _env(port)
    from minivllm_amd.engine.distributed_manager import DistributedManager
    from minivllm_amd.engine.llm_engine import LLMEngine, serve_worker

    cfg = _config(model_dir)
    dm = DistributedManager(cfg, 1)
    dm.initialize()
    engine = LLMEngine(cfg, model_runner=FakeRunner(dm, log_path, fail_at), _rank=1)
    serve_worker(engine, 1)


def _driver(model_dir, port, log_path, fail_at=None):
    _env(port)
    from minivllm_amd.engine.distributed_manager import DistributedManager
    from minivllm_amd.engine.llm_engine import LLMEngine

    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_worker, args=(model_dir, port, log_path + ".1", fail_at))
    proc.start()
    cfg = _config(model_dir)
    dm = DistributedManager(cfg, 0)
    dm.initialize()
    engine = LLMEngine(cfg, model_runner=FakeRunner(dm, log_path + ".0"))
    assert engine._is_driver and engine.rank == 0
    engine.ps.append(proc)
    return engine, proc


@pytest.fixture
def model_dir(tmp_path):
    d = tmp_path / "tiny"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(TINY))
    return str(d)


@pytest.mark.timeout(180)
def test_add_request_and_step_are_relayed_like_generate(model_dir, tmp_path):
    """The reference's public loop on the driver of a spawned TP=2 engine: add_request x3, step until finished, then a
    generate() on the same engine.  The worker must have run EXACTLY the driver's batches (same sequence ids, same
    prefill/decode flags, same tokens) -- before this round only generate() crossed the process boundary and a bare
    step() walked into the all-reduce alone."""
    from minivllm_amd import SamplingParams

    log = str(tmp_path / "log")
    engine, proc = _driver(model_dir, _free_port(), log)
    try:
        sp = SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True)
        prompts = [[1, 2, 3], list(range(10, 90)), [7] * 70]
        ids = [engine.add_request(p, sp) for p in prompts]
        done = {}
        while not engine.is_finished():
            finished, _ = engine.step()
            done.update(finished)
        assert sorted(done) == sorted(ids) and all(len(t) == 5 for t in done.values())
        outs = engine.generate([[5, 6], [9] * 65], SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True),
                               use_tqdm=False)
        assert [len(o["token_ids"]) for o in outs] == [3, 3]
    finally:
        engine.exit()
        proc.join(30)
    assert proc.exitcode == 0
    rows0 = [json.loads(l) for l in open(log + ".0")]
    rows1 = [json.loads(l) for l in open(log + ".1")]
    assert rows0 == rows1 and len(rows0) == (1 + 4) + (1 + 2)       # prefill + 4 decode steps, then prefill + 2


@pytest.mark.timeout(180)
def test_a_dead_worker_is_an_exception_at_the_next_call_not_a_wait(model_dir, tmp_path):
    """The worker's runner raises in its 3rd step: the worker logs, leaves with a non-zero code, and the driver's next
    public call -- step(), add_request() or generate() -- raises at once (it must not relay to, or enter a collective
    with, a rank that is gone; the peer-to-peer all-reduce's own bounded wait is 20 s)."""
    from minivllm_amd import SamplingParams

    log = str(tmp_path / "log")
    engine, proc = _driver(model_dir, _free_port(), log, fail_at=3)
    try:
        engine.add_request([1, 2, 3, 4], SamplingParams(temperature=0.0, max_tokens=16, ignore_eos=True))
        engine.step()
        engine.step()
        engine.step()                      # the worker fails inside this one (the stand-in has no collective to hang in)
        proc.join(30)
        assert proc.exitcode == 1
        t0 = time.perf_counter()
        for call in (lambda: engine.step(), lambda: engine.add_request([1, 2]),
                     lambda: engine.generate([[1, 2]], use_tqdm=False)):
            with pytest.raises(RuntimeError, match="worker process.*died"):
                call()
        assert time.perf_counter() - t0 < 2.0
        assert len(engine.scheduler.waiting) == 0     # the refused add_request queued nothing on the driver either
    finally:
        engine.exit()
