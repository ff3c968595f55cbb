"""Request loop (mirror of minivllm/engine/llm_engine.py:40-367): tokenize -> schedule -> run ->
postprocess until every request has finished.

Tensor parallelism is SPMD.  With ``tensor_parallel_size > 1`` either
  * the process was launched by ``torch.distributed.run`` (RANK/WORLD_SIZE set): every rank builds
    its own engine and makes the same ``add_request`` / ``step`` / ``generate`` calls, or
  * rank 0 spawns the other ranks (one process per GPU, like the reference) and every PUBLIC call of the
    driver -- ``add_request``, ``step``, ``generate`` -- is relayed once over the host control group (gloo,
    one small object per call, never inside a graph) and then runs identically everywhere.  This is the
    reference's dispatch seam, ``ModelRunner.call("run", seqs, is_prefill)`` broadcast to the workers
    (model_runner.py:173-198) under its ``add_request`` / ``step`` loop (llm_engine.py:171-251), with one
    difference: what crosses the process boundary is the CALL, not the pickled batch -- every rank keeps
    its own scheduler.
Scheduling, block allocation and sampling (Philox keyed by seed/step/row) are deterministic, so
all ranks stay in lock-step; ``generate`` needs one relay for the whole request list, not one per step."""
from __future__ import annotations

import atexit
import logging
import os
import socket
import warnings
from time import perf_counter

import torch.multiprocessing as mp

from ..config import Config
from ..sampling_params import SamplingParams
from .model_runner import ModelRunner
from .scheduler import Scheduler
from .sequence import Sequence

logger = logging.getLogger(__name__)
__all__ = ["LLMEngine"]


def _worker_main(config: Config, rank: int) -> None:
    os.environ["LOCAL_RANK"] = str(rank)
    serve_worker(LLMEngine(config, _rank=rank), rank)


def serve_worker(engine: "LLMEngine", rank: int) -> None:
    """Loop of a spawned tensor-parallel worker (reference: ``ModelRunner._worker_loop``, model_runner.py:124-158):
    wait for the driver's next relayed call, make it on this rank's engine, until "exit"."""
    dm = engine.model_runner.distributed_manager
    while True:
        cmd = dm.broadcast_data(None, src=0)
        if cmd[0] == "exit":
            break
        try:
            engine._serve_relayed(cmd)
        except Exception:
            # the driver validated the request before telling us, so this is a real fault: say so and leave
            # with a non-zero code (the driver sees the dead worker and stops using the collective group)
            # NO collective clean-up on the way out (ModelRunner.exit ends in a barrier + destroy_process_group that the
            # other ranks are not in: the worker would sit there, alive, and the driver could not tell)
            logger.exception("tensor-parallel worker %d failed inside %s()", rank, cmd[0])
            logging.shutdown()
            os._exit(1)
    engine.model_runner.exit()


class LLMEngine:
    def __init__(self, config: Config, model_runner=None, _rank: int | None = None) -> None:
        if not isinstance(config, Config):
            raise TypeError(f"config must be a Config instance, got {type(config).__name__}")
        self.config = config
        self.ps: list = []
        self.events: list = []
        tp = config.tensor_parallel_size
        launched = _rank is not None or (tp > 1 and int(os.environ.get("WORLD_SIZE", "1")) == tp
                                         and "RANK" in os.environ and "TORCHELASTIC_RUN_ID" in os.environ)
        self.rank = _rank if _rank is not None else int(os.environ.get("RANK", "0")) if launched else 0
        self._is_driver = tp > 1 and not launched
        if model_runner is not None:            # injected runner (tests drive the host logic with it)
            self.model_runner = model_runner
        else:
            if self._is_driver:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
                ctx = mp.get_context("spawn")
                for i in range(1, tp):
                    p = ctx.Process(target=_worker_main, args=(config, i))
                    p.start()
                    self.ps.append(p)
            self.model_runner = ModelRunner(config, self.rank, self.events)
        mm = getattr(self.model_runner, "model_manager", None)
        self.tokenizer = getattr(mm, "tokenizer", None)
        if self.tokenizer is not None and self.tokenizer.eos_token_id is not None:
            config.eos = self.tokenizer.eos_token_id
        elif config.eos == -1:
            eos = getattr(config.hf_config, "eos_token_id", None)
            config.eos = eos[0] if isinstance(eos, (list, tuple)) else (eos if eos is not None else -1)
        self.scheduler = Scheduler(config)
        atexit.register(self.exit)

    def exit(self) -> None:
        runner = self.__dict__.pop("model_runner", None)
        if runner is None:
            return
        try:
            # a dead worker can no longer take part in a collective: do not block the exit on it
            workers_alive = all(p.is_alive() for p in self.ps)
            if self._is_driver and workers_alive:
                runner.distributed_manager.broadcast_data(("exit",), src=0)
            if workers_alive or not self._is_driver:
                runner.call("exit")
        except Exception as e:
            warnings.warn(f"Errors during engine cleanup: {e}", RuntimeWarning, stacklevel=2)
        for i, p in enumerate(self.ps):
            p.join(timeout=10)
            if p.is_alive():
                warnings.warn(f"Worker process {i} did not terminate gracefully, forcing termination",
                              RuntimeWarning, stacklevel=2)
                p.terminate()
                p.join(timeout=2)
                if p.is_alive():
                    p.kill()
        self.ps.clear()

    # ------------------------------------------------------------------ requests
    def _validate_prompt(self, prompt: str | list[int]) -> list[int]:
        """Token ids of a prompt that one prefill can take, or an exception.  The ONE check behind both
        ``add_request`` and the tensor-parallel driver's pre-validation (a request that would make a rank raise
        half-way through must fail before the workers are told)."""
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer loaded: pass prompts as lists of token ids")
            prompt = self.tokenizer.encode(prompt)
        if not prompt:
            raise ValueError("token_ids must contain at least one token")
        # one prefill step takes max_num_batched_tokens; longer prompts are prefilled in chunks (scheduler.py)
        limit = self.config.max_model_len
        if not getattr(self.config, "enable_chunked_prefill", False):
            limit = min(limit, self.config.max_num_batched_tokens)
        if self.config.num_kvcache_blocks > 0:   # (-1 before the KV cache is sized)
            limit = min(limit, self.config.num_kvcache_blocks * self.config.kvcache_block_size)
        if len(prompt) > limit:
            # beyond the reference, whose scheduler only notices when nothing can be scheduled any more
            raise ValueError(f"prompt of {len(prompt)} tokens exceeds what can be prefilled (max_model_len, KV pool; max_num_batched_tokens without chunked prefill): {limit}")
        return prompt

    # ------------------------------------------------------------------ tensor-parallel relay (spawned workers)
    def _relay(self, *cmd) -> None:
        """Driver of spawned workers: tell them to make the same call.  A worker that has died can never take part in
        the collectives of the step again -- that is an exception HERE, before this rank walks into an all-reduce alone
        (whose bounded wait would only give up after csrc/allreduce.hip's 20 s)."""
        if not self._is_driver:
            return
        dead = [i + 1 for i, p in enumerate(self.ps) if not p.is_alive()]
        if dead:
            raise RuntimeError(f"tensor-parallel worker process(es) {dead} have died (exit codes "
                               f"{[self.ps[i - 1].exitcode for i in dead]}): this engine cannot serve any more")
        self.model_runner.distributed_manager.broadcast_data(cmd, src=0)

    def _serve_relayed(self, cmd) -> None:
        """Worker side of ``_relay``: the same call on this rank's engine."""
        op = cmd[0]
        if op == "generate":
            self._generate_local(cmd[1], cmd[2], use_tqdm=False)
        elif op == "add_request":
            self._add_request_local(cmd[1], cmd[2])
        elif op == "step":
            self._step_local()
        else:
            raise ValueError(f"unknown relayed call {op!r}")

    def add_request(self, prompt: str | list[int], sampling_params: SamplingParams | None = None) -> int:
        if sampling_params is None:
            sampling_params = SamplingParams()
        prompt = self._validate_prompt(prompt)        # (tokenized on the driver: the workers get token ids)
        self._relay("add_request", prompt, sampling_params)
        return self._add_request_local(prompt, sampling_params)

    def _add_request_local(self, prompt: list[int], sampling_params: SamplingParams) -> int:
        seq = Sequence(prompt, sampling_params, block_size=self.config.kvcache_block_size)
        self.scheduler.add(seq)
        return seq.seq_id

    def step(self):
        """-> ([(seq_id, completion_token_ids) for finished], +prefill tokens | -decode seqs)."""
        self._relay("step")
        return self._step_local()

    def _step_local(self):
        sequences, is_prefill = self.scheduler.schedule()
        ex = getattr(self.model_runner, "inference_executor", None)
        if ex is not None:  # hint: with nothing waiting, the next step is a decode of (at most) this batch
            ex.speculate_next = not self.scheduler.waiting
        token_ids = self.model_runner.call("run", sequences, is_prefill)
        if token_ids is None:
            raise RuntimeError("ModelRunner returned no tokens on rank 0")
        # tokens this step computed (a prefix-cache hit or an earlier chunk is not recomputed), before postprocess
        # moves the counters
        num_tokens = sum(s.prefill_stop - s.num_cached_tokens for s in sequences) if is_prefill else -len(sequences)
        self.scheduler.postprocess(sequences, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids) for s in sequences if s.is_finished]
        return outputs, num_tokens

    def is_finished(self) -> bool:
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params=None, use_tqdm: bool = True):
        # the WHOLE list is validated (and tokenized) before anything is enqueued or any worker is told: a bad
        # prompt in the middle must not leave its predecessors queued in the scheduler, nor kill a worker
        if isinstance(sampling_params, list) and len(sampling_params) != len(prompts):
            raise ValueError("Length of sampling_params list must match prompts")
        prompts = [self._validate_prompt(p) for p in prompts]
        self._relay("generate", prompts, sampling_params)
        return self._generate_local(prompts, sampling_params, use_tqdm)

    def _generate_local(self, prompts, sampling_params, use_tqdm: bool):
        if sampling_params is None:
            sp_list = [SamplingParams() for _ in prompts]
        elif not isinstance(sampling_params, list):
            sp_list = [sampling_params] * len(prompts)
        else:
            if len(sampling_params) != len(prompts):
                raise ValueError("Length of sampling_params list must match prompts")
            sp_list = sampling_params
        pbar = None
        if use_tqdm and self.rank == 0:
            from tqdm.auto import tqdm

            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        ids = [self._add_request_local(p, sp) for p, sp in zip(prompts, sp_list)]   # (validated by generate())
        outputs: dict[int, list[int]] = {}
        prefill_tps = decode_tps = 0.0
        while not self.is_finished():
            t0 = perf_counter()
            finished, num_tokens = self._step_local()
            dt = perf_counter() - t0
            if num_tokens > 0:
                prefill_tps = num_tokens / dt
            else:
                decode_tps = -num_tokens / dt
            if pbar is not None:
                pbar.set_postfix({"Prefill": f"{int(prefill_tps)} token/s", "Decode": f"{int(decode_tps)} token/s"})
            for seq_id, toks in finished:
                outputs[seq_id] = toks
                if pbar is not None:
                    pbar.update(1)
        if pbar is not None:
            pbar.close()
        ordered = [outputs[i] for i in sorted(ids) if i in outputs]
        if self.tokenizer is not None:
            texts = self.tokenizer.batch_decode(ordered, skip_special_tokens=True, clean_up_tokenization_spaces=True)
        else:
            texts = [""] * len(ordered)
        return [{"text": t, "token_ids": toks} for t, toks in zip(texts, ordered)]
