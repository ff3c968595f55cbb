"""Per-GPU dispatch boundary (mirror of minivllm/engine/model_runner.py:30-266): the engine only
ever calls ``call("run", seqs, is_prefill) -> list[int]`` and ``call("exit")``.

Execution is SPMD: every tensor-parallel rank holds the same scheduler state and calls ``run`` with
the same batch, so — unlike the reference — no per-step pickle of ``Sequence`` objects crosses a
process boundary; only ``generate`` requests do (see LLMEngine).  Worker errors are raised, not
swallowed."""
from __future__ import annotations

import logging
import os
from typing import Any

from ..config import Config
from .distributed_manager import DistributedManager
from .inference_executor import InferenceExecutor
from .model_manager import ModelManager
from .sequence import Sequence

logger = logging.getLogger(__name__)
__all__ = ["ModelRunner"]


class ModelRunner:
    def __init__(self, config: Config, rank: int, event=None) -> None:
        os.environ["RANK"] = str(rank)
        os.environ.setdefault("LOCAL_RANK", str(rank))
        os.environ["WORLD_SIZE"] = str(config.tensor_parallel_size)
        self.config = config
        self.rank = rank
        self.events = event
        self.world_size = config.tensor_parallel_size
        self.model_manager = ModelManager(config)
        self.distributed_manager = DistributedManager(config, rank, event)
        self.inference_executor: InferenceExecutor | None = None
        self._initialize()
        if self.world_size > 1:
            self.distributed_manager.synchronize()

    def _initialize(self) -> None:
        try:
            self.distributed_manager.initialize()   # communicator first: layers shard at construction
            self.model_manager.initialize()
            self.inference_executor = InferenceExecutor(self.config, self.model_manager.model,
                                                        self.model_manager.device)
            self.inference_executor.initialize(self.config.max_num_batched_tokens, self.config.max_num_seqs)
            self.inference_executor.capture_device_graphs(self.config.max_num_seqs)  # every rank, not only 0
        except Exception:
            logger.exception("Rank %d: ModelRunner initialization failed", self.rank)
            self.exit()
            raise

    def call(self, method_name: str, *args: Any, **kwargs: Any) -> Any:
        if not hasattr(self, method_name):
            raise AttributeError(f"ModelRunner has no method '{method_name}'")
        return getattr(self, method_name)(*args, **kwargs)

    def run(self, sequences: list[Sequence], is_prefill: bool) -> list[int]:
        if self.inference_executor is None:
            raise RuntimeError("Inference executor not initialized")
        _, tokens = self.inference_executor.execute_batch(sequences, is_prefill)
        return tokens

    def exit(self) -> None:
        if self.inference_executor is not None:
            self.inference_executor.cleanup()
            self.inference_executor = None
        self.distributed_manager.cleanup()
        self.model_manager.cleanup()
