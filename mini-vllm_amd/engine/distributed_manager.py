"""Tensor-parallel communicator bootstrap (mirror of minivllm/engine/distributed_manager.py):
one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm) for the
data plane (C1-C3) and a small gloo group on the host for the control plane (C4).

Unlike the reference this runs BEFORE the model is built and registers the
rank/size with ``parallel_state`` so that every layer shards itself."""
from __future__ import annotations

import logging
import os
from datetime import timedelta
from typing import Any

import torch
import torch.distributed as dist

from ..config import Config
from .. import parallel_state

logger = logging.getLogger(__name__)
__all__ = ["DistributedManager"]


class DistributedManager:
    def __init__(self, config: Config, rank: int, events=None) -> None:
        self.config = config
        self.rank = rank
        self.world_size = config.tensor_parallel_size
        self.events = events  # kept for signature compatibility; the reference never uses them either
        self.is_distributed = self.world_size > 1
        self.backend: str | None = None
        self.control_group = None
        self._initialized = False
        self._owns_group = False

    def initialize(self) -> None:
        if self._initialized:
            return
        if self.is_distributed:
            self._setup_distributed_backend()
            self._validate_setup()
        parallel_state.init_tensor_parallel(self.rank if self.is_distributed else 0,
                                            self.world_size if self.is_distributed else 1)
        if self.is_distributed and torch.cuda.is_available():
            from ..custom_allreduce import maybe_create

            # the device this rank will run on (ModelManager._setup_device makes the same choice)
            local = int(os.environ.get("LOCAL_RANK", self.rank)) % torch.cuda.device_count()
            torch.cuda.set_device(local)
            parallel_state.set_custom_all_reduce(
                maybe_create(self.rank, self.world_size, torch.device("cuda", local)))
        self._initialized = True

    def _setup_distributed_backend(self) -> None:
        self.backend = os.environ.get("MINIVLLM_TP_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if self.backend == "nccl":
                local = int(os.environ.get("LOCAL_RANK", self.rank)) % torch.cuda.device_count()
                torch.cuda.set_device(local)
                kw["device_id"] = torch.device("cuda", local)
            try:
                dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size,
                                        timeout=timedelta(minutes=10), **kw)
            except RuntimeError as e:
                raise RuntimeError(f"Failed to setup distributed backend: {e}") from e
            self._owns_group = True
        elif dist.get_world_size() != self.world_size:
            raise RuntimeError(f"existing process group has world size {dist.get_world_size()}, "
                               f"tensor_parallel_size is {self.world_size}")
        if self.backend != "gloo":
            self.control_group = dist.new_group(backend="gloo")

    def _validate_setup(self) -> None:
        t = torch.ones(1) * self.rank
        if self.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        expected = sum(range(self.world_size))
        if abs(t.item() - expected) > 1e-6:
            raise RuntimeError(f"Distributed test failed: got {t.item()}, expected {expected}")

    def synchronize(self) -> None:
        if self.is_distributed and dist.is_initialized():
            dist.barrier()

    def broadcast_data(self, data: Any, src: int = 0) -> Any:
        """Control-plane object broadcast (host side, gloo): never inside a hipGraph."""
        if not self.is_distributed:
            return data
        box = [data if self.rank == src else None]
        try:
            dist.broadcast_object_list(box, src=src, group=self.control_group)
        except Exception as e:
            raise RuntimeError(f"Broadcast failed: {e}") from e
        return box[0]

    def cleanup(self) -> None:
        if not self._initialized:
            return
        try:
            if self.is_distributed and dist.is_initialized() and self._owns_group:
                dist.barrier()
                dist.destroy_process_group()
        except Exception as e:
            logger.warning("Rank %d: Error during distributed cleanup: %s", self.rank, e)
        parallel_state.reset_tensor_parallel()
        self._initialized = False
