"""Engine configuration (mirror of minivllm/config.py:16-287: same fields, defaults and
validation errors).  One optional field is new: ``load_format`` ("auto" loads weights from
the model directory, "dummy" random-initialises at the architecture's true dimensions —
there are no checkpoints on the benchmark machines)."""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from pathlib import Path
from typing import ClassVar

from transformers import AutoConfig, PretrainedConfig

__all__ = ["Config"]


@dataclass
class Config:
    MIN_DEVICE_MEMORY_UTIL: ClassVar[float] = 0.1
    MAX_DEVICE_MEMORY_UTIL: ClassVar[float] = 1.0
    MIN_TENSOR_PARALLEL_SIZE: ClassVar[int] = 1
    MAX_TENSOR_PARALLEL_SIZE: ClassVar[int] = 8
    BLOCK_SIZE_DIVISOR: ClassVar[int] = 64
    DEFAULT_MAX_MODEL_LEN: ClassVar[int] = 4096

    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    device_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    use_buffered_page_attention: bool = False  # accepted for compatibility; one HIP kernel serves both
    hf_config: PretrainedConfig | None = None
    eos: int = -1
    kvcache_block_size: int = 64
    num_kvcache_blocks: int = -1
    trust_remote_code: bool = False
    dtype: str = "auto"
    seed: int | None = None
    load_format: str = "auto"
    # beyond the reference (opt-in): prompts longer than max_num_batched_tokens are prefilled in chunks
    # (scheduler.py); max_num_batched_tokens may then be smaller than max_model_len
    enable_chunked_prefill: bool = False
    # beyond the reference (opt-in): a prefill step also carries one token of every running sequence (scheduler.py),
    # so decoding does not stall while prompts are prefilled
    enable_mixed_batches: bool = False

    @property
    def gpu_memory_utilization(self) -> float:
        return self.device_memory_utilization

    @gpu_memory_utilization.setter
    def gpu_memory_utilization(self, value: float) -> None:
        self.device_memory_utilization = value

    # -- validation, in the reference's order (config.py:112-124) ------------------------
    def __post_init__(self) -> None:
        self._check_model_path()
        if self.dtype not in ("auto", "float16", "bfloat16", "float32"):
            raise ValueError(
                f"dtype must be one of ('auto', 'float16', 'bfloat16', 'float32'), got {self.dtype!r}")
        if self.load_format not in ("auto", "dummy"):
            raise ValueError(f"load_format must be 'auto' or 'dummy', got {self.load_format!r}")
        if self.eos < -1:
            raise ValueError(f"eos must be -1 (auto-detect) or a non-negative token ID, got {self.eos}")
        if self.num_kvcache_blocks < -1 or self.num_kvcache_blocks == 0:
            raise ValueError(
                f"num_kvcache_blocks must be -1 (auto) or a positive integer, got {self.num_kvcache_blocks}")
        lo, hi = self.MIN_DEVICE_MEMORY_UTIL, self.MAX_DEVICE_MEMORY_UTIL
        if not lo <= self.device_memory_utilization <= hi:
            raise ValueError(
                f"device_memory_utilization must be between {lo} and {hi}, got "
                f"{self.device_memory_utilization}. Values outside this range may cause OOM errors "
                f"or underutilization.")
        if self.kvcache_block_size % self.BLOCK_SIZE_DIVISOR != 0:
            raise ValueError(
                f"kvcache_block_size must be divisible by {self.BLOCK_SIZE_DIVISOR} for optimal "
                f"performance, got {self.kvcache_block_size}. Common values: 64, 128, 256.")
        lo, hi = self.MIN_TENSOR_PARALLEL_SIZE, self.MAX_TENSOR_PARALLEL_SIZE
        if not lo <= self.tensor_parallel_size <= hi:
            raise ValueError(
                f"tensor_parallel_size must be between {lo} and {hi}, got {self.tensor_parallel_size}. "
                f"Higher values may not provide additional benefits and consume more memory.")
        for name in ("max_num_batched_tokens", "max_num_seqs", "max_model_len"):
            if getattr(self, name) <= 0:
                raise ValueError(f"{name} must be positive, got {getattr(self, name)}")
        self._load_hf_config()
        self._clamp_max_model_len()
        if self.max_num_batched_tokens < self.max_model_len and not self.enable_chunked_prefill:
            raise ValueError(
                f"max_num_batched_tokens ({self.max_num_batched_tokens}) must be >= max_model_len "
                f"({self.max_model_len}) to accommodate the full context length in a single batch.")

    def _check_model_path(self) -> None:
        if Path(self.model).is_dir():
            return
        if not self.model.startswith(("/", "./", "../", "~")):
            return  # a hub id; resolved (or not) by AutoConfig below
        raise ValueError(
            f"Model path '{self.model}' is not a valid directory. Please ensure the model is "
            f"properly downloaded and accessible.")

    def _load_hf_config(self) -> None:
        if self.hf_config is not None:
            return
        try:
            self.hf_config = AutoConfig.from_pretrained(
                self.model, trust_remote_code=self.trust_remote_code, local_files_only=True)
        except OSError:
            # the reference retries online (config.py:222-227); kept for API parity — there is
            # no network on the benchmark machines, so this raises the same OSError again
            self.hf_config = AutoConfig.from_pretrained(self.model, trust_remote_code=self.trust_remote_code)
        except Exception as e:
            raise ValueError(
                f"Failed to load HuggingFace model configuration from {self.model}: {e}. Please "
                f"ensure the model is a valid HuggingFace model or local directory.") from e

    def _clamp_max_model_len(self) -> None:
        if self.hf_config is None:
            return
        model_max = getattr(self.hf_config, "max_position_embeddings", self.DEFAULT_MAX_MODEL_LEN)
        if (self.max_num_batched_tokens < self.max_model_len and model_max < self.max_model_len
                and not self.enable_chunked_prefill):
            new_len = min(model_max, self.max_num_batched_tokens)
            warnings.warn(
                f"auto-adjusting max_model_len from {self.max_model_len} to {new_len} "
                f"(model/batch constraints).", UserWarning, stacklevel=2)
            self.max_model_len = new_len

    def __repr__(self) -> str:
        names = ("model", "max_num_batched_tokens", "max_num_seqs", "max_model_len",
                 "device_memory_utilization", "tensor_parallel_size", "enforce_eager",
                 "kvcache_block_size", "num_kvcache_blocks", "eos")
        return "Config(" + ", ".join(f"{n}={getattr(self, n, None)!r}" for n in names) + ")"
