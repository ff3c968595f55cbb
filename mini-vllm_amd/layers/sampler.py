"""Token sampler on the fused HIP kernel (replaces minivllm/sampling/sampler.py:12-110 and
sampling/functional.py as wired by the engine: temperature -> top-k -> top-p -> min-p ->
multinomial).  Penalties / typical / avoid-top-k are not reachable from ``SamplingParams`` in
the reference and are not built."""
from __future__ import annotations

import torch
from torch import nn

from .. import hip_ops

__all__ = ["Sampler"]


class Sampler(nn.Module):
    def __init__(self, seed: int | None = None) -> None:
        super().__init__()
        self.seed = 0 if seed is None else int(seed)
        self.calls = 0          # Philox offset: one fresh stream per call
        self.offset_dev = None  # optional device-side extra offset (hipGraph replay)

    def forward(self, logits, temperatures=None, top_ks=None, top_ps=None, min_ps=None, **unused):
        """logits [B,V] -> int64 [B].  temperature 0 = argmax (the reference's tensor-temperature
        path turns 0 into softmax(l/1e-8) + multinomial)."""
        if logits.dim() != 2:
            raise ValueError(f"logits must be 2D [batch_size, vocab_size], got {logits.shape}")
        b = logits.shape[0]
        dev = logits.device
        if temperatures is None:
            temperatures = torch.ones(b, device=dev)
        if top_ks is None:
            top_ks = torch.full((b,), -1, dtype=torch.int64, device=dev)
        if top_ps is None:
            top_ps = torch.ones(b, device=dev)
        if min_ps is None:
            min_ps = torch.zeros(b, device=dev)
        self.calls += 1
        return hip_ops.sample(logits, temperatures, top_ks, top_ps, min_ps, seed=self.seed,
                              offset=self.calls, offset_dev=self.offset_dev, return_stats=bool(unused.get("return_stats")))
