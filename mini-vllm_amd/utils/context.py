"""Per-step side channel from the executor to the attention layers (mirror of
minivllm/utils/context.py:30-143).  These tensors are the implicit argument list of the
HIP attention kernels."""
from __future__ import annotations

from contextvars import ContextVar
from dataclasses import dataclass

import torch

__all__ = ["Context", "get_context", "set_context", "reset_context"]


@dataclass
class Context:
    is_prefill: bool = False
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    cum_seqlens_q: torch.Tensor | None = None   # int32 [n+1], prefill only
    cum_seqlens_k: torch.Tensor | None = None   # int32 [n+1], prefill only
    slot_mapping: torch.Tensor | None = None    # int32 [tokens]; -1 = skip
    context_lens: torch.Tensor | None = None    # int32 [B], decode only
    block_tables: torch.Tensor | None = None    # int32 [B, blocks], -1 padded
    # build-only hint (not in the reference): some sequence of this prefill batch has a cached
    # prefix, so keys must be read from the paged cache instead of the contiguous k/v
    prefix_cached: bool = False
    # build-only (mixed batches): the first ``decode_rows`` sequences of this prefill batch contribute ONE token each
    # on top of a cached prefix — decode rows; attention serves them with the paged-decode kernel (their lengths in
    # ``context_lens``, their tables the first rows of ``block_tables``) and the rest with the prefill kernel
    # (``rest_*``: cumulative lengths re-based to the first non-decode sequence)
    decode_rows: int = 0
    rest_cum_seqlens_q: torch.Tensor | None = None
    rest_cum_seqlens_k: torch.Tensor | None = None
    rest_max_seqlen_q: int = 0
    rest_max_seqlen_k: int = 0


_CURRENT: ContextVar[Context | None] = ContextVar("minivllm_amd_context", default=None)


def get_context() -> Context:
    ctx = _CURRENT.get()
    return ctx if ctx is not None else Context()


def set_context(is_prefill: bool, max_seqlen_q: int = 0, max_seqlen_k: int = 0,
                cum_seqlens_q=None, cum_seqlens_k=None, slot_mapping=None, context_lens=None,
                block_tables=None, prefix_cached: bool = False, decode_rows: int = 0, rest_cum_seqlens_q=None,
                rest_cum_seqlens_k=None, rest_max_seqlen_q: int = 0, rest_max_seqlen_k: int = 0) -> None:
    _CURRENT.set(Context(is_prefill, max_seqlen_q, max_seqlen_k, cum_seqlens_q, cum_seqlens_k,
                         slot_mapping, context_lens, block_tables, prefix_cached, decode_rows,
                         rest_cum_seqlens_q, rest_cum_seqlens_k, rest_max_seqlen_q, rest_max_seqlen_k))


def reset_context() -> None:
    _CURRENT.set(Context())
