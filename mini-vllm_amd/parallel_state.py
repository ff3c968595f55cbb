"""Tensor-parallel process state: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI).  Set ONCE, before the model is built — the reference derives tp_size from
``dist.is_initialized()`` while constructing layers but only initialises the process group
afterwards, so its TP>1 silently builds replicated models."""
from __future__ import annotations

import torch
import torch.distributed as dist

_state = {"rank": 0, "size": 1, "group": None, "custom_ar": None}


def init_tensor_parallel(rank: int, size: int, group=None) -> None:
    if size < 1 or not 0 <= rank < size:
        raise ValueError(f"invalid tensor-parallel rank/size {rank}/{size}")
    _state.update(rank=rank, size=size, group=group, custom_ar=None)


def set_custom_all_reduce(ar) -> None:
    """The validated peer-to-peer kernel (custom_allreduce.CustomAllReduce) for small 16-bit messages."""
    _state["custom_ar"] = ar


def get_custom_all_reduce():
    return _state["custom_ar"]


def reset_tensor_parallel() -> None:
    ar = _state.get("custom_ar")
    if ar is not None:
        ar.close()
    _state.update(rank=0, size=1, group=None, custom_ar=None)


def get_tensor_parallel_rank() -> int:
    return _state["rank"]


def get_tensor_parallel_world_size() -> int:
    return _state["size"]


def get_tensor_parallel_group():
    return _state["group"]


def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place SUM over the TP group (C1/C2: after o_proj, down_proj and the embedding): decode-sized
    16-bit messages on the peer-to-peer kernel when it passed its validation, everything else on RCCL."""
    if _state["size"] > 1:
        ar = _state["custom_ar"]
        if ar is not None and ar.accepts(x):
            return ar.all_reduce(x)
        dist.all_reduce(x, group=_state["group"])
    return x


def tp_all_reduce_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """``tp_all_reduce(x)`` followed by the fused residual-add + RMSNorm -> (normed, residual), residual
    updated in place.  One launch on the peer-to-peer kernel when it takes the shape,
    otherwise the two steps."""
    from . import hip_ops

    ar = _state["custom_ar"]
    if (ar is not None and ar.accepts_rows(x) and residual.shape == x.shape and residual.dtype == x.dtype
            and residual.is_contiguous()):
        return ar.all_reduce_add_rmsnorm(x, residual, weight, eps)
    return hip_ops.add_rmsnorm(tp_all_reduce(x), residual, weight, eps, inplace=True)


def tp_all_gather_stack(x: torch.Tensor) -> torch.Tensor:
    """``[tp, *x.shape]``: every rank's ``x``, rank-major, on every rank."""
    n = _state["size"]
    if n == 1:
        return x.unsqueeze(0)
    if x.is_cuda and dist.get_backend(_state["group"]) == "gloo":
        # gloo has no device all-gather; only the 2-process single-GPU TP test takes this path
        out = torch.zeros((n,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        out[_state["rank"]] = x
        dist.all_reduce(out, group=_state["group"])
        return out
    x = x.contiguous()
    flat = torch.empty((n * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(flat, x, group=_state["group"])  # rank-major along dim 0
    return flat.view((n,) + tuple(x.shape))


def tp_all_gather_last_dim(x: torch.Tensor) -> torch.Tensor:
    """Concatenate vocab shards along the last dim on EVERY rank (C3).  The reference gathers to
    rank 0 only (embed_head.py:219-229); here all ranks run the same scheduler and sampler, so
    each needs the full row and no per-step control broadcast exists."""
    n = _state["size"]
    if n == 1:
        return x
    return tp_all_gather_stack(x).movedim(0, -2).reshape(*x.shape[:-1], n * x.shape[-1])


def tp_merge_greedy(local_tokens: torch.Tensor, local_values: torch.Tensor, vocab_per_rank: int) -> torch.Tensor:
    """Greedy decoding without the full-logit gather: each rank brings the argmax of its
    vocab shard (shard-local index, lowest on ties) and its value; the row's token is the global index
    of the largest value, the lowest global index on ties -- exactly ``argmax`` of the gathered row.
    One all-gather of B x 16 bytes per rank instead of B x V/tp logits."""
    gidx = local_tokens + _state["rank"] * vocab_per_rank
    pair = torch.stack((local_values.double(), gidx.double()), dim=-1)      # exact for 16/32-bit logits
    allp = tp_all_gather_stack(pair)                                          # [tp, B, 2]
    vals, idx = allp[..., 0], allp[..., 1]
    best = vals.max(dim=0, keepdim=True).values
    return torch.where(vals == best, idx, torch.full_like(idx, float("inf"))).min(dim=0).values.long()


def tp_greedy_tokens(logits_local: torch.Tensor, local_tokens: torch.Tensor) -> torch.Tensor:
    """Tokens of an all-greedy batch from every rank's shard-local argmax.  With the peer-to-peer all-reduce
    available the exchange is three launches of this library and no backend collective (the whole decode step
    then captures into a graph without RCCL inside); otherwise ``tp_merge_greedy`` over the backend."""
    n, vpr = logits_local.shape
    ar, world = _state["custom_ar"], _state["size"]
    if (ar is not None and logits_local.dtype in (torch.bfloat16, torch.float16) and world * vpr <= 1 << 24):
        from . import hip_ops

        buf = hip_ops.greedy_pack(logits_local, local_tokens, _state["rank"], world, n + (n & 1))
        flat = buf.view(-1)
        if ar.accepts(flat):
            ar.all_reduce(flat)                      # sum with zeros = exact gather
            return hip_ops.greedy_merge(buf, n)
    vals = logits_local.gather(1, local_tokens.unsqueeze(1)).squeeze(1)
    return tp_merge_greedy(local_tokens, vals, vpr)


def tp_candidate_tokens(logits_local: torch.Tensor, sample_fn, temperatures: torch.Tensor | None = None) -> torch.Tensor:
    """Sampled tokens without the full-logit gather (sampled rows): valid for batches whose rows are
    greedy or have 1 <= top_k <= hip_ops.TOPC.  Every rank packs the 64 largest logits of its vocab shard per row
    (+ global indices) into its slot of a zeroed buffer, the all-reduce sums the slots (x + 0: an exact gather,
    on the peer-to-peer kernel when it is up: no backend collective), ``sample_fn(cand_logits [B, tp*64]) -> int64
    positions`` runs the ordinary sampler on the candidates — every rank the same Philox stream, so every rank
    the same token — and the candidate's global index is the token.  512 bytes per row and rank on the wire
    instead of 2 * V / tp."""
    from . import hip_ops

    n = logits_local.shape[0]
    buf = hip_ops.topc_pack(logits_local, _state["rank"], _state["size"], n)
    tp_all_reduce(buf.view(-1))
    cand_logits, cand_idx = hip_ops.topc_unpack(buf, n)
    pos = sample_fn(cand_logits)
    # ``temperatures`` (fp32, one per row): greedy rows take the lowest token id among the maxima, like TP = 1 and the
    # all-greedy path do -- candidate order is not index order (mv_topc_pick)
    return hip_ops.topc_pick(cand_logits, cand_idx, pos, temperatures)


def tp_mass_tokens(logits_local: torch.Tensor, sample_stats_fn, temperatures: torch.Tensor, seed: int, offset: int,
                   offset_dev=None) -> torch.Tensor:
    """Tokens of rows sampled from the WHOLE vocabulary (top_k off, top_p = 1, min_p = 0: the default SamplingParams)
    or greedy, without the logit gather: an exact two-stage draw.  ``sample_stats_fn(logits_local) ->
    (local tokens, fp32 [B, 2] {shard maximum, shard mass})`` is the ordinary sampler on this rank's shard; the ranks
    exchange 16 bytes per row through the slot scheme (peer-to-peer all-reduce when it is up: no backend collective)
    and pick the same rank with probability proportional to its mass from a second uniform of the same Philox counter
    (``seed`` / ``offset`` / ``offset_dev`` = what the sampler call used), then take that rank's token
    (mv_mass_pack / mv_mass_merge): the softmax of the full row."""
    from . import hip_ops

    n, vpr = logits_local.shape
    local, stats = sample_stats_fn(logits_local)
    buf = hip_ops.mass_pack(stats, local, _state["rank"], _state["size"], vpr, n, logits_local.dtype)
    tp_all_reduce(buf.view(-1))
    return hip_ops.mass_merge(buf, n, temperatures, seed, offset, offset_dev)


def tp_nucleus_tokens(logits_local: torch.Tensor, sample_stats_fn, temperatures: torch.Tensor, top_ps: torch.Tensor,
                      min_ps: torch.Tensor, seed: int, offset: int, offset_dev=None) -> torch.Tensor:
    """Tokens of rows with a nucleus (top_p < 1) and / or a min-p filter and top_k off (plus greedy / unfiltered rows
    in the same batch) without the logit gather.  The ranks find the edge of the kept set together:
    the row maximum (MAX all-reduce of B floats), then a two-level radix walk over the 16-bit order keys below it with
    one all-reduced 256-bin histogram of fixed-point masses per level (mv_shard_key_hist / mv_nucleus_pick: integer
    sums, so every rank derives the same floor); every rank masks its shard below the floor and the batch finishes as
    ``tp_mass_tokens`` does -- ``sample_stats_fn(masked shard)`` must run the sampler with its filters OFF."""
    from . import hip_ops

    n, vpr = logits_local.shape
    group = _state["group"]
    t32 = temperatures.float().contiguous()
    p32 = top_ps.float().contiguous()
    m32 = min_ps.float().contiguous()
    row_max = logits_local.amax(dim=1).float()
    dist.all_reduce(row_max, op=dist.ReduceOp.MAX, group=group)
    hist = hip_ops.shard_key_hist(logits_local, row_max, t32, None)
    dist.all_reduce(hist, group=group)
    state, prefix = hip_ops.nucleus_pick(hist, 0, p32, m32, t32, row_max, logits_local.dtype)
    hist = hip_ops.shard_key_hist(logits_local, row_max, t32, prefix)
    dist.all_reduce(hist, group=group)
    floor = hip_ops.nucleus_pick(hist, 1, p32, m32, t32, row_max, logits_local.dtype, state, prefix)
    masked = hip_ops.mask_below(logits_local, floor)
    local, stats = sample_stats_fn(masked)
    buf = hip_ops.mass_pack(stats, local, _state["rank"], _state["size"], vpr, n, logits_local.dtype)
    tp_all_reduce(buf.view(-1))
    return hip_ops.mass_merge(buf, n, t32, seed, offset, offset_dev)


def decode_step_is_native(dtype: torch.dtype, hidden_size: int, vocab_size: int) -> bool:
    """True when an all-greedy decode step of a model with these dimensions needs no backend collective: every
    all-reduce and the greedy exchange run on this library's peer-to-peer kernel (-> the step captures into a
    graph whatever the backend is)."""
    ar, world = _state["custom_ar"], _state["size"]
    return (ar is not None and ar.ready and dtype in (torch.bfloat16, torch.float16)
            and hidden_size % (8 * world) == 0 and vocab_size <= 1 << 24)


def divide(numerator: int, denominator: int) -> int:
    if numerator % denominator != 0:
        raise ValueError(f"{numerator} is not divisible by {denominator}")
    return numerator // denominator
