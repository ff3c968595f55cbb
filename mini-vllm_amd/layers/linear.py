"""Tensor-parallel linear layers (mirror of minivllm/models/layers/linear.py:36-319).
Prefill GEMMs stay on hipBLASLt through ``F.linear``.  Decode steps go to the hand-written kernels:
  * up to ``SMALL_M_MAX`` tokens every projection is a pure HBM stream of its weights
    (csrc/linear_small_m.hip; on the fragment-ordered copy below when there is one — whole-line loads —
    else on the row-major weights);
  * up to ``MID_M_MAX`` tokens (the headline batch, 256) the projections are bound by how fast a CU takes bytes
    in; csrc/linear_mid_m.hip streams a fragment-ordered copy of the weight (``pack_for_decode``), with
    SiluAndMul in the gate_up epilogue and, for the row-parallel layers, a K split whose fp32 partial sums are
    added up by the residual-add + RMSNorm kernel behind them (``PartialSum``).  Which tile / split runs for a
    shape comes from a table measured on MI355X (tuned/mid_m_gfx950.json, tools/tune_mid.py) or, for shapes
    that were not tuned, from a cost model; entries where the library GEMM measured faster keep it.
What else matters here is which slice of a checkpoint tensor each rank owns (the ``weight_loader`` hooks) and
the all-reduce after row-parallel GEMMs."""
from __future__ import annotations

import json
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops
from ..parallel_state import (divide, get_tensor_parallel_rank, get_tensor_parallel_world_size,
                              tp_all_reduce)
from ..utils.loader import materialize

__all__ = ["LinearBase", "ReplicatedLinear", "ColumnParallelLinear", "MergedColumnParallelLinear",
           "QKVParallelLinear", "RowParallelLinear", "divide", "linear", "small_m", "PartialSum", "mid_plan",
           "tuned_plan", "mid_m"]

# rows up to which the weight-streaming kernel beats the library GEMM (tools/bench_linear.py); 0 disables
SMALL_M_MAX = int(os.environ.get("MV_LINEAR_SMALL_M_MAX", "16"))


def small_m(x, weight) -> bool:
    """Does this projection go to the weight-streaming kernel?  Up to SMALL_M_MAX rows always; up to twice that
    for the shapes where it still measured faster than hipBLASLt (tools/dbg/xlds_bench.sh, M = 32): the small
    layers (qkv / o of Qwen3-1.7B: 8.3 / 7.2 vs 11.8 / 11.2 us), the LM head (123 vs 149 us) and very deep K
    (Qwen2-7B down: 48 vs 108 us) -- not the 25-270 MB gate_up matrices."""
    if not x.is_cuda:
        return False
    rows = x.numel() // x.shape[-1]
    if not 0 < rows <= 2 * SMALL_M_MAX or not hip_ops.linear_small_m_ok(x, weight):
        return False
    if rows <= SMALL_M_MAX:
        return True
    n, k = weight.shape
    return n * k <= 10_000_000 or n >= 65536 or k >= 16384


# ---- decode batches of SMALL_M_MAX+1 .. MID_M_MAX rows: csrc/linear_mid_m.hip --------------------------------
MID_M_MAX = int(os.environ.get("MV_LINEAR_MID_M_MAX", "512"))   # 0 disables (everything above small-M on the library)
# workgroup tiles of mv_linear_mid_m (ids of the table in csrc/linear_mid_m.hip): id -> (BM, BN, takes the gated form)
MID_CFGS = {1: (128, 96, True), 2: (256, 48, False), 3: (128, 32, True), 4: (64, 32, False), 5: (256, 64, True),
            6: (128, 128, True), 7: (256, 160, True), 8: (256, 128, True), 9: (128, 48, False), 10: (64, 64, True),
            11: (64, 64, True), 12: (64, 64, True), 13: (128, 96, True), 14: (128, 96, True), 15: (64, 64, True),
            16: (128, 96, True), 17: (128, 128, True), 18: (256, 160, True), 19: (256, 128, True),
            # 20..27: tiles 14 / 11 / 15 / 2 / 16 / 4 / 17 / 9 with their LDS reads issued between the MFMAs (round 3)
            20: (128, 96, True), 21: (64, 64, True), 22: (64, 64, True), 23: (256, 48, False), 24: (128, 96, True),
            25: (64, 32, False), 26: (128, 128, True), 27: (128, 48, False)}
_TUNED_PATH = os.environ.get("MV_MID_TUNED") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuned", "mid_m_gfx950.json")
_TUNED_MS = (64, 128, 192, 256, 384, 512)
try:
    with open(_TUNED_PATH) as _f:
        _TUNED = json.load(_f).get("plans", {})
except (OSError, ValueError):
    _TUNED = {}


class PartialSum:
    """fp32 partial sums ``[splits, rows, cols]`` of a K-split row-parallel projection: the residual-add +
    RMSNorm behind it adds them up (split order, then ONE rounding to the activation dtype — the GEMM's)."""
    __slots__ = ("data", "dtype")

    def __init__(self, data, dtype):
        self.data, self.dtype = data, dtype

    @property
    def shape(self):
        return self.data.shape[1:]

    def to_tensor(self):
        """The projection output itself (only needed where no fused consumer follows)."""
        return self.data.sum(0).to(self.dtype)


def _model_cost(m, n, k, bm, bn, gated, splits):
    """Estimated microseconds of one mv_linear_mid_m launch (fit of the M = 256 sweeps, profiles/r02_mid_m_*):
    a fixed 5 us + what one workgroup takes in at ~75 GB/s, rounds of 256 workgroups, plus the partial slabs."""
    bno = bn // 2 if gated else bn
    wgs = -(-m // bm) * -(-n // bno) * splits
    rounds = -(-wgs // 256)
    t = 5.0 + rounds * (bm + bn) * (k // splits) * 2 / 75e3 * (1.25 if rounds > 1 else 1.0)
    t = max(t, 2.5 + (2 if gated else 1) * n * k * 2 / 5.0e6)      # the weight stream itself at ~5 TB/s
    if wgs < 96:
        t *= 1.5                                                    # most of the chip idle
    if splits > 1:
        t += 1.5 + splits * m * n * 4 / 3.0e6
    return t


def mid_plan(kind: str, m: int, n: int, k: int):
    """-> (cfg, splits): tile id of mv_linear_mid_m (0 = keep the library GEMM) and K split for a projection of
    ``kind`` 'col' (plain), 'gated' (gate_up + SiluAndMul, n = intermediate size) or 'row' (followed by the
    residual-add + RMSNorm, so a K split is allowed).  Measured table first, cost model for untuned shapes."""
    mt = next((t for t in _TUNED_MS if t >= m), None)
    hit = _TUNED.get(f"{kind}:{mt}:{n}:{k}") if mt else None
    if hit is not None:
        return int(hit["cfg"]), int(hit["splits"])
    best, best_t = (0, 1), float("inf")
    for cid, (bm, bn, gated_ok) in MID_CFGS.items():
        if kind == "gated" and not gated_ok:
            continue
        for splits in (((1, 2, 4, 8) if k >= 8192 else (1, 2, 4)) if kind == "row" else (1,)):
            if (k // 64) % splits:
                continue
            t = _model_cost(m, n, k, bm, bn, kind == "gated", splits)
            if t < best_t:
                best, best_t = (cid, splits), t
    return best


def tuned_plan(kind: str, m: int, n: int, k: int):
    """The measured (cfg, splits) of this exact shape, or None when the table does not hold it (the LM head only
    leaves the library GEMM on a measurement, never on the cost model)."""
    mt = next((t for t in _TUNED_MS if t >= m), None)
    hit = _TUNED.get(f"{kind}:{mt}:{n}:{k}") if mt else None
    return None if hit is None else (int(hit["cfg"]), int(hit["splits"]))


def mid_m(x, packed) -> bool:
    """Does this projection go to the decode-batch kernel (rows above the small-M range, a packed weight)?"""
    if packed is None or not x.is_cuda or MID_M_MAX <= 0:
        return False
    rows = x.numel() // x.shape[-1]
    return SMALL_M_MAX < rows <= MID_M_MAX


def linear(x, weight, bias=None, packed=None, kind: str = "col"):
    """``F.linear`` semantics; decode-sized inputs take the HIP kernels.  ``kind == 'row'`` may return a
    ``PartialSum`` (the caller hands it to the RMSNorm behind the projection)."""
    if small_m(x, weight):
        return hip_ops.linear_small_m(x, weight, bias, packed=packed)
    if mid_m(x, packed):
        n, k = weight.shape
        if hip_ops.linear_mid_m_ok(x, n):
            rows = x.numel() // k
            cfg, splits = mid_plan(kind if bias is None else "col", rows, n, k)
            if cfg and splits > 1:
                return PartialSum(hip_ops.linear_mid_m_partial(x, packed, n, splits, cfg), x.dtype)
            if cfg:
                return hip_ops.linear_mid_m(x, packed, n, bias, cfg)
    return F.linear(x, weight, bias)


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.input_size, self.output_size, self.tp_dim = input_size, output_size, tp_dim
        self.tp_rank = get_tensor_parallel_rank()
        self.tp_size = get_tensor_parallel_world_size()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        self.weight.weight_loader = self.weight_loader
        self.weight_packed = None   # fragment-ordered copy for mv_linear_mid_m (pack_for_decode)
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def extra_repr(self) -> str:
        return (f"input_size={self.input_size}, output_size={self.output_size}, "
                f"bias={self.bias is not None}, tp_dim={self.tp_dim}, tp_size={self.tp_size}")

    def forward(self, x):
        raise NotImplementedError("Subclasses must implement forward method")

    def pack_for_decode(self) -> bool:
        """(Re)build the fragment-ordered copy of the weight that decode steps stream (csrc/linear_small_m.hip up
        to 16-32 rows, csrc/linear_mid_m.hip up to MID_M_MAX).  Call after the weights are loaded / whenever they
        change.  Costs one more copy of the weight in HBM; prefill keeps reading the row-major one through the
        library GEMM."""
        w = self.weight
        ok = (w.is_cuda and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] % 64 == 0
              and w.shape[0] % 4 == 0)
        self.weight_packed = hip_ops.pack_weight(w.data) if ok else None
        return ok

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        raise NotImplementedError("Subclasses must implement weight_loader method")


class ReplicatedLinear(LinearBase):
    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        param.data.copy_(materialize(loaded_weight))

    def forward(self, x):
        return linear(x, self.weight, self.bias, self.weight_packed)


class ColumnParallelLinear(LinearBase):
    """Output features split over ranks: rank r owns rows [r*n/tp, (r+1)*n/tp) of the weight."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, divide(output_size, get_tensor_parallel_world_size()), bias, 0)

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        shard = param.data.size(0)
        param.data.copy_(materialize(loaded_weight.narrow(0, self.tp_rank * shard, shard)))

    def forward(self, x):
        return linear(x, self.weight, self.bias, self.weight_packed)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices fused along the output dim (gate|up)."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: int = 0, *args, **kwargs):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        piece = loaded_weight.chunk(self.tp_size, 0)[self.tp_rank]
        param.data.narrow(0, offset, size).copy_(materialize(piece))


class QKVParallelLinear(ColumnParallelLinear):
    """q|k|v fused; each rank owns its q heads and its kv heads, stored [q_r | k_r | v_r]."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        tp = get_tensor_parallel_world_size()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp)
        self.num_kv_heads = divide(total_num_kv_heads, tp)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: str | None = None, *args, **kwargs):
        if loaded_shard_id is None:
            raise ValueError("loaded_shard_id must be provided for QKVParallelLinear")
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        try:
            offset, size = {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}[loaded_shard_id]
        except KeyError:
            raise AssertionError(f"Invalid shard_id: {loaded_shard_id}") from None
        piece = loaded_weight.chunk(self.tp_size, 0)[self.tp_rank]
        param.data.narrow(0, offset, size).copy_(materialize(piece))

    def extra_repr(self) -> str:
        return (f"hidden_size={self.input_size}, head_size={self.head_size}, num_heads={self.num_heads}, "
                f"num_kv_heads={self.num_kv_heads}, bias={self.bias is not None}, tp_size={self.tp_size}")


class RowParallelLinear(LinearBase):
    """Input features split over ranks; partial products are summed with one all-reduce.
    The bias lives on rank 0 only (linear.py:296-316)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(divide(input_size, get_tensor_parallel_world_size()), output_size, bias, 1)

    def weight_loader(self, param, loaded_weight, *args, **kwargs):
        if param.data.dim() == 1:
            if self.tp_rank == 0:
                param.data.copy_(materialize(loaded_weight))
            return
        shard = param.data.size(1)
        param.data.copy_(materialize(loaded_weight.narrow(1, self.tp_rank * shard, shard)))

    def forward(self, x, reduce: bool = True, partial_ok: bool = False):
        """``reduce=False`` returns this rank's partial product: the caller sums it together with the
        residual add + RMSNorm that follows (parallel_state.tp_all_reduce_add_rmsnorm).
        ``partial_ok`` (single rank): the caller takes a ``PartialSum`` — K-split fp32 partial results that the
        residual add + RMSNorm behind this projection adds up (hip_ops.add_rmsnorm_partial)."""
        kind = "row" if partial_ok and self.tp_size == 1 else "col"
        y = linear(x, self.weight, self.bias if self.tp_rank == 0 else None, self.weight_packed, kind)
        return tp_all_reduce(y) if self.tp_size > 1 and reduce else y
