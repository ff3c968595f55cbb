"""ctypes binding of libmvhip.so (include/mvhip.h) for torch tensors.

This is the only place the Python host touches the C-ABI.  Every wrapper
launches on torch's *current* stream, so the calls are captured by
``torch.cuda.graph`` (= hipGraph) like any torch op.  There is no fallback:
if the shared library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvhip.so")

MV_BF16, MV_F16, MV_F32 = 0, 1, 2
_DTYPES = {torch.bfloat16: MV_BF16, torch.float16: MV_F16, torch.float32: MV_F32}

# name -> (restype, argtypes); mirrors include/mvhip.h one-to-one
_SIGNATURES = {
    "mv_version": (c_int, []),
    "mv_strerror": (c_char_p, [c_int]),
    "mv_store_kvcache": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "mv_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_float,
                           c_int, c_void_p]),
    "mv_add_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                               c_int, c_int, c_float, c_int, c_void_p]),
    "mv_add_rmsnorm_partial": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                       c_float, c_int, c_void_p]),
    "mv_rope_qk_store": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                 c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p]),
    "mv_paged_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mv_paged_decode": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_float,
                                c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mv_rope_paged_decode": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int64, c_float, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mv_prefill_varlen": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                  c_int, c_void_p]),
    "mv_sample_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mv_sample": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                          c_uint64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t,
                          c_int, c_void_p]),
    "mv_decode_advance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mv_silu_and_mul": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "mv_linear_small_m": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                  c_int, c_int, c_int, c_void_p]),
    "mv_gate_up_silu_small_m": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p]),
    "mv_linear_small_m_packed": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                         c_int, c_int, c_int, c_void_p]),
    "mv_gate_up_silu_small_m_packed": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                               c_int, c_int, c_void_p]),
    "mv_norm_linear_small_m_packed": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                              c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                              c_int, c_int, c_void_p]),
    "mv_norm_gate_up_silu_small_m_packed": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                                    c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int, c_int,
                                                    c_int, c_int, c_int, c_void_p]),
    "mv_packed_weight_elems": (c_size_t, [c_int, c_int]),
    "mv_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mv_linear_mid_m": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                c_int, c_int, c_int, c_void_p]),
    "mv_gate_up_silu_mid_m": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "mv_linear_mid_m_partial": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p]),
    "mv_ar_region_bytes": (c_size_t, [c_size_t]),
    "mv_ar_max_world": (c_int, []),
    "mv_ar_alloc": (c_int, [c_size_t, c_void_p]),
    "mv_ar_free": (c_int, [c_void_p]),
    "mv_ar_ipc_handle_bytes": (c_int, []),
    "mv_ar_ipc_export": (c_int, [c_void_p, c_void_p]),
    "mv_ar_ipc_open": (c_int, [c_void_p, c_void_p]),
    "mv_ar_ipc_close": (c_int, [c_void_p]),
    "mv_ar_status": (c_int, [c_void_p, c_size_t, c_void_p]),
    "mv_ar_status_async": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "mv_ar_set_wait_limit_ms": (c_int, [c_void_p, c_size_t, ctypes.c_uint32]),
    "mv_ar_all_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_size_t, c_int, c_void_p]),
    "mv_ar_all_reduce_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p,
                                             c_int, c_int, c_size_t, c_int, c_void_p]),
    "mv_greedy_pack": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int,
                               c_void_p]),
    "mv_greedy_merge": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mv_mass_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mv_mass_merge": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                              c_void_p]),
    "mv_topc_width": (c_int, []),
    "mv_topc_pack": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mv_topc_unpack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mv_shard_key_hist": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mv_nucleus_pick": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_void_p]),
    "mv_mask_below": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mv_topc_pick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mv_decode_layers_persistent_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mv_decode_layers_persistent": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p]),
    "mv_persist_struct_bytes": (c_size_t, [c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class MvHipError(RuntimeError):
    pass


# ---------------------------------------------------------------- tuning switches
# The C-ABI reads no environment: the overrides below are read HERE, once, and passed as explicit arguments.
MV_DECODE_VECTOR, MV_DECODE_MFMA, MV_DECODE_KLDS_OFF, MV_DECODE_KLDS_ON, MV_DECODE_PER_HEAD = 1, 2, 4, 8, 16
MV_PREFILL_SIMPLE = 1
MV_E_SHAPE = -2   # include/mvhip.h


def _decode_flags_from_env() -> int:
    """MV_DECODE_MFMA_MIN_G=1 / 99: matrix-core decode kernel always / never; MV_DECODE_MFMA_KLDS=0 / 1: K straight
    to registers / through LDS; MV_DECODE_PER_HEAD=1: vector kernel with one kv head per workgroup;
    MV_DECODE_TARGET_WGS=n: workgroups the context split aims for (multiple of 16).
    (Bench and debugging aids; unset = the kernels choose from the shape.)"""
    f = 0
    g = os.environ.get("MV_DECODE_MFMA_MIN_G")
    if g is not None:
        f |= MV_DECODE_MFMA if int(g) <= 1 else (MV_DECODE_VECTOR if int(g) >= 17 else 0)
    k = os.environ.get("MV_DECODE_MFMA_KLDS")
    if k is not None:
        f |= MV_DECODE_KLDS_ON if int(k) else MV_DECODE_KLDS_OFF
    if os.environ.get("MV_DECODE_PER_HEAD") == "1":
        f |= MV_DECODE_PER_HEAD
    t = os.environ.get("MV_DECODE_TARGET_WGS")
    if t is not None:
        f |= (max(1, min(255, int(t) // 16))) << 8
    return f


DECODE_FLAGS = _decode_flags_from_env()
PREFILL_FLAGS = MV_PREFILL_SIMPLE if os.environ.get("MV_PREFILL_FORCE_SIMPLE") == "1" else 0


def _load():
    if not os.path.exists(LIB_PATH):
        raise MvHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C mini-vllm_amd/csrc`). "
            "There is no CPU fallback for the GPU hot path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()


def lib():
    return _lib


def version() -> int:
    return int(_lib.mv_version())


def _check(code: int, what: str) -> None:
    if code != 0:
        msg = _lib.mv_strerror(code)
        raise MvHipError(f"{what} failed: [{code}] {msg.decode() if msg else '?'}")


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise MvHipError(f"unsupported dtype {t.dtype}") from None


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MvHipError("libmvhip kernels need device tensors (got a CPU tensor); "
                             "the product path has no CPU fallback")


def _i32(t):
    if t is None:
        return None
    return t if t.dtype == torch.int32 else t.to(torch.int32)


# ---------------------------------------------------------------- workspace
_workspace: dict[int, torch.Tensor] = {}
_workspace_retired: list[torch.Tensor] = []   # outgrown buffers stay alive: captured graphs hold their pointers


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-device scratch buffer (allocate before graph capture).  A buffer that is outgrown is never
    freed: a hipGraph captured earlier still reads and writes it (split-KV partials), and handing its memory
    back to the caching allocator would let another tensor land under a replaying graph."""
    idx = torch.device(device).index or 0
    buf = _workspace.get(idx)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _workspace_retired.append(buf)
        # generous first allocation: captured graphs hold this pointer, so it should not have to move later
        buf = torch.empty(max(int(nbytes), 32 << 20), dtype=torch.uint8, device=device)
        _workspace[idx] = buf
    return buf


# ---------------------------------------------------------------- K1
def store_kvcache(key, value, k_cache, v_cache, slot_mapping) -> None:
    """key/value [N,Hkv,D] (last two dims dense), caches [nb,bs,Hkv,D] contiguous."""
    _need_cuda(key, value, k_cache, v_cache, slot_mapping)
    n, hkv, d = key.shape
    if slot_mapping.numel() != n:
        raise ValueError(f"slot_mapping size {slot_mapping.numel()} != batch_size {n}")
    if key.stride(-1) != 1 or key.stride(1) != d or value.stride(-1) != 1 or value.stride(1) != d:
        raise MvHipError("key/value must be dense in their last two dims")
    if key.dtype != k_cache.dtype:
        key, value = key.to(k_cache.dtype), value.to(v_cache.dtype)
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise MvHipError("KV cache must be contiguous")
    num_slots = k_cache.numel() // (hkv * d)
    _check(_lib.mv_store_kvcache(key.data_ptr(), value.data_ptr(), key.stride(0), value.stride(0),
                                 k_cache.data_ptr(), v_cache.data_ptr(), _i32(slot_mapping).data_ptr(),
                                 n, hkv, d, num_slots, _dt(k_cache), _stream()), "mv_store_kvcache")


# ---------------------------------------------------------------- K6
def _rows2d(x):
    cols = x.shape[-1]
    if x.stride(-1) != 1:
        x = x.contiguous()
    x2 = x.reshape(-1, cols)  # view when possible
    return x2, cols


def rmsnorm(x, weight, eps: float):
    _need_cuda(x, weight)
    x2, cols = _rows2d(x)
    out = torch.empty((x2.shape[0], cols), dtype=x.dtype, device=x.device)
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(_lib.mv_rmsnorm(x2.data_ptr(), x2.stride(0), w.data_ptr(), out.data_ptr(), out.stride(0),
                           x2.shape[0], cols, float(eps), _dt(x), _stream()), "mv_rmsnorm")
    return out.view(x.shape)


def add_rmsnorm(x, residual, weight, eps: float, inplace: bool = True):
    """Returns (normed, new_residual).  With inplace=True ``residual`` is overwritten
    by the rounded sum (the reference allocates a new tensor, layernorm.py:95-105)."""
    _need_cuda(x, residual, weight)
    x2, cols = _rows2d(x)
    res = residual if inplace else residual.clone()
    if not res.is_contiguous():
        res = res.contiguous()
    r2 = res.view(-1, cols)
    out = torch.empty((x2.shape[0], cols), dtype=x.dtype, device=x.device)
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(_lib.mv_add_rmsnorm(x2.data_ptr(), x2.stride(0), r2.data_ptr(), r2.stride(0), w.data_ptr(),
                               out.data_ptr(), out.stride(0), x2.shape[0], cols, float(eps), _dt(x),
                               _stream()), "mv_add_rmsnorm")
    return out.view(x.shape), res.view(residual.shape)


def add_rmsnorm_partial(partial, residual, weight, eps: float):
    """``add_rmsnorm`` whose x is the sum of fp32 K-split partials [S, rows, cols] (linear_mid_m_partial);
    residual is updated in place.  Returns (normed, residual)."""
    _need_cuda(partial, residual, weight)
    s, rows, cols = partial.shape
    if partial.dtype != torch.float32 or not partial.is_contiguous() or not residual.is_contiguous():
        raise MvHipError("partial sums must be contiguous fp32, the residual contiguous")
    r2 = residual.view(-1, cols)
    out = torch.empty((rows, cols), dtype=residual.dtype, device=residual.device)
    w = weight if weight.dtype == residual.dtype else weight.to(residual.dtype)
    _check(_lib.mv_add_rmsnorm_partial(partial.data_ptr(), s, r2.data_ptr(), r2.stride(0), w.data_ptr(),
                                       out.data_ptr(), out.stride(0), rows, cols, float(eps), _dt(residual),
                                       _stream()), "mv_add_rmsnorm_partial")
    return out.view(residual.shape), residual


# ---------------------------------------------------------------- K7
def rope_qk_store(q, k, v, positions, cos, sin, q_norm_w=None, k_norm_w=None, eps: float = 1e-6,
                  k_cache=None, v_cache=None, slot_mapping=None, want_k_out: bool = True):
    """q [T,H,D], k/v [T,Hkv,D] (views of the qkv GEMM output are fine).
    Returns (q_rot [T,H,D], k_rot [T,Hkv,D] or None)."""
    _need_cuda(q, k, positions, cos, sin)
    t, h, d = q.shape
    hkv = k.shape[1]
    for name, x in (("q", q), ("k", k), ("v", v)):
        if x is not None and (x.stride(-1) != 1 or x.stride(1) != d):
            raise MvHipError(f"{name} must be dense in its last two dims")
    q_out = torch.empty((t, h, d), dtype=q.dtype, device=q.device)
    k_out = torch.empty((t, hkv, d), dtype=q.dtype, device=q.device) if want_k_out else None
    has_cache = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None \
        and slot_mapping.numel() > 0
    num_slots = 0
    if has_cache:
        if k_cache.dtype != q.dtype:
            raise MvHipError("q/k/v dtype must equal the KV-cache dtype")
        num_slots = k_cache.numel() // (hkv * d)
        if slot_mapping.numel() != t:
            raise ValueError(f"slot_mapping size {slot_mapping.numel()} != batch_size {t}")
    pos = positions if positions.dtype == torch.int64 else positions.to(torch.int64)
    qn = q_norm_w if q_norm_w is None or q_norm_w.dtype == q.dtype else q_norm_w.to(q.dtype)
    kn = k_norm_w if k_norm_w is None or k_norm_w.dtype == q.dtype else k_norm_w.to(q.dtype)
    _check(_lib.mv_rope_qk_store(
        q.data_ptr(), k.data_ptr(), _ptr(v), q.stride(0), k.stride(0), v.stride(0) if v is not None else 0,
        pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), cos.shape[0], _ptr(qn), _ptr(kn), float(eps),
        q_out.data_ptr(), _ptr(k_out),
        k_cache.data_ptr() if has_cache else None, v_cache.data_ptr() if has_cache else None,
        _i32(slot_mapping).data_ptr() if has_cache else None, num_slots, t, h, hkv, d, _dt(q),
        _stream()), "mv_rope_qk_store")
    return q_out, k_out


# ---------------------------------------------------------------- K3
def paged_decode_workspace_bytes(batch, n_heads, n_kv_heads, head_dim, block_size, max_blocks,
                                 flags: int | None = None) -> int:
    return int(_lib.mv_paged_decode_workspace_bytes(batch, n_heads, n_kv_heads, head_dim, block_size,
                                                    max_blocks, DECODE_FLAGS if flags is None else flags))


def paged_decode(q, k_cache, v_cache, block_tables, context_lens, scale: float, out=None, flags: int | None = None):
    """q [B,H,D]; caches [nb,bs,Hkv,D]; block_tables int [B,nblk] (row-strided view ok);
    context_lens int [B].  Returns [B,H,D] in q's dtype."""
    _need_cuda(q, k_cache, v_cache, block_tables, context_lens)
    b, h, d = q.shape
    if q.stride(-1) != 1 or q.stride(1) != d:
        q = q.contiguous()
    if out is None:
        out = torch.empty((b, h, d), dtype=q.dtype, device=q.device)
    nb, bs, hkv = k_cache.shape[0], k_cache.shape[1], k_cache.shape[2]
    bt = _i32(block_tables)
    if bt.dim() != 2 or (bt.shape[1] > 0 and bt.stride(1) != 1):
        bt = bt.contiguous()
    cl = _i32(context_lens)
    max_blocks = bt.shape[1]
    flags = DECODE_FLAGS if flags is None else flags
    ws_bytes = paged_decode_workspace_bytes(b, h, hkv, d, bs, max_blocks, flags)
    ws = workspace(ws_bytes, q.device)
    _check(_lib.mv_paged_decode(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
                                bt.data_ptr(), bt.stride(0) if max_blocks else 0, cl.data_ptr(),
                                out.data_ptr(), b, h, hkv, d, bs, max_blocks, nb, float(scale),
                                ws.data_ptr(), ws.numel(), _dt(q), flags, _stream()), "mv_paged_decode")
    return out


# ---------------------------------------------------------------- K2
def rope_paged_decode(q, k, v, positions, cos, sin, q_norm_w, k_norm_w, eps: float, slot_mapping,
                      k_cache, v_cache, block_tables, context_lens, scale: float, flags: int | None = None):
    """Decode step of one layer in ONE launch: q [B,H,D], k/v [B,Hkv,D] RAW rows of the qkv projection (views are
    fine) -> per-head RMSNorm (weights or None) + RoPE on q and k, the new K/V rows stored at ``slot_mapping``, paged
    attention over ``block_tables`` / ``context_lens`` (which count the new token).  Returns [B,H,D], or None when
    the fused kernel does not take the layout (the caller then runs rope_qk_store + paged_decode)."""
    _need_cuda(q, k, v, k_cache, v_cache, block_tables, context_lens, slot_mapping)
    b, h, d = q.shape
    hkv = k.shape[1]
    nb, bs = k_cache.shape[0], k_cache.shape[1]
    if q.dtype not in (torch.bfloat16, torch.float16) or k_cache.dtype != q.dtype or slot_mapping.numel() != b:
        return None
    for x in (q, k, v):
        if x.stride(-1) != 1 or x.stride(1) != d:
            return None
    bt, cl = _i32(block_tables), _i32(context_lens)
    if bt.dim() != 2 or bt.stride(1) != 1 or bt.shape[1] == 0:
        return None
    flags = DECODE_FLAGS if flags is None else flags
    ws_bytes = paged_decode_workspace_bytes(b, h, hkv, d, bs, bt.shape[1], flags)
    ws = workspace(ws_bytes, q.device)
    out = torch.empty((b, h, d), dtype=q.dtype, device=q.device)
    pos = positions if positions.dtype == torch.int64 else positions.to(torch.int64)
    qn = q_norm_w if q_norm_w is None or q_norm_w.dtype == q.dtype else q_norm_w.to(q.dtype)
    kn = k_norm_w if k_norm_w is None or k_norm_w.dtype == q.dtype else k_norm_w.to(q.dtype)
    rc = _lib.mv_rope_paged_decode(
        q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), pos.data_ptr(),
        cos.data_ptr(), sin.data_ptr(), cos.shape[0], _ptr(qn), _ptr(kn), float(eps), _i32(slot_mapping).data_ptr(),
        k_cache.data_ptr(), v_cache.data_ptr(), bt.data_ptr(), bt.stride(0), cl.data_ptr(), out.data_ptr(), b, h, hkv,
        d, bs, bt.shape[1], nb, float(scale), ws.data_ptr(), ws.numel(), _dt(q), flags, _stream())
    if rc == MV_E_SHAPE:
        return None
    _check(rc, "mv_rope_paged_decode")
    return out


def prefill_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, scale: float,
                   k_cache=None, v_cache=None, block_tables=None, out=None, flags: int | None = None):
    """Packed causal prefill.  Contiguous K/V (k,v [Tk,Hkv,D]) unless ``block_tables`` is
    given, in which case keys come from the paged caches (prefix-cache hit)."""
    _need_cuda(q, cu_seqlens_q, cu_seqlens_k)
    t, h, d = q.shape
    if q.stride(-1) != 1 or q.stride(1) != d:
        q = q.contiguous()
    if out is None:
        out = torch.empty((t, h, d), dtype=q.dtype, device=q.device)
    n_seqs = cu_seqlens_q.numel() - 1
    paged = block_tables is not None
    if paged:
        nb, bs, hkv = k_cache.shape[0], k_cache.shape[1], k_cache.shape[2]
        bt = _i32(block_tables)
        if bt.stride(1) != 1:
            bt = bt.contiguous()
        args_kv = (None, None, 0, 0, k_cache.data_ptr(), v_cache.data_ptr(), bt.data_ptr(), bt.stride(0), nb)
    else:
        hkv = k.shape[1]
        bs = 0
        for name, x in (("k", k), ("v", v)):
            if x.stride(-1) != 1 or x.stride(1) != d:
                raise MvHipError(f"{name} must be dense in its last two dims")
        args_kv = (k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0), None, None, None, 0, 0)
    _check(_lib.mv_prefill_varlen(q.data_ptr(), q.stride(0), *args_kv, out.data_ptr(),
                                  _i32(cu_seqlens_q).data_ptr(), _i32(cu_seqlens_k).data_ptr(), n_seqs,
                                  int(max_seqlen_q), int(max_seqlen_k), h, hkv, d, bs, float(scale),
                                  _dt(q), PREFILL_FLAGS if flags is None else flags, _stream()), "mv_prefill_varlen")
    return out


# ---------------------------------------------------------------- K8
def sample(logits, temperatures, top_ks, top_ps, min_ps, seed: int = 0, offset: int = 0,
           offset_dev=None, return_mask: bool = False, return_stats: bool = False):
    """logits [B,V]; per-row params [B].  Returns int64 tokens [B] (and the uint8 keep mask; and, with
    ``return_stats``, fp32 [B, 2] = {row maximum, kept mass}: the input of ``mass_pack``)."""
    _need_cuda(logits, temperatures, top_ks, top_ps, min_ps)
    b, v = logits.shape
    if logits.stride(-1) != 1:
        logits = logits.contiguous()
    out = torch.empty((b,), dtype=torch.int64, device=logits.device)
    mask = torch.empty((b, v), dtype=torch.uint8, device=logits.device) if return_mask else None
    stats = torch.empty((b, 2), dtype=torch.float32, device=logits.device) if return_stats else None

    def f32(x):
        return None if x is None else (x if x.dtype == torch.float32 else x.float())
    tk = None if top_ks is None else (top_ks if top_ks.dtype == torch.int64 else top_ks.to(torch.int64))
    temperatures, top_ps, min_ps = f32(temperatures), f32(top_ps), f32(min_ps)
    _check(_lib.mv_sample(logits.data_ptr(), logits.stride(0), _ptr(temperatures), _ptr(tk), _ptr(top_ps),
                          _ptr(min_ps), int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                          _ptr(offset_dev), out.data_ptr(), _ptr(mask), b, v, _ptr(stats), 8 * b if return_stats else 0,
                          _dt(logits), _stream()), "mv_sample")
    if return_stats:
        return (out, mask, stats) if return_mask else (out, stats)
    return (out, mask) if return_mask else out


# ---------------------------------------------------------------- K9
def silu_and_mul(x):
    _need_cuda(x)
    if x.size(-1) % 2 != 0:
        raise ValueError(f"Input last dimension must be even for SiluAndMul, got {x.size(-1)}")
    x2, cols2 = _rows2d(x)
    cols = cols2 // 2
    out = torch.empty((x2.shape[0], cols), dtype=x.dtype, device=x.device)
    _check(_lib.mv_silu_and_mul(x2.data_ptr(), x2.stride(0), out.data_ptr(), out.stride(0), x2.shape[0],
                                cols, _dt(x), _stream()), "mv_silu_and_mul")
    return out.view(*x.shape[:-1], cols)


# ---------------------------------------------------------------- K10
LINEAR_SMALL_M_MAX = 64


def linear_small_m_ok(x, weight) -> bool:
    """True when mv_linear_small_m takes this shape (else the caller uses the library GEMM)."""
    k = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x.dtype
            and weight.dim() == 2 and weight.shape[1] == k and k >= 256 and k % 32 == 0
            and 0 < x.numel() // k <= LINEAR_SMALL_M_MAX and weight.shape[0] % 4 == 0
            and weight.is_contiguous() and x.stride(-1) == 1 and x.data_ptr() % 16 == 0
            and (x.dim() != 2 or x.stride(0) % 8 == 0))


def small_m_shape(rpw: int = 0, waves: int = 0, xlds: int = 0, wlds: int = 0) -> int:
    """Launch-shape override of the small-M kernels (tools/bench_linear.py --sweep): xlds / wlds 1 = on, -1 = off."""
    return (rpw & 15) | ((waves & 31) << 4) | (({0: 0, 1: 1, -1: 2}[xlds]) << 9) | (({0: 0, 1: 1, -1: 2}[wlds]) << 11)


def _packed_ok(packed, n_rows: int, k: int, dtype) -> bool:
    return (packed is not None and packed.dtype == dtype and packed.is_contiguous()
            and packed.numel() == -(-n_rows // 16) * 16 * k)


def linear_small_m(x, weight, bias=None, shape: int = 0, packed=None):
    """x [..., K] (<= 64 rows), weight [N, K], bias [N] | None -> [..., N]; fp32 accumulation.  ``packed``: the
    fragment-ordered copy of ``weight`` (pack_weight) — same result, streamed in whole lines."""
    _need_cuda(x, weight)
    x2, k = _rows2d(x)
    n = weight.shape[0]
    if weight.shape[1] != k or not weight.is_contiguous():
        raise ValueError(f"weight {tuple(weight.shape)} does not match x [..., {k}]")
    if bias is not None and (bias.dtype != x.dtype or bias.shape[0] != n):
        raise ValueError("bias dtype/shape mismatch")
    out = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    if _packed_ok(packed, n, k, x.dtype):
        _check(_lib.mv_linear_small_m_packed(x2.data_ptr(), x2.stride(0), packed.data_ptr(), _ptr(bias),
                                             out.data_ptr(), out.stride(0), x2.shape[0], n, k, _dt(x), int(shape),
                                             _stream()), "mv_linear_small_m_packed")
    else:
        _check(_lib.mv_linear_small_m(x2.data_ptr(), x2.stride(0), weight.data_ptr(), _ptr(bias), out.data_ptr(),
                                      out.stride(0), x2.shape[0], n, k, _dt(x), int(shape), _stream()),
               "mv_linear_small_m")
    return out.view(*x.shape[:-1], n)


def gate_up_silu_small_m(x, weight, shape: int = 0, packed=None):
    """x [..., K] (<= 64 rows), weight [2I, K] = [gate rows ; up rows] -> silu(x Wg^T) * (x Wu^T), [..., I]."""
    _need_cuda(x, weight)
    x2, k = _rows2d(x)
    if weight.shape[0] % 2 or weight.shape[1] != k or not weight.is_contiguous():
        raise ValueError(f"weight {tuple(weight.shape)} is not a merged [2I, {k}] gate_up matrix")
    n = weight.shape[0] // 2
    out = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    if n % 16 == 0 and _packed_ok(packed, 2 * n, k, x.dtype):
        _check(_lib.mv_gate_up_silu_small_m_packed(x2.data_ptr(), x2.stride(0), packed.data_ptr(), out.data_ptr(),
                                                   out.stride(0), x2.shape[0], n, k, _dt(x), int(shape), _stream()),
               "mv_gate_up_silu_small_m_packed")
    else:
        _check(_lib.mv_gate_up_silu_small_m(x2.data_ptr(), x2.stride(0), weight.data_ptr(), out.data_ptr(),
                                            out.stride(0), x2.shape[0], n, k, _dt(x), int(shape), _stream()),
               "mv_gate_up_silu_small_m")
    return out.view(*x.shape[:-1], n)


# rows x vectors-per-norm-thread up to which the layer norm rides in the prologue of the small-M projection behind it.  The
# kernel takes up to 8 (csrc/linear_small_m.hip: kNormRows, in passes of 4 / PT rows, round 4), but beyond one pass the
# prologue costs more than the norm launch it saves — Qwen3-1.7B, ctx 1024: B = 5 1.52 -> 1.61 ms per step, B = 8 1.66 ->
# 1.81 (gpurun_out/r4e_*) — so the default stays at one pass (4); MV_NORM_FOLD_MAX_ROWS=8 selects the longer fold
NORM_FUSED_MAX_ROWS = min(8, int(os.environ.get("MV_NORM_FOLD_MAX_ROWS", "4")))


def norm_linear_small_m_ok(x, n_rows_w: int, packed) -> bool:
    """True when the norm-in-the-prologue form of the small-M projection takes ``x`` [rows, K] (rows x vectors per
    norm thread <= 4, the packed weight present)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.stride(1) == 1):
        return False
    rows, k = x.shape
    if not 0 < rows <= NORM_FUSED_MAX_ROWS or k % 64 or k > 8192 or x.data_ptr() % 16 or x.stride(0) % 8:
        return False
    nvec = k // 8
    threads = min(256, -(-nvec // 64) * 64)
    per_thread = -(-nvec // threads)
    pt = 1 if per_thread <= 1 else (2 if per_thread == 2 else 4)
    return rows * pt <= NORM_FUSED_MAX_ROWS and _packed_ok(packed, n_rows_w, k, x.dtype)


def norm_linear_small_m(x, residual, norm_weight, eps: float, weight_packed, n: int, bias=None, gated: bool = False):
    """(RMSNorm -> projection) of a <= 4-row decode step in ONE launch.  x [rows, K]; residual [rows, K] | None.
    Returns (out [rows, n], new_residual) — new_residual is ``x`` itself when residual is None (reference:
    layernorm.py:81-105, the caller keeps x as the residual), else a fresh tensor holding x + residual.
    gated: weight_packed is the packed merged [2n, K] gate_up matrix, out = silu(gate) * up.
    Returns None when the kernel declines the shape (the caller then runs norm and projection separately)."""
    _need_cuda(x, norm_weight, weight_packed)
    rows, k = x.shape
    out = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    nw = norm_weight if norm_weight.dtype == x.dtype else norm_weight.to(x.dtype)
    if residual is not None:
        r2 = residual if residual.is_contiguous() else residual.contiguous()
        r2 = r2.view(rows, k)
        res_out = torch.empty_like(r2)
        rp, rs, rop, ros = r2.data_ptr(), r2.stride(0), res_out.data_ptr(), res_out.stride(0)
    else:
        res_out, rp, rs, rop, ros = x, None, 0, None, 0
    if gated:
        rc = _lib.mv_norm_gate_up_silu_small_m_packed(x.data_ptr(), x.stride(0), rp, rs, rop, ros, nw.data_ptr(),
                                                      float(eps), weight_packed.data_ptr(), out.data_ptr(),
                                                      out.stride(0), rows, n, k, _dt(x), 0, _stream())
    else:
        rc = _lib.mv_norm_linear_small_m_packed(x.data_ptr(), x.stride(0), rp, rs, rop, ros, nw.data_ptr(),
                                                float(eps), weight_packed.data_ptr(), _ptr(bias), out.data_ptr(),
                                                out.stride(0), rows, n, k, _dt(x), 0, _stream())
    if rc == MV_E_SHAPE:
        return None
    _check(rc, "mv_norm_linear_small_m_packed")
    return out, res_out


# ---------------------------------------------------------------- K11
def pack_weight(weight):
    """[N, K] row-major -> the fragment-ordered copy mv_linear_mid_m streams ([ceil(N/16), K/32, 64, 8])."""
    _need_cuda(weight)
    n, k = weight.shape
    if not weight.is_contiguous():
        weight = weight.contiguous()
    out = torch.empty((int(_lib.mv_packed_weight_elems(n, k)),), dtype=weight.dtype, device=weight.device)
    _check(_lib.mv_pack_weight(weight.data_ptr(), out.data_ptr(), n, k, _dt(weight), _stream()), "mv_pack_weight")
    return out.view((n + 15) // 16, k // 32, 64, 8)


def linear_mid_m_ok(x, n: int) -> bool:
    """True when mv_linear_mid_m takes this shape (else the caller uses the library GEMM)."""
    k = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and k % 64 == 0 and k >= 64
            and x.numel() // k > 0 and n % 4 == 0 and x.stride(-1) == 1 and x.data_ptr() % 16 == 0
            and (x.dim() != 2 or x.stride(0) % 8 == 0))


def linear_mid_m(x, w_packed, n: int, bias=None, cfg: int = 0, out=None):
    """x [..., K], w_packed = pack_weight(W [n, K]), bias [n] | None -> [..., n]; fp32 accumulation."""
    _need_cuda(x, w_packed)
    x2, k = _rows2d(x)
    if w_packed.numel() != ((n + 15) // 16) * 16 * k or w_packed.dtype != x.dtype:
        raise ValueError(f"packed weight of {w_packed.numel()} elements does not match [{n}, {k}]")
    if bias is not None and (bias.dtype != x.dtype or bias.shape[0] != n):
        raise ValueError("bias dtype/shape mismatch")
    if out is None:
        out = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    _check(_lib.mv_linear_mid_m(x2.data_ptr(), x2.stride(0), w_packed.data_ptr(), _ptr(bias), out.data_ptr(),
                                out.stride(0), x2.shape[0], n, k, _dt(x), int(cfg), _stream()), "mv_linear_mid_m")
    return out.view(*x.shape[:-1], n)


def gate_up_silu_mid_m(x, w_packed, n_inter: int, cfg: int = 0):
    """x [..., K], w_packed = pack_weight([gate rows ; up rows] of shape [2I, K]) -> silu(x Wg^T) * (x Wu^T)."""
    _need_cuda(x, w_packed)
    x2, k = _rows2d(x)
    if w_packed.numel() != 2 * n_inter * k or n_inter % 16:
        raise ValueError(f"packed weight of {w_packed.numel()} elements is not a merged [2*{n_inter}, {k}] matrix")
    out = torch.empty((x2.shape[0], n_inter), dtype=x.dtype, device=x.device)
    _check(_lib.mv_gate_up_silu_mid_m(x2.data_ptr(), x2.stride(0), w_packed.data_ptr(), out.data_ptr(),
                                      out.stride(0), x2.shape[0], n_inter, k, _dt(x), int(cfg), _stream()),
           "mv_gate_up_silu_mid_m")
    return out.view(*x.shape[:-1], n_inter)


def linear_mid_m_partial(x, w_packed, n: int, splits: int, cfg: int = 0):
    """fp32 partial sums [splits, M, n] of x @ W^T over `splits` K slices."""
    _need_cuda(x, w_packed)
    x2, k = _rows2d(x)
    out = torch.empty((splits, x2.shape[0], n), dtype=torch.float32, device=x.device)
    _check(_lib.mv_linear_mid_m_partial(x2.data_ptr(), x2.stride(0), w_packed.data_ptr(), out.data_ptr(),
                                        x2.shape[0], n, k, int(splits), _dt(x), int(cfg), _stream()),
           "mv_linear_mid_m_partial")
    return out


# ---------------------------------------------------------------- K12: persistent small-batch decode
class MvPersistArgs(ctypes.Structure):
    """include/mvhip.h: MvPersistArgs (host struct, read at call time)."""
    _fields_ = [("m", c_int), ("hidden", c_int), ("n_heads", c_int), ("n_kv_heads", c_int), ("head_dim", c_int),
                ("inter", c_int), ("eps", c_float), ("scale", c_float), ("x_in", c_void_p), ("res_in", c_void_p),
                ("x_in_stride", c_int64), ("res_in_stride", c_int64), ("positions", c_void_p),
                ("slot_mapping", c_void_p), ("block_tables", c_void_p), ("context_lens", c_void_p),
                ("table_stride", c_int64), ("num_blocks", c_int64), ("max_pos", c_int64), ("block_size", c_int),
                ("max_blocks", c_int), ("cos", c_void_p), ("sin", c_void_p), ("out_h_offset", c_int64),
                ("out_res_offset", c_int64)]


PERSIST_LAYER_FIELDS = ("ln1_w", "wqkv", "bqkv", "q_norm_w", "k_norm_w", "wo", "ln2_w", "wgu", "wdown", "k_cache", "v_cache")
PERSIST_MAX_ROWS = 16


def persist_layer_table(rows, device) -> torch.Tensor:
    """Device array of MvPersistLayer: ``rows`` = per layer a dict field -> tensor | None (PERSIST_LAYER_FIELDS)."""
    flat = [[0 if r[f] is None else r[f].data_ptr() for f in PERSIST_LAYER_FIELDS] for r in rows]
    return torch.tensor(flat, dtype=torch.int64).to(device)


def decode_layers_persistent_workspace(m, hidden, n_heads, n_kv_heads, head_dim, inter, block_size, max_blocks, device):
    n = int(_lib.mv_decode_layers_persistent_workspace_bytes(m, hidden, n_heads, n_kv_heads, head_dim, inter, block_size,
                                                             max_blocks))
    return torch.zeros(n, dtype=torch.uint8, device=device)


def decode_layers_persistent(table, n_layers, x, positions, slot_mapping, block_tables, context_lens, cos, sin, k_cache0,
                             n_heads, n_kv_heads, head_dim, inter, eps, scale, workspace, status, flags: int = 0):
    """Every decoder layer of a decode step of <= 16 rows in ONE launch (csrc/decode_persistent.hip).  ``x`` [M, hidden] =
    the embedding output.  -> (h, residual): views of ``workspace`` holding the last layer's down_proj output and the
    residual (the caller applies the final norm), or None when the kernel does not take the shape (MV_E_SHAPE)."""
    _need_cuda(x, positions, slot_mapping, block_tables, context_lens, cos, sin, workspace, status)
    m, hidden = x.shape
    bt = _i32(block_tables)
    if bt.dim() != 2 or bt.stride(1) != 1 or x.stride(1) != 1 or positions.dtype != torch.int64:
        return None
    a = MvPersistArgs()
    a.m, a.hidden, a.n_heads, a.n_kv_heads, a.head_dim, a.inter = m, hidden, n_heads, n_kv_heads, head_dim, inter
    a.eps, a.scale = float(eps), float(scale)
    a.x_in, a.res_in, a.x_in_stride, a.res_in_stride = x.data_ptr(), None, x.stride(0), 0
    a.positions, a.slot_mapping = positions.data_ptr(), _i32(slot_mapping).data_ptr()
    a.block_tables, a.context_lens = bt.data_ptr(), _i32(context_lens).data_ptr()
    a.table_stride, a.num_blocks, a.max_pos = bt.stride(0), k_cache0.shape[0], cos.shape[0]
    a.block_size, a.max_blocks = k_cache0.shape[1], bt.shape[1]
    a.cos, a.sin = cos.data_ptr(), sin.data_ptr()
    rc = _lib.mv_decode_layers_persistent(table.data_ptr(), n_layers, ctypes.byref(a), workspace.data_ptr(),
                                          workspace.numel(), status.data_ptr(), _dt(x), int(flags), _stream())
    if rc == MV_E_SHAPE:
        return None
    _check(rc, "mv_decode_layers_persistent")
    esz = x.element_size()
    h = workspace[a.out_h_offset:a.out_h_offset + m * hidden * esz].view(x.dtype).view(m, hidden)
    r = workspace[a.out_res_offset:a.out_res_offset + m * hidden * esz].view(x.dtype).view(m, hidden)
    return h, r


# ---------------------------------------------------------------- f2
def greedy_pack(logits_local, local_tokens, rank: int, world: int, rows_padded: int):
    """-> buf [world, rows_padded, 4] in the logits' dtype (see mv_greedy_pack)."""
    _need_cuda(logits_local, local_tokens)
    rows, vpr = logits_local.shape
    buf = torch.empty((world, rows_padded, 4), dtype=logits_local.dtype, device=logits_local.device)
    _check(_lib.mv_greedy_pack(logits_local.data_ptr(), logits_local.stride(0), local_tokens.data_ptr(), rank, world,
                               vpr, rows, rows_padded, buf.data_ptr(), _dt(logits_local), _stream()),
           "mv_greedy_pack")
    return buf


def greedy_merge(buf, rows: int):
    world, rows_padded, _ = buf.shape
    out = torch.empty((rows,), dtype=torch.int64, device=buf.device)
    _check(_lib.mv_greedy_merge(buf.data_ptr(), world, rows, rows_padded, out.data_ptr(), _dt(buf), _stream()),
           "mv_greedy_merge")
    return out


TOPC = 64   # == mv_topc_width(): candidates a rank contributes per row


def topc_pack(logits_local, rank: int, world: int, rows_padded: int):
    """-> buf [world, rows_padded, 64, 4] in the logits' dtype (see mv_topc_pack)."""
    _need_cuda(logits_local)
    rows, vpr = logits_local.shape
    buf = torch.empty((world, rows_padded, TOPC, 4), dtype=logits_local.dtype, device=logits_local.device)
    _check(_lib.mv_topc_pack(logits_local.data_ptr(), logits_local.stride(0), vpr, rank, world, vpr, rows, rows_padded,
                             buf.data_ptr(), _dt(logits_local), _stream()), "mv_topc_pack")
    return buf


def topc_unpack(buf, rows: int):
    """-> (cand_logits [rows, world*64] in buf's dtype, cand_idx int32 [rows, world*64] global token ids)."""
    world, rows_padded = buf.shape[0], buf.shape[1]
    cl = torch.empty((rows, world * TOPC), dtype=buf.dtype, device=buf.device)
    ci = torch.empty((rows, world * TOPC), dtype=torch.int32, device=buf.device)
    _check(_lib.mv_topc_unpack(buf.data_ptr(), world, rows, rows_padded, cl.data_ptr(), ci.data_ptr(), _dt(buf), _stream()),
           "mv_topc_unpack")
    return cl, ci


def mass_pack(stats, local_tokens, rank: int, world: int, vocab_per_rank: int, rows_padded: int, dtype):
    """-> buf [world, rows_padded, 8] in the logits' 16-bit dtype (see mv_mass_pack)."""
    _need_cuda(stats, local_tokens)
    rows = local_tokens.shape[0]
    assert stats.dtype == torch.float32 and stats.shape == (rows, 2) and stats.is_contiguous()
    assert local_tokens.dtype == torch.int64 and dtype in (torch.bfloat16, torch.float16)
    buf = torch.empty((world, rows_padded, 8), dtype=dtype, device=stats.device)
    _check(_lib.mv_mass_pack(stats.data_ptr(), local_tokens.data_ptr(), rank, world, vocab_per_rank, rows, rows_padded,
                             buf.data_ptr(), _DTYPES[dtype], _stream()), "mv_mass_pack")
    return buf


def mass_merge(buf, rows: int, temperatures, seed: int = 0, offset: int = 0, offset_dev=None):
    """Summed ``mass_pack`` buffers -> int64 tokens [rows] (see mv_mass_merge; seed / offset as given to ``sample``)."""
    _need_cuda(buf, temperatures)
    world, rows_padded, _ = buf.shape
    assert temperatures is None or (temperatures.dtype == torch.float32 and temperatures.is_contiguous())
    out = torch.empty((rows,), dtype=torch.int64, device=buf.device)
    _check(_lib.mv_mass_merge(buf.data_ptr(), _ptr(temperatures), int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                              _ptr(offset_dev), world, rows, rows_padded, out.data_ptr(), _dt(buf), _stream()),
           "mv_mass_merge")
    return out


def shard_key_hist(logits_local, row_max, temperatures, prefix=None):
    """-> int64 [rows, 256]: this shard's fixed-point mass per key bin below the row's global maximum (mv_shard_key_hist)."""
    _need_cuda(logits_local, row_max, temperatures, prefix)
    rows, vl = logits_local.shape
    assert logits_local.stride(1) == 1 and row_max.dtype == torch.float32 and row_max.is_contiguous()
    assert temperatures is None or (temperatures.dtype == torch.float32 and temperatures.is_contiguous())
    assert prefix is None or (prefix.dtype == torch.int32 and prefix.is_contiguous())
    hist = torch.empty((rows, 256), dtype=torch.int64, device=logits_local.device)
    _check(_lib.mv_shard_key_hist(logits_local.data_ptr(), logits_local.stride(0), vl, rows, row_max.data_ptr(),
                                  _ptr(temperatures), _ptr(prefix), hist.data_ptr(), _dt(logits_local), _stream()),
           "mv_shard_key_hist")
    return hist


def nucleus_pick(hist, level: int, top_ps, min_ps, temperatures, row_max, dtype, state=None, prefix=None):
    """Level 0 -> (state int64 [rows, 2], prefix int32 [rows]); level 1 (with those) -> floor fp32 [rows]
    (mv_nucleus_pick; ``hist`` = the all-reduced ``shard_key_hist`` of that level)."""
    _need_cuda(hist, top_ps, min_ps, temperatures, row_max)
    rows = hist.shape[0]
    assert hist.dtype == torch.int64 and hist.shape == (rows, 256) and hist.is_contiguous()
    for x in (top_ps, min_ps, temperatures, row_max):
        assert x is None or (x.dtype == torch.float32 and x.is_contiguous())
    dev = hist.device
    if level == 0:
        state = torch.empty((rows, 2), dtype=torch.int64, device=dev)
        prefix = torch.empty((rows,), dtype=torch.int32, device=dev)
    floor = torch.empty((rows,), dtype=torch.float32, device=dev) if level == 1 else None
    _check(_lib.mv_nucleus_pick(hist.data_ptr(), level, _ptr(top_ps), _ptr(min_ps), _ptr(temperatures), _ptr(row_max),
                                state.data_ptr(), prefix.data_ptr(), _ptr(floor), rows, _DTYPES[dtype], _stream()),
           "mv_nucleus_pick")
    return (state, prefix) if level == 0 else floor


def mask_below(logits, floor):
    """-> copy of ``logits`` [rows, V] with everything below the row's ``floor`` (fp32 [rows]) at -inf (mv_mask_below)."""
    _need_cuda(logits, floor)
    rows, vl = logits.shape
    assert logits.stride(1) == 1 and floor.dtype == torch.float32 and floor.is_contiguous()
    out = torch.empty((rows, vl), dtype=logits.dtype, device=logits.device)
    _check(_lib.mv_mask_below(logits.data_ptr(), logits.stride(0), vl, rows, floor.data_ptr(), out.data_ptr(),
                              out.stride(0), _dt(logits), _stream()), "mv_mask_below")
    return out


def topc_pick(cand_logits, cand_idx, pos, temperatures=None):
    """Candidate positions -> int64 token ids; temperature-0 rows take the lowest global index among the row maxima
    (see mv_topc_pick)."""
    _need_cuda(cand_logits, cand_idx, pos)
    rows, width = cand_logits.shape
    assert cand_idx.dtype == torch.int32 and pos.dtype == torch.int64 and cand_idx.shape == cand_logits.shape
    assert cand_logits.is_contiguous() and cand_idx.is_contiguous()
    if temperatures is not None:
        assert temperatures.dtype == torch.float32 and temperatures.shape[0] >= rows and temperatures.is_contiguous()
    out = torch.empty((rows,), dtype=torch.int64, device=cand_logits.device)
    _check(_lib.mv_topc_pick(cand_logits.data_ptr(), cand_idx.data_ptr(), pos.data_ptr(), _ptr(temperatures),
                             out.data_ptr(), rows, width, _dt(cand_logits), _stream()), "mv_topc_pick")
    return out


# ---------------------------------------------------------------- a2 (device side)
def decode_advance(input_ids, tokens, positions, slot_mapping, context_lens, rng_offset) -> None:
    """In place: ids <- tokens, positions/context/slots += 1 (padding rows untouched), rng += 1."""
    _need_cuda(input_ids, tokens, positions, slot_mapping, context_lens, rng_offset)
    assert input_ids.dtype == tokens.dtype == positions.dtype == torch.int64
    assert slot_mapping.dtype == context_lens.dtype == torch.int32
    _check(_lib.mv_decode_advance(input_ids.data_ptr(), tokens.data_ptr(), positions.data_ptr(),
                                  slot_mapping.data_ptr(), context_lens.data_ptr(), _ptr(rng_offset),
                                  input_ids.shape[0], _stream()), "mv_decode_advance")
