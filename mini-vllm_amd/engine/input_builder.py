"""Host-side batch preparation: ``list[Sequence]`` -> the integer tensors the kernels index
with.  This is the reference's ``InferenceExecutor._prepare_prefill_input`` /
``_prepare_decode_input`` / ``_prepare_block_tables[_decode]``
(minivllm/engine/inference_executor.py:468-688), split out of the executor so that it can be
checked bit for bit against reference traces without a GPU.

Everything is built with numpy in (pinned, when a GPU is present) staging buffers that are
allocated once; the executor ships each batch to the device with non-blocking copies.
Value and dtype contract (identical to the reference):
  input_ids, positions  int64 [T]          slot_mapping   int32 [T] or None
  cum_seqlens_q/k       int32 [n+1]        context_lens   int32 [B]
  block_tables          int32 [n, max blocks in batch], -1 padded, or None
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .sequence import Sequence

__all__ = ["BatchInputs", "InputBuilder"]


@dataclass
class BatchInputs:
    is_prefill: bool
    input_ids: torch.Tensor
    positions: torch.Tensor
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    cum_seqlens_q: torch.Tensor | None = None
    cum_seqlens_k: torch.Tensor | None = None
    slot_mapping: torch.Tensor | None = None
    context_lens: torch.Tensor | None = None
    block_tables: torch.Tensor | None = None
    num_seqs: int = 0


class InputBuilder:
    def __init__(self, block_size: int, max_num_seqs: int, max_model_len: int,
                 max_num_batched_tokens: int, pin_memory: bool = False) -> None:
        self.block_size = block_size
        self.max_num_seqs = max_num_seqs
        self.max_blocks_per_seq = -(-max_model_len // block_size)
        self.pin = bool(pin_memory)
        cap_tok = max(max_num_batched_tokens, max_num_seqs)
        # int64 plane: [input_ids | positions]; int32 plane: [slots | cu_q | cu_k / context_lens]
        self._i64 = self._alloc((2, cap_tok), torch.int64)
        self._slots = self._alloc((cap_tok,), torch.int32)
        self._seq32 = self._alloc((3, max_num_seqs + 1), torch.int32)
        self._tables = self._alloc((max_num_seqs, self.max_blocks_per_seq), torch.int32)
        self._np_i64 = self._i64.numpy()
        self._np_slots = self._slots.numpy()
        self._np_seq32 = self._seq32.numpy()
        self._np_tables = self._tables.numpy()

    def _alloc(self, shape, dtype) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, pin_memory=self.pin)

    # ------------------------------------------------------------------ prefill (a1)
    def prefill(self, sequences: list[Sequence]) -> BatchInputs:
        bs = self.block_size
        ids: list[int] = []
        pos: list[np.ndarray] = []
        slots: list[np.ndarray] = []
        cu_q, cu_k = [0], [0]
        max_q = max_k = 0
        for seq in sequences:
            n, cached = len(seq), seq.num_cached_tokens
            ids.extend(seq.token_ids[cached:])
            pos.append(np.arange(cached, n, dtype=np.int64))
            cu_q.append(cu_q[-1] + n - cached)
            cu_k.append(cu_k[-1] + n)
            max_q, max_k = max(max_q, n - cached), max(max_k, n)
            if seq.block_table:
                first, last = seq.num_cached_blocks, seq.num_blocks - 1
                for i in range(first, last + 1):
                    start = seq.block_table[i] * bs
                    count = bs if i != last else seq.last_block_num_tokens
                    slots.append(np.arange(start, start + count, dtype=np.int32))
        t = len(ids)
        out_ids = self._fit_i64(0, t)
        out_pos = self._fit_i64(1, t)
        out_ids.numpy()[:] = ids
        if t:
            out_pos.numpy()[:] = np.concatenate(pos)
        n = len(sequences)
        cq, ck = self._fit_seq32(0, n + 1), self._fit_seq32(1, n + 1)
        cq.numpy()[:] = cu_q
        ck.numpy()[:] = cu_k
        slot_t = None
        if slots:
            flat = np.concatenate(slots)
            slot_t = self._fit_slots(flat.shape[0])
            slot_t.numpy()[:] = flat
        tables = None
        if any(seq.block_table for seq in sequences):
            tables = self._tables_for(sequences, decode=False)
        return BatchInputs(True, out_ids, out_pos, max_q, max_k, cq, ck, slot_t, None, tables, n)

    # ------------------------------------------------------------------ decode (a2)
    def decode(self, sequences: list[Sequence]) -> BatchInputs:
        b, bs = len(sequences), self.block_size
        ids = self._fit_i64(0, b)
        pos = self._fit_i64(1, b)
        slot_t = self._fit_slots(b)
        ctx = self._fit_seq32(2, b)
        lens = np.fromiter((s.num_tokens for s in sequences), dtype=np.int64, count=b)
        ids.numpy()[:] = np.fromiter((s.last_token for s in sequences), dtype=np.int64, count=b)
        pos.numpy()[:] = lens - 1
        ctx.numpy()[:] = lens
        # slot of the token being written: last block, last occupied offset; 0 without a table (warm-up)
        slot_t.numpy()[:] = np.fromiter(
            ((s.block_table[-1] * bs + s.last_block_num_tokens - 1) if s.block_table else 0
             for s in sequences), dtype=np.int32, count=b)
        tables = self._tables_for(sequences, decode=True)
        return BatchInputs(False, ids, pos, 0, 0, None, None, slot_t, ctx, tables, b)

    # ------------------------------------------------------------------ block tables
    def _tables_for(self, sequences: list[Sequence], decode: bool) -> torch.Tensor:
        n = len(sequences)
        if n == 0:
            return torch.empty((0, 0), dtype=torch.int32)
        width = max(len(s.block_table) for s in sequences)
        if width == 0:
            return torch.empty((n, 0), dtype=torch.int32)
        if n <= self.max_num_seqs and width <= self.max_blocks_per_seq:
            view = self._np_tables[:n, :width]
            out = self._tables[:n, :width]  # row-strided view of the staging buffer, as in the reference
        else:  # oversize batch: one-off allocation (inference_executor.py:687-688)
            out = torch.empty((n, width), dtype=torch.int32, pin_memory=self.pin)
            view = out.numpy()
        view.fill(-1)
        for i, s in enumerate(sequences):
            t = s.block_table
            if t:
                view[i, :len(t)] = t
        return out

    # ------------------------------------------------------------------ staging slices
    def _fit_i64(self, plane: int, n: int) -> torch.Tensor:
        if n <= self._i64.shape[1]:
            return self._i64[plane, :n]
        return torch.empty((n,), dtype=torch.int64, pin_memory=self.pin)

    def _fit_slots(self, n: int) -> torch.Tensor:
        if n <= self._slots.shape[0]:
            return self._slots[:n]
        return torch.empty((n,), dtype=torch.int32, pin_memory=self.pin)

    def _fit_seq32(self, plane: int, n: int) -> torch.Tensor:
        if n <= self._seq32.shape[1]:
            return self._seq32[plane, :n]
        return torch.empty((n,), dtype=torch.int32, pin_memory=self.pin)
