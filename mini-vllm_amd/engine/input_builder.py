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
    # decode fast-path hints for the executor (values above are always complete and correct)
    packed: bool = False          # ids/positions/slots/context_lens are views of InputBuilder.decode_pack
    tables_dirty: bool = True     # block-table staging changed since the previous decode batch
    batch_changed: bool = True    # different sequences (or order) than the previous decode batch
    # mixed batch (prefill step whose first ``decode_rows`` sequences are decode rows; utils/context.py)
    decode_rows: int = 0
    rest_cum_seqlens_q: torch.Tensor | None = None
    rest_cum_seqlens_k: torch.Tensor | None = None
    rest_max_seqlen_q: int = 0
    rest_max_seqlen_k: int = 0


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
        self._seq32 = self._alloc((5, max_num_seqs + 1), torch.int32)
        self._tables = self._alloc((max_num_seqs, self.max_blocks_per_seq), torch.int32)
        self._np_i64 = self._i64.numpy()
        self._np_slots = self._slots.numpy()
        self._np_seq32 = self._seq32.numpy()
        self._np_tables = self._tables.numpy()
        # decode pack: ONE pinned buffer -> one H2D copy per step.
        # layout (bytes): ids i64[ms] | positions i64[ms] | slots i32[ms] | context_lens i32[ms] | rng i64[2]
        ms = max_num_seqs
        self.pack_layout = {"ids": (0, ms * 8), "pos": (ms * 8, ms * 8), "slots": (ms * 16, ms * 4),
                            "ctx": (ms * 20, ms * 4), "rng": (ms * 24, 16)}
        self.pack_nbytes = ms * 24 + 16
        self.decode_pack = torch.zeros(self.pack_nbytes, dtype=torch.uint8, pin_memory=self.pin)
        self._p_ids = self.pack_view(self.decode_pack, "ids")
        self._p_pos = self.pack_view(self.decode_pack, "pos")
        self._p_slots = self.pack_view(self.decode_pack, "slots")
        self._p_ctx = self.pack_view(self.decode_pack, "ctx")
        self.p_rng = self.pack_view(self.decode_pack, "rng")
        self._last_batch: list | None = None   # decode batch whose block tables are in the staging buffer
        self._last_table_lens: list[int] = []
        self._last_width = -1

    def pack_view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        off, n = self.pack_layout[name]
        return buf[off:off + n].view(torch.int32 if name in ("slots", "ctx") else torch.int64)

    def _alloc(self, shape, dtype) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, pin_memory=self.pin)

    # ------------------------------------------------------------------ prefill (a1)
    def prefill(self, sequences: list[Sequence]) -> BatchInputs:
        self._last_batch = None  # the table staging buffer is about to be reused
        bs = self.block_size
        ids: list[int] = []
        pos: list[np.ndarray] = []
        slots: list[np.ndarray] = []
        cu_q, cu_k = [0], [0]
        max_q = max_k = 0
        for seq in sequences:
            # n = end of the range this step computes: the whole sequence, or a chunk boundary (a multiple of the
            # block size) of a prompt that is prefilled in several steps (Sequence.prefill_end)
            n, cached = seq.prefill_stop, seq.num_cached_tokens
            ids.extend(seq.token_ids[cached:n])
            pos.append(np.arange(cached, n, dtype=np.int64))
            cu_q.append(cu_q[-1] + n - cached)
            cu_k.append(cu_k[-1] + n)
            max_q, max_k = max(max_q, n - cached), max(max_k, n)
            if seq.block_table:
                # slots of tokens [cached, n): whole blocks for the reference's flows (cached is a multiple of the
                # block size there); a decode row of a mixed batch starts inside its last block
                first, last = cached // bs, -(-n // bs) - 1
                for i in range(first, last + 1):
                    lo, hi = max(cached, i * bs), min(n, (i + 1) * bs)
                    start = seq.block_table[i] * bs + lo - i * bs
                    slots.append(np.arange(start, start + hi - lo, dtype=np.int32))
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
        out = BatchInputs(True, out_ids, out_pos, max_q, max_k, cq, ck, slot_t, None, tables, n)
        # leading sequences that add exactly one token to a cached prefix are decode rows (mixed batches)
        nd = 0
        while nd < n and sequences[nd].block_table and sequences[nd].num_cached_tokens > 0 \
                and sequences[nd].prefill_stop - sequences[nd].num_cached_tokens == 1:
            nd += 1
        if nd:
            ctx = self._fit_seq32(2, nd)
            ctx.numpy()[:] = [s.prefill_stop for s in sequences[:nd]]
            out.decode_rows, out.context_lens = nd, ctx
            rq, rk = self._fit_seq32(3, n - nd + 1), self._fit_seq32(4, n - nd + 1)
            rq.numpy()[:] = np.asarray(cu_q[nd:], dtype=np.int64) - cu_q[nd]
            rk.numpy()[:] = np.asarray(cu_k[nd:], dtype=np.int64) - cu_k[nd]
            out.rest_cum_seqlens_q, out.rest_cum_seqlens_k = rq, rk
            rest = sequences[nd:]
            out.rest_max_seqlen_q = max((s.prefill_stop - s.num_cached_tokens for s in rest), default=0)
            out.rest_max_seqlen_k = max((s.prefill_stop for s in rest), default=0)
        return out

    # ------------------------------------------------------------------ decode (a2)
    def decode(self, sequences: list[Sequence], pad_to: int | None = None) -> BatchInputs:
        """``pad_to``: graph bucket size; rows b..pad_to of the pack get slot -1 / context 0 (no KV
        store, empty attention) like the reference's replay padding (inference_executor.py:758-764)."""
        b, bs = len(sequences), self.block_size
        packed = b <= self.max_num_seqs
        if packed:
            ids, pos, slot_t, ctx = self._p_ids[:b], self._p_pos[:b], self._p_slots[:b], self._p_ctx[:b]
        else:
            ids, pos = self._fit_i64(0, b), self._fit_i64(1, b)
            slot_t, ctx = self._fit_slots(b), self._fit_seq32(2, b)
        lens = np.fromiter((s.num_tokens for s in sequences), dtype=np.int64, count=b)
        ids.numpy()[:] = np.fromiter((s.last_token for s in sequences), dtype=np.int64, count=b)
        pos.numpy()[:] = lens - 1
        ctx.numpy()[:] = lens
        # slot of the token being written = last block * bs + (len-1) % bs  (== last_block_num_tokens - 1);
        # 0 without a table (warm-up)
        last_blk = np.fromiter((s.block_table[-1] if s.block_table else -1 for s in sequences),
                               dtype=np.int64, count=b)
        slot_t.numpy()[:] = np.where(last_blk >= 0, last_blk * bs + (lens - 1) % bs, 0)
        if packed and pad_to is not None and pad_to > b:
            self._p_slots.numpy()[b:pad_to] = -1
            self._p_ctx.numpy()[b:pad_to] = 0
        last = self._last_batch
        same = (packed and last is not None and len(last) == b
                and all(x is y for x, y in zip(sequences, last)))
        tables, dirty = self._tables_decode(sequences, same)
        self._last_batch = list(sequences) if packed else None
        return BatchInputs(False, ids, pos, 0, 0, None, None, slot_t, ctx, tables, b,
                           packed=packed, tables_dirty=dirty, batch_changed=not same)

    def _tables_decode(self, sequences: list[Sequence], same: bool):
        """Block tables of a decode batch; when the batch is the previous one, only rows whose table
        grew are rewritten (a block is appended once every block_size steps)."""
        lens = [len(s.block_table) for s in sequences]
        width = max(lens) if lens else 0
        if (not same or width != self._last_width or width == 0
                or len(sequences) > self.max_num_seqs or width > self.max_blocks_per_seq):
            out = self._tables_for(sequences, decode=True)
            self._last_table_lens, self._last_width = lens, width
            return out, True
        dirty = False
        view = self._np_tables
        for i, (n, old) in enumerate(zip(lens, self._last_table_lens)):
            if n != old:
                view[i, :n] = sequences[i].block_table
                dirty = True
        self._last_table_lens = lens
        return self._tables[:len(sequences), :width], dirty

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
