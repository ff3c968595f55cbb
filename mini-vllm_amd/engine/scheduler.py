"""Two-phase FIFO batch scheduler (mirror of minivllm/engine/scheduler.py:55-293):
prefill has priority and packs whole prompts up to max_num_seqs / max_num_batched_tokens;
otherwise every running sequence decodes one token, preempting the newest when blocks run
out.  Preempted sequences lose their blocks and are re-prefilled from the front of the queue
(recompute; the prefix cache often rescues them).

Beyond the reference (BASELINE config 4 "chunked prefill"): a prompt longer than
max_num_batched_tokens — which the reference can never schedule — is prefilled ALONE in chunks of at most
max_num_batched_tokens tokens (whole blocks), one chunk per step, each attending the chunks before it through
the paged KV cache (the prefix-cached form of the prefill kernel).  Only the last chunk's token is kept.  Every
flow the reference supports schedules exactly as before (the host traces pin that).

Mixed batches (``enable_mixed_batches``, also beyond the reference): a prefill step additionally carries ONE token of
every running sequence — they come first in the batch, as "prompts" whose all-but-last tokens are cached
(``num_cached_tokens = len - 1``), which is exactly what a decode row is; the input builder and the attention layer
route such rows to the paged-decode kernel.  The projections then stream their weights once for prompt tokens and
decode rows together, and decoding no longer stalls while prompts (or the chunks of a long one) are prefilled.  The
token budget of the prompts shrinks by the number of decode rows."""
from __future__ import annotations

import logging
from collections import deque

from ..config import Config
from .block_manager import BlockManager
from .sequence import Sequence, SequenceStatus

__all__ = ["Scheduler"]
logger = logging.getLogger(__name__)


class Scheduler:
    def __init__(self, config: Config) -> None:
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        # a sequence stops when it fills max_model_len -- or the whole KV pool (alone in it, it could only
        # preempt and re-prefill itself for ever)
        self.max_model_len = min(getattr(config, "max_model_len", None) or (1 << 62),
                                 config.num_kvcache_blocks * config.kvcache_block_size)
        self.chunked_prefill = bool(getattr(config, "enable_chunked_prefill", False))
        self.mixed_batches = bool(getattr(config, "enable_mixed_batches", False))
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()
        self.prefilling: Sequence | None = None   # a long prompt between two of its chunks
        self.num_preemptions = 0

    def is_finished(self) -> bool:
        return not self.waiting and not self.running and self.prefilling is None

    def add(self, sequence: Sequence) -> None:
        self.waiting.append(sequence)

    def schedule(self) -> tuple[list[Sequence], bool]:
        if self.is_finished():
            raise RuntimeError("No sequences scheduled")
        if self.mixed_batches and self.running and self._prompt_could_join(len(self.running)):
            rows = self._schedule_decode()                  # the reference's decode rules (may preempt)
            prompts = self._schedule_prefill(len(rows)) if rows else []
            if prompts:
                for seq in rows:                            # a decode row = a prompt with one uncached token
                    seq.num_cached_tokens, seq.prefill_end = len(seq) - 1, None
                return rows + prompts, True
            if rows:
                return rows, False
        batch = self._schedule_prefill()
        if batch:
            return batch, True
        batch = self._schedule_decode()
        if batch:
            return batch, False
        if not self.is_finished():
            msg = (f"No sequences scheduled, but scheduler is not finished. State: "
                   f"waiting={len(self.waiting)}, running={len(self.running)}, "
                   f"max_num_seqs={self.max_num_seqs}, max_num_batched_tokens={self.max_num_batched_tokens}")
            if self.waiting:
                msg += f", first_waiting_seq_len={len(self.waiting[0])}"
            if self.running:
                msg += f", running_seq_lens={[len(s) for s in list(self.running)[:5]]}"
            raise RuntimeError(msg)
        return [], False

    def _next_chunk(self, seq: Sequence, used: int = 0) -> None:
        """Set the token range [num_cached_tokens, prefill_end) of the next chunk of a long prompt (``used``: tokens
        of the step already taken by decode rows)."""
        bs = self.block_manager.block_size
        budget = max(bs, (self.max_num_batched_tokens - used) // bs * bs)       # whole blocks
        end = seq.num_cached_tokens + budget
        seq.prefill_end = None if end >= len(seq) else end

    def _prompt_could_join(self, rows: int) -> bool:
        """Mixed batches: is there a prompt (or the next chunk of one) that fits beside ``rows`` decode rows?  Checked
        BEFORE the decode rows are scheduled (that step takes blocks); when the head of the queue only fits alone,
        the step is the reference's: that prompt first, by itself."""
        room = self.max_num_batched_tokens - rows
        if rows >= self.max_num_seqs:          # the step's rows (decode rows + prompts) never exceed max_num_seqs:
            return False                       # a chunk in flight waits for a step of its own like a new prompt does
        if self.prefilling is not None:
            return room >= self.block_manager.block_size
        if not self.waiting:
            return False
        head = self.waiting[0]
        if not self.block_manager.can_allocate(head):
            return False
        if len(head) <= room:
            return True
        return self.chunked_prefill and len(head) > self.max_num_batched_tokens and room >= self.block_manager.block_size

    def _schedule_prefill(self, used: int = 0) -> list[Sequence]:
        """``used``: decode rows already in the step (mixed batches): they count against both budgets."""
        if self.prefilling is not None:                                 # the next chunk of the prompt in flight
            self._next_chunk(self.prefilling, used)
            return [self.prefilling]
        batch: list[Sequence] = []
        tokens = used
        if (self.chunked_prefill and self.waiting and len(self.waiting[0]) > self.max_num_batched_tokens
                and self.block_manager.can_allocate(self.waiting[0])):
            seq = self.waiting.popleft()
            self.block_manager.allocate(seq)                            # (prefix-cache hits shorten the first chunk)
            seq.status = SequenceStatus.RUNNING
            self.prefilling = seq
            self._next_chunk(seq, used)
            return [seq]
        while self.waiting and len(batch) + used < self.max_num_seqs:
            seq = self.waiting[0]
            # the budget test uses the full length; the charge below only the uncached part
            if tokens + len(seq) > self.max_num_batched_tokens or not self.block_manager.can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            tokens += len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            batch.append(seq)
        return batch

    def _can_step(self, seq: Sequence) -> bool:
        bm = self.block_manager
        if not bm.can_append(seq):
            return False
        # beyond the reference: never walk into may_append's "No free blocks available"
        return not (bm.needs_block_now(seq) and bm.get_num_free_blocks() == 0)

    def _schedule_decode(self) -> list[Sequence]:
        fast = self._schedule_decode_fast()
        if fast is not None:
            return fast
        batch: list[Sequence] = []
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            while not self._can_step(seq):
                if self.running:
                    self.preempt(self.running.pop())  # newest first
                else:
                    self.preempt(seq)
                    break
            else:
                self.block_manager.may_append(seq)
                batch.append(seq)
        if batch:
            self.running.extendleft(reversed(batch))
        return batch

    def _schedule_decode_fast(self) -> list[Sequence] | None:
        """Steady-state decode without the per-sequence deque shuffle.  When every running sequence
        fits the batch and the free pool covers every block that can be asked for this step, the
        reference loop degenerates to "may_append each sequence in order" with the queue order
        unchanged — same batch, same block ids, same digests (checked against the reference traces)."""
        running = self.running
        if not running or len(running) > self.max_num_seqs:
            return None
        bm = self.block_manager
        bs = bm.block_size
        pos = [s.num_tokens % bs for s in running]
        takes = pos.count(1)                      # may_append takes a block at len % bs == 1
        reserve = 1 if 0 in pos else 0            # can_append wants a free block at len % bs == 0
        if bm.get_num_free_blocks() < takes + reserve:
            return None                           # preemption possible: take the general path
        batch = list(running)
        if takes or reserve:
            for s, p in zip(batch, pos):
                if p <= 1:
                    bm.may_append(s)              # p == 1: new block; p == 0: publish the digest
        return batch

    def preempt(self, sequence: Sequence) -> None:
        logger.info("Preempting sequence %d due to memory constraints.", sequence.seq_id)
        self.num_preemptions += 1
        sequence.status = SequenceStatus.WAITING
        self.block_manager.deallocate(sequence)
        self.waiting.appendleft(sequence)

    def postprocess(self, sequences: list[Sequence], token_ids: list[int]) -> None:
        chunk = self.prefilling if any(s is self.prefilling for s in sequences) else None
        if chunk is not None:
            if chunk.prefill_end is not None:        # a middle chunk: its KV is in the cache, its token is not a token
                chunk.num_cached_tokens = chunk.prefill_end
                keep = [(s, t) for s, t in zip(sequences, token_ids, strict=True) if s is not chunk]
                sequences, token_ids = [s for s, _ in keep], [t for _, t in keep]
            else:
                self.prefilling = None               # last chunk: from here on an ordinary running sequence
                self.running.append(chunk)
        for seq, token in zip(sequences, token_ids, strict=True):
            seq.append_token(token)
            hit_eos = token == self.eos and not seq.ignore_eos
            # beyond the reference: a sequence that fills max_model_len stops there (the reference walks off the
            # end of its block-table staging buffer with an index error)
            if hit_eos or seq.num_completion_tokens >= seq.max_tokens or len(seq) >= self.max_model_len:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
