"""Physical KV-cache block pool with hash-chained prefix caching (mirror of
minivllm/engine/block_manager.py:24-450).

Block ids, reference counts and digests must match the reference bit for bit — they
become the block tables and slot mappings the HIP kernels index with.  What fixes the ids:
  * the free list is a FIFO deque with lazy tombstones (an id stays in the deque after it is
    taken; stale entries are dropped only when they reach the front) plus a membership set;
  * a full block's digest is xxh64(prefix digest as 8 LE bytes (if any) || int32-LE tokens);
  * a freed block keeps its digest/tokens until it is handed out again (so it can be revived).
Method names, the deque-plus-tombstone pool and the order of its operations are the reference's by necessity (the ids
they produce are compared bit for bit with reference traces, tests/test_host_logic.py); the error texts are this
build's own, except the leading words of "No free blocks available", which DESIGN.md Q10 and the scheduler refer to.
"""
from __future__ import annotations

import struct
from collections import deque

import xxhash

from .sequence import Sequence

__all__ = ["Block", "BlockManager"]

NO_HASH = -1


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int) -> None:
        self.block_id = block_id
        self.ref_count = 0
        self.hash = NO_HASH
        self.token_ids: list[int] = []

    def update(self, hash_val: int, token_ids: list[int]) -> None:
        self.hash = hash_val
        self.token_ids = list(token_ids)

    def reset(self) -> None:
        self.ref_count = 1
        self.hash = NO_HASH
        self.token_ids.clear()


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int) -> None:
        if num_blocks <= 0:
            raise ValueError(f"a block pool needs at least one block (num_blocks={num_blocks})")
        if block_size <= 0:
            raise ValueError(f"a block holds at least one token (block_size={block_size})")
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self._free_set: set[int] = set(range(num_blocks))
        self.used_block_ids: set[int] = set()
        self.stats = {"total_allocations": 0, "cache_hits": 0, "cache_misses": 0}

    # -- digests (block_manager.py:128-152) ---------------------------------------------------
    @classmethod
    def compute_hash(cls, token_ids: list[int], prefix: int = -1) -> int:
        if not token_ids:
            raise ValueError("the digest of a block is taken over its tokens: got none")
        h = xxhash.xxh64()
        if prefix != -1:
            h.update(prefix.to_bytes(8, "little"))
        h.update(struct.pack(f"<{len(token_ids)}i", *token_ids))
        return h.intdigest()

    # -- pool mechanics -------------------------------------------------------------------------
    def _get_any_free_block(self) -> int:
        q = self.free_block_ids
        while q:
            if q[0] in self._free_set:
                return q[0]
            q.popleft()  # tombstone of a block that was taken since
        raise RuntimeError("No free blocks available: the KV pool is exhausted")  # (first words = the reference's)

    def _take(self, block_id: int, reset: bool = True) -> Block:
        if not 0 <= block_id < self.num_blocks:
            raise RuntimeError(f"block id {block_id} is outside the pool [0, {self.num_blocks})")
        if block_id not in self._free_set:
            raise RuntimeError(f"block {block_id} was asked for but is not free")
        self._free_set.remove(block_id)
        block = self.blocks[block_id]
        if block.ref_count != 0:
            raise RuntimeError(f"free block {block_id} still has {block.ref_count} reference(s)")
        if reset:
            if block.hash != NO_HASH and self.hash_to_block_id.get(block.hash) == block_id:
                del self.hash_to_block_id[block.hash]
            block.reset()
        else:  # reviving a cached block: keep digest and tokens
            block.ref_count = 1
        self.used_block_ids.add(block_id)
        self.stats["total_allocations"] += 1
        return block

    _allocate_block = _take  # reference-compatible name

    def _deallocate_block(self, block_id: int) -> None:
        if not 0 <= block_id < self.num_blocks:
            raise RuntimeError(f"block id {block_id} is outside the pool [0, {self.num_blocks})")
        block = self.blocks[block_id]
        if block.ref_count != 0:
            raise RuntimeError(f"block {block_id} is still referenced {block.ref_count} time(s): it cannot go back to the pool")
        self.used_block_ids.remove(block_id)
        self.free_block_ids.append(block_id)
        self._free_set.add(block_id)

    def get_num_free_blocks(self) -> int:
        return len(self._free_set)

    # -- prefill-time allocation (block_manager.py:252-344) -------------------------------------
    def can_allocate(self, sequence: Sequence) -> bool:
        return len(self._free_set) >= sequence.num_blocks

    def allocate(self, sequence: Sequence) -> None:
        if sequence.block_table:
            raise ValueError("this sequence already owns a block table")
        if len(self._free_set) < sequence.num_blocks:
            raise ValueError(f"prompt needs {sequence.num_blocks} blocks, the pool has {len(self._free_set)} free "
                             f"(callers test can_allocate first)")
        chain = NO_HASH
        for i in range(sequence.num_blocks):
            tokens = sequence.block(i)
            full = len(tokens) == self.block_size
            chain = self.compute_hash(tokens, chain) if full else NO_HASH  # partial blocks break the chain
            hit_id = self.hash_to_block_id.get(chain, -1) if chain != NO_HASH else -1
            hit = (chain != NO_HASH and 0 <= hit_id < self.num_blocks
                   and self.blocks[hit_id].token_ids == tokens)
            if hit:
                self.stats["cache_hits"] += 1
                sequence.num_cached_tokens += self.block_size
                if hit_id in self.used_block_ids:
                    self.blocks[hit_id].ref_count += 1
                else:
                    self._take(hit_id, reset=False)
                block_id = hit_id
            else:
                block_id = self._get_any_free_block()
                block = self._take(block_id)
                self.stats["cache_misses"] += 1
                if chain != NO_HASH:
                    block.update(chain, tokens)
                    self.hash_to_block_id[chain] = block_id
            sequence.block_table.append(block_id)
        if sequence.num_cached_tokens >= sequence.num_tokens:
            # every block hit (len % block_size == 0): the reference then prefills ZERO query tokens
            # and samples from another sequence's row.  Recompute the last block
            # instead, so that there is a last-token logit; the rewrite of the shared block is
            # byte-identical.
            sequence.num_cached_tokens -= self.block_size

    def deallocate(self, sequence: Sequence) -> None:
        for block_id in reversed(sequence.block_table):
            block = self.blocks[block_id]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._deallocate_block(block_id)
        sequence.num_cached_tokens = 0
        sequence.block_table.clear()

    # -- decode-time growth (block_manager.py:369-450) ------------------------------------------
    def can_append(self, sequence: Sequence) -> bool:
        """Reference predicate, kept verbatim for trace parity: a block is "needed" when
        len % block_size == 0 — one step BEFORE may_append actually takes it."""
        need = 1 if len(sequence) % self.block_size == 0 else 0
        return len(self._free_set) >= need

    def needs_block_now(self, sequence: Sequence) -> bool:
        """True when this step's may_append will take a block (len % block_size == 1)."""
        return len(sequence) % self.block_size == 1

    def may_append(self, sequence: Sequence) -> None:
        table = sequence.block_table
        last = self.blocks[table[-1]]
        pos = len(sequence) % self.block_size
        if pos == 1:  # first token of a new block: take one
            if last.hash == NO_HASH:
                raise RuntimeError(f"block {last.block_id} is full but its digest was never published")
            block_id = self._get_any_free_block()
            self._take(block_id)
            table.append(block_id)
        elif pos == 0:  # block just filled: publish its digest for prefix caching
            if last.hash != NO_HASH:
                raise RuntimeError(f"block {last.block_id} just filled up, yet it already carries a digest")
            tokens = sequence.block(sequence.num_blocks - 1)
            prefix = self.blocks[table[-2]].hash if len(table) > 1 else NO_HASH
            digest = self.compute_hash(tokens, prefix)
            last.update(digest, tokens)
            self.hash_to_block_id[digest] = last.block_id
        elif last.hash != NO_HASH:
            raise RuntimeError(f"block {last.block_id} is only partly filled, yet it carries a digest")
