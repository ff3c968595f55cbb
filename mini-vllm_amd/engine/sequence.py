"""One generation request and its paging bookkeeping (mirror of
minivllm/engine/sequence.py:31-365).  All quantities here are host integers that the
block manager / input builder turn into block tables and slot mappings."""
from __future__ import annotations

from enum import Enum, auto
from itertools import count

from ..sampling_params import SamplingParams

__all__ = ["Sequence", "SequenceStatus"]


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    counter = count()

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams | None = None,
                 block_size: int = 64) -> None:
        if not token_ids:
            raise ValueError("token_ids must contain at least one token")
        sp = sampling_params if sampling_params is not None else SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = self.num_tokens
        self.num_cached_tokens = 0
        # chunked prefill (beyond the reference): a prefill step computes tokens
        # [num_cached_tokens, prefill_end); None = up to the end of the sequence (the reference's only mode)
        self.prefill_end: int | None = None
        self.block_table: list[int] = []
        self.block_size = block_size
        self._adopt_sampling(sp)

    def _adopt_sampling(self, sp: SamplingParams) -> None:
        self.sampling_params = sp
        self.temperature, self.top_p, self.top_k = sp.temperature, sp.top_p, sp.top_k
        self.min_p, self.max_tokens, self.ignore_eos = sp.min_p, sp.max_tokens, sp.ignore_eos

    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    # -- derived quantities (sequence.py:120-201) -----------------------------------------
    @property
    def is_finished(self) -> bool:
        return self.status == SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[: self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    @property
    def prefill_stop(self) -> int:
        """End (exclusive) of the token range the next prefill step computes."""
        return self.num_tokens if self.prefill_end is None else self.prefill_end

    def block(self, i: int) -> list[int]:
        if not 0 <= i < self.num_blocks:
            raise IndexError(
                f"Block index {i} out of range [0, {self.num_blocks}). Sequence ID: {self.seq_id}, "
                f"Total tokens: {self.num_tokens}, Block size: {self.block_size}")
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    def append_token(self, token_id: int) -> None:
        if token_id < 0:
            raise ValueError(f"Token ID must be non-negative, got {token_id}. Sequence ID: {self.seq_id}")
        if self.status == SequenceStatus.FINISHED:
            raise RuntimeError(
                f"Cannot append token to finished sequence. Sequence ID: {self.seq_id}, Token ID: {token_id}")
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # -- compact pickling for the TP control plane (sequence.py:239-365): once a sequence
    #    is decoding, workers only need its last token, never the whole prompt again -------
    def __getstate__(self):
        payload = self.token_ids if self.num_completion_tokens == 0 else self.last_token
        return (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.block_table,
                self.block_size, payload, self.temperature, self.top_p, self.top_k, self.min_p,
                self.max_tokens, self.ignore_eos, self.prefill_end)

    def __setstate__(self, state) -> None:
        (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.block_table,
         self.block_size, payload, temperature, top_p, top_k, min_p, max_tokens, ignore_eos, prefill_end) = state
        decoding = self.num_tokens - self.num_prompt_tokens > 0
        if isinstance(payload, int):
            if decoding:
                self.token_ids = [0] * self.num_tokens
                self.token_ids[-1] = payload
            else:
                self.token_ids = [payload]
            self.last_token = payload
        else:
            self.token_ids = payload
            self.last_token = payload[-1] if payload else 0
        self.status = SequenceStatus.RUNNING if decoding else SequenceStatus.WAITING
        self.prefill_end = prefill_end   # a chunk of a long prompt: the workers must compute the same token range
        self._adopt_sampling(SamplingParams(temperature=temperature, top_p=top_p, top_k=top_k,
                                            min_p=min_p, max_tokens=max_tokens, ignore_eos=ignore_eos))
