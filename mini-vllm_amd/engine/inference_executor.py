"""Batch execution on one GPU: KV-cache sizing/allocation, input shipping, eager or hipGraph
model execution, sampling (mirror of minivllm/engine/inference_executor.py:35-943).

MI355X-first differences from the reference:
  * KV cache sized for 288 GB HBM3E with the LOCAL kv-head count (the reference uses the total
    head count even under TP) — layout [2, L, num_blocks, block_size, Hkv/tp, D];
  * the decode step — model forward, LM head AND the sampler — is ONE captured hipGraph per
    batch-size bucket; every per-step quantity (ids, positions, slots, context lens, block
    tables, sampling params, RNG offset) lives in static device buffers that are refreshed by
    a handful of async copies; the only host sync of a step is reading back the B token ids;
  * the reference's graph path is dead code (row-0 broadcast bug, fp32 output buffer, host syncs
    in the attention fallback: SURVEY §9 Q2); here graph replay is checked bit-identical
    against eager execution.
"""
from __future__ import annotations

import logging
import os

import torch

from ..config import Config
from ..layers.sampler import Sampler
from ..parallel_state import (decode_step_is_native, get_custom_all_reduce as get_custom_allreduce,
                              get_tensor_parallel_world_size, tp_candidate_tokens, tp_greedy_tokens, tp_mass_tokens,
                              tp_nucleus_tokens)
from ..utils.context import reset_context, set_context
from .. import hip_ops
from .input_builder import BatchInputs, InputBuilder
from .model_manager import resolve_dtype
from .sequence import Sequence

logger = logging.getLogger(__name__)
__all__ = ["InferenceExecutor", "graph_batch_sizes", "kv_block_bytes"]


def graph_batch_sizes(max_batch_size: int) -> list[int]:
    """Decode graph buckets: 1,2,4,8,12 then every 16 up to min(max, 512)
    (inference_executor.py:839-841) -- plus the maximum itself when it is not on that grid (the reference's
    list leaves batches between the last multiple of 16 and max_num_seqs without a graph)."""
    top = min(max(max_batch_size, 1), 512)
    sizes = [s for s in [1, 2, 4, 8, 12] + list(range(16, top + 1, 16)) if s <= top]
    if sizes[-1] != top:
        sizes.append(top)
    return sizes


def kv_block_bytes(num_layers: int, block_size: int, num_kv_heads_local: int, head_dim: int, itemsize: int) -> int:
    return 2 * num_layers * block_size * num_kv_heads_local * head_dim * itemsize


class InferenceExecutor:
    def __init__(self, config: Config, model, device: torch.device | None = None) -> None:
        self.config = config
        self.model = model
        self.device = device if device is not None else next(model.parameters()).device
        self.block_size = config.kvcache_block_size
        hf = config.hf_config
        tp = self.tp_size = get_tensor_parallel_world_size()
        total_kv = getattr(hf, "num_key_value_heads", getattr(hf, "num_attention_heads", 1))
        self.num_kv_heads = max(total_kv // tp, 1)
        self.head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        self.num_layers = hf.num_hidden_layers
        try:
            self.dtype = next(model.parameters()).dtype
        except (StopIteration, AttributeError):
            self.dtype = resolve_dtype(config)
        self.enforce_eager = config.enforce_eager
        self.kv_cache: torch.Tensor | None = None
        self.sampler: Sampler | None = None
        self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
        self.greedy_graphs: dict[int, torch.cuda.CUDAGraph] = {}   # TP > 1: per-rank argmax instead of the logit gather
        self.cand_graphs: dict[int, torch.cuda.CUDAGraph] = {}     # TP > 1: per-rank top-64 candidates (sampled rows with 1 <= top_k <= 64)
        self.mass_graphs: dict[int, torch.cuda.CUDAGraph] = {}     # TP > 1: two-stage draw (rows sampled from the whole vocabulary)
        self._mass_capture = "no"
        self.nuc_graphs: dict[int, torch.cuda.CUDAGraph] = {}      # TP > 1: cross-rank nucleus / min-p floor + two-stage draw
        self._nuc_capture = "no"                                   # "no" | "lazy" | "done" (backend collectives inside)
        self._cand_capture = "no"                                  # "no" | "lazy" | "done"
        self.graph_bs: list[int] = []
        self.graph_vars: dict[str, torch.Tensor] = {}
        self.graph_pool = None
        self._general_capture = "no"   # "done" | "lazy" (first sampled batch under TP) | "no"
        self.total_tokens_generated = self.total_prefill_tokens = self.total_decode_tokens = 0
        self.inference_count = 0
        self.max_blocks_per_seq = -(-config.max_model_len // self.block_size)
        self.builder = InputBuilder(self.block_size, config.max_num_seqs, config.max_model_len,
                                    config.max_num_batched_tokens, pin_memory=self.device.type == "cuda")
        ms = config.max_num_seqs
        # device mirror of the builder's decode pack (ids | positions | slots | context_lens | rng): the whole
        # per-step integer state arrives with ONE async copy; the typed views below alias it
        b_ = self.builder
        self._pack_dev = torch.zeros(b_.pack_nbytes, dtype=torch.uint8, device=self.device)
        self._dev_ids = b_.pack_view(self._pack_dev, "ids")
        self._dev_pos = b_.pack_view(self._pack_dev, "pos")
        self._dev_slots = b_.pack_view(self._pack_dev, "slots")
        self._dev_ctx = b_.pack_view(self._pack_dev, "ctx")
        self._dev_rng = b_.pack_view(self._pack_dev, "rng")
        self._dev_tables = torch.full((ms, self.max_blocks_per_seq), -1, dtype=torch.int32, device=self.device)
        self._samp_rows = -1  # rows of the device sampling planes that are current (-1: stale)
        # speculative next-step replay (see _replay_graph): set by the engine when nothing is waiting
        self.speculate_next = False
        self.allow_speculation = os.environ.get("MV_DECODE_SPECULATE", "1") != "0"
        self._spec = None            # (batch, bucket, expected lens, event, host token slot) of the step in flight
        self._tok_host = [torch.zeros(ms, dtype=torch.int64, pin_memory=self.device.type == "cuda") for _ in range(3)]
        self._ar_slots: list = [None] * 3   # custom all-reduce status slot posted with each token copy
        self._tok_events = [torch.cuda.Event() for _ in range(3)] if self.device.type == "cuda" else []
        self._tok_slot = 0
        self.spec_hits = self.spec_misses = 0
        # sampling parameters: one pinned plane, one device plane  [temps | top_ps | min_ps] f32, top_ks i64
        pin = self.device.type == "cuda"
        self._samp_f32 = torch.zeros((3, ms), dtype=torch.float32, pin_memory=pin)
        self._samp_i64 = torch.zeros((ms,), dtype=torch.int64, pin_memory=pin)
        self._dev_samp_f32 = torch.zeros((3, ms), dtype=torch.float32, device=self.device)
        self._dev_samp_i64 = torch.zeros((ms,), dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ set-up
    def initialize(self, max_num_batched_tokens: int, max_num_seqs: int) -> None:
        self._allocate_kv_cache()
        self._initialize_sampler()
        self._optimize_model()
        self._warmup_model(max_num_batched_tokens, max_num_seqs)

    def _calculate_num_kv_blocks(self) -> int:
        """blocks = min(avail // block_bytes, max_num_seqs * ceil(max_model_len / bs)) with
        avail = total*util - used - peak + current (inference_executor.py:245-318)."""
        cfg = self.config
        free, total = torch.cuda.mem_get_info(self.device)
        stats = torch.cuda.memory_stats(self.device)
        peak = stats.get("allocated_bytes.all.peak", 0)
        current = stats.get("allocated_bytes.all.current", 0)
        block_bytes = kv_block_bytes(self.num_layers, self.block_size, self.num_kv_heads, self.head_dim,
                                     self.dtype.itemsize)
        avail = int(total * cfg.device_memory_utilization - (total - free) - peak + current)
        if avail <= 0:
            raise RuntimeError(f"Insufficient memory for KV cache allocation. Device: {self.device.type}, "
                               f"Available: {avail} bytes, Block size: {block_bytes} bytes")
        cap = cfg.max_num_seqs * self.max_blocks_per_seq
        n = min(avail // block_bytes, cap)
        if get_tensor_parallel_world_size() > 1:
            # every rank runs the same scheduler: they must agree on the pool size
            import torch.distributed as dist

            t = torch.tensor([n], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n = int(t.item())
        if n <= 0:
            raise ValueError(f"Calculated KV cache blocks ({n}) must be > 0. Available memory: {avail} bytes, "
                             f"Block size: {block_bytes} bytes")
        return int(n)

    def _allocate_kv_cache(self) -> None:
        n = self.config.num_kvcache_blocks
        if n <= 0:
            n = self._calculate_num_kv_blocks()
            self.config.num_kvcache_blocks = n  # the scheduler's BlockManager is sized from this
        shape = (2, self.num_layers, n, self.block_size, self.num_kv_heads, self.head_dim)
        try:
            self.kv_cache = torch.empty(shape, device=self.device, dtype=self.dtype)
        except torch.cuda.OutOfMemoryError as e:
            raise RuntimeError(f"OOM during KV cache allocation. Try reducing device_memory_utilization "
                               f"(current: {self.config.device_memory_utilization}) or max_num_seqs. Error: {e}") from e
        self._assign_kv_cache_to_layers()
        logger.info("Allocated KV cache: %.2f GB (%d blocks x %d tokens)",
                    self.kv_cache.numel() * self.kv_cache.element_size() / 2**30, n, self.block_size)

    def _assign_kv_cache_to_layers(self) -> None:
        layer = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache, module.v_cache = self.kv_cache[0, layer], self.kv_cache[1, layer]
                layer += 1
        if layer == 0:
            logger.warning("No attention layers found with k_cache/v_cache attributes.")

    def _initialize_sampler(self) -> None:
        self.sampler = Sampler(seed=self.config.seed)

    def _optimize_model(self) -> None:
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    @staticmethod
    def warmup_sequences(max_num_batched_tokens: int, max_num_seqs: int, max_model_len: int) -> int:
        """The reference's count (inference_executor.py:373-410), but never 0: with a token budget below
        max_model_len / 2 (chunked prefill: 16384 against 40960) its quotient is 0 and the warm-up -- and the
        library set-up behind it -- would be skipped although the 16-token dummies always fit the budget."""
        return max(1, min(max_num_batched_tokens // max(max_model_len // 2, 1), max_num_seqs, 2))

    def _warmup_model(self, max_num_batched_tokens: int, max_num_seqs: int) -> None:
        """2 x 16-token dummy prefill + one decode with EMPTY block tables
        (inference_executor.py:373-410): exercises the no-slot / zero-column paths."""
        mml = self.config.max_model_len
        n = self.warmup_sequences(max_num_batched_tokens, max_num_seqs, mml)
        dummy = [Sequence([0] * min(16, mml), block_size=self.block_size) for _ in range(n)]
        if not dummy:
            return
        self.execute_batch(dummy, prefill=True)
        for s in dummy:
            s.append_token(0)
        self.execute_batch(dummy, prefill=False)
        # the library GEMM's one-time set-up (hipBLASLt handle, workspace, kernel database: ~170 ms) belongs here, not
        # in the first real prefill step: with every decode-sized projection on the hand-written kernels nothing
        # else would call it before that step
        if self.device.type == "cuda":
            for mod in self.model.modules():
                w = getattr(mod, "weight", None)
                if w is not None and w.dim() == 2 and hasattr(mod, "input_size"):   # a LinearBase projection
                    torch.nn.functional.linear(torch.zeros(1024, w.shape[1], dtype=w.dtype, device=w.device), w)
                    break
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ one step
    @torch.inference_mode()
    def execute_batch(self, sequences: list[Sequence], prefill: bool = True):
        """-> (logits [n, V], next_tokens list[int]).  Under tensor parallelism an all-greedy batch is
        decoded from per-rank argmaxes (parallel_state.tp_merge_greedy) and ``logits`` is then this rank's
        vocab shard [n, V/tp]: the full-logit all-gather is the largest message of a decode step."""
        if not sequences:
            return torch.empty(0, device=self.device), []
        try:
            eager = True
            if prefill:
                self._drop_speculation()  # a speculative decode step in flight (if any) is simply ignored
                inp = self.builder.prefill(sequences)
                ids, pos = self._ship_prefill(inp, sequences)
                mode = self._tp_mode(sequences)
                logits = self.model.compute_logits(self.model(ids, pos), gather=mode == "gather")
                tokens = self._sample_tokens(logits, sequences, mode)
            else:
                bs = self._graph_bucket(sequences)
                inp = self.builder.decode(sequences, pad_to=bs)
                if bs is not None and inp.packed:
                    logits, tokens = self._replay_graph(bs, inp, sequences)
                    eager = False
                else:
                    self._drop_speculation()
                    ids, pos = self._ship_decode(inp)
                    mode = self._tp_mode(sequences, decode=True)
                    logits = self.model.compute_logits(self.model(ids, pos), gather=mode == "gather")
                    tokens = self._sample_tokens(logits, sequences, mode)
            self._check_peer_waits(eager)
            self._check_persistent_launch()
            self._update_metrics(sequences, prefill)
            self.inference_count += 1
            return logits, tokens
        finally:
            reset_context()

    def _check_peer_waits(self, eager: bool) -> None:
        """Tensor parallel on the peer-to-peer all-reduce: a wait that gave up (stalled / dead rank) left this rank
        with sums over a subset of the ranks.  The kernel's status block came back with this step's tokens
        (graph path: _stash_tokens / _replay_graph); eager steps read it here, after their tokens were read (the
        stream is idle then).  Raises custom_allreduce.PeerWaitExpired: the engine must not keep serving."""
        ar = get_custom_allreduce() if self.tp_size > 1 else None
        if ar is None or not ar.ready:
            return
        if eager:
            slot = ar.post_status()
            torch.cuda.current_stream().synchronize()
            ar.verify(slot)

    def _check_persistent_launch(self) -> None:
        """Small decode steps run every layer in one persistent launch whose grid barriers are bounded waits
        (csrc/decode_persistent.hip): a wait that gave up means the step's activations are garbage.  The status word came
        back with the step (glue/qwen.py copies it behind the launch); the tokens have been read, so it is current."""
        inner = getattr(self.model, "model", None)
        fn = getattr(inner, "persistent_give_ups", None)
        if fn is not None and fn():
            w = inner.persistent_status()
            where = "the start-up census" if w[1] >> 16 == 0xFFFF else f"grid barrier #{w[1] >> 16}"
            raise RuntimeError(f"persistent decode launch: a bounded wait gave up after 2 s at {where} on workgroup "
                               f"{w[1] & 0xFFFF} (waiting for {w[2]}, last saw {w[3]}): workgroups not co-resident or a "
                               "device fault; the results of this step are invalid.  MV_PERSISTENT=0 selects the "
                               "multi-launch path")

    def _to_dev(self, t: torch.Tensor | None):
        return None if t is None else t.to(self.device, non_blocking=True)

    def _ship_prefill(self, inp: BatchInputs, sequences):
        set_context(True, inp.max_seqlen_q, inp.max_seqlen_k, self._to_dev(inp.cum_seqlens_q),
                    self._to_dev(inp.cum_seqlens_k), self._to_dev(inp.slot_mapping),
                    self._to_dev(inp.context_lens) if inp.decode_rows else None,
                    self._to_dev(inp.block_tables),
                    prefix_cached=any(s.num_cached_tokens > 0 for s in sequences),
                    decode_rows=inp.decode_rows, rest_cum_seqlens_q=self._to_dev(inp.rest_cum_seqlens_q),
                    rest_cum_seqlens_k=self._to_dev(inp.rest_cum_seqlens_k),
                    rest_max_seqlen_q=inp.rest_max_seqlen_q, rest_max_seqlen_k=inp.rest_max_seqlen_k)
        return self._to_dev(inp.input_ids), self._to_dev(inp.positions)

    def _ship_decode(self, inp: BatchInputs):
        b = inp.num_seqs
        if not inp.packed:  # oversize batch: one-off tensors (inference_executor.py:687-688)
            set_context(False, 0, 0, None, None, self._to_dev(inp.slot_mapping), self._to_dev(inp.context_lens),
                        self._to_dev(inp.block_tables))
            return self._to_dev(inp.input_ids), self._to_dev(inp.positions)
        self._pack_dev.copy_(self.builder.decode_pack, non_blocking=True)
        tables = self._ship_tables(inp)
        set_context(False, 0, 0, None, None, self._dev_slots[:b], self._dev_ctx[:b], tables)
        return self._dev_ids[:b], self._dev_pos[:b]

    def _ship_tables(self, inp: BatchInputs):
        b, w = inp.num_seqs, inp.block_tables.shape[1]
        if w == 0:
            return torch.empty((b, 0), dtype=torch.int32, device=self.device)
        view = self._dev_tables[:b, :w]  # row-strided view, like the reference
        if inp.tables_dirty:
            view.copy_(inp.block_tables, non_blocking=True)
        return view

    def _ship_sampling(self, sequences, b_pad: int | None = None, unchanged: bool = False):
        b = len(sequences)
        n_rows = b if b_pad is None else b_pad
        d = self._dev_samp_f32
        if unchanged and self._samp_rows == n_rows:  # same batch as the previous decode step
            return d[0, :n_rows], self._dev_samp_i64[:n_rows], d[1, :n_rows], d[2, :n_rows]
        self._samp_rows = n_rows if b_pad is not None else -1
        f, k = self._samp_f32.numpy(), self._samp_i64.numpy()
        f[0, :b] = [s.temperature for s in sequences]
        f[1, :b] = [s.top_p for s in sequences]
        f[2, :b] = [s.min_p for s in sequences]
        k[:b] = [s.top_k for s in sequences]
        n = b if b_pad is None else b_pad
        if n > b:  # padded graph rows: greedy, cheapest path
            f[0, b:n], f[1, b:n], f[2, b:n], k[b:n] = 0.0, 1.0, 0.0, -1
        self._dev_samp_f32[:, :n].copy_(self._samp_f32[:, :n], non_blocking=True)
        self._dev_samp_i64[:n].copy_(self._samp_i64[:n], non_blocking=True)
        d = self._dev_samp_f32
        return d[0, :n], self._dev_samp_i64[:n], d[1, :n], d[2, :n]

    def _tp_greedy(self, sequences) -> bool:
        return self.tp_size > 1 and self.sampler is not None and all(s.temperature == 0 for s in sequences)

    def _tp_mode(self, sequences, decode: bool = False) -> str:
        """How a tensor-parallel batch gets its tokens: 'greedy' (per-rank argmax merge), 'cand' (per-rank top-64
        candidates: every row greedy or with 1 <= top_k <= 64, 16-bit logits), 'mass' (two-stage draw: every row
        greedy or sampled from the whole vocabulary — top_k off, top_p = 1, min_p = 0, the default SamplingParams) or
        'nuc' (top_k off everywhere, nucleus / min-p rows present: cross-rank floor, then the two-stage draw) or
        'gather' (all-gather of the [B, V/tp] logits: everything else, i.e. rows with 64 < top_k < vocab).
        TP = 1: 'gather' (the plain sampler on the full row).  ``decode``: the batch is a decode step that may replay a
        graph -- a 'nuc' batch whose graph family cannot exist (backend not capturable, or its capture failed) while
        the gathered-logits family does is a 'gather' batch: a replayed gather beats an eager step with three extra
        backend all-reduces.  The eager path, ``compute_logits(gather=...)`` and ``_replay_graph`` all ask here."""
        if self.tp_size == 1 or self.sampler is None:
            return "gather"
        if all(s.temperature == 0 for s in sequences):
            return "greedy"
        if self.dtype in (torch.bfloat16, torch.float16) and self.config.hf_config.vocab_size <= 1 << 24 and all(
                s.temperature == 0 or 0 < s.top_k <= hip_ops.TOPC for s in sequences):
            return "cand"
        vocab = self.config.hf_config.vocab_size
        if self.dtype in (torch.bfloat16, torch.float16) and vocab <= 1 << 24 and self.tp_size <= 16 and all(
                s.temperature == 0 or ((s.top_k <= 0 or s.top_k >= vocab) and s.top_p >= 1.0 and s.min_p <= 0.0)
                for s in sequences):
            return "mass"       # the default SamplingParams: the whole vocabulary, an exact two-stage draw
        if self.dtype in (torch.bfloat16, torch.float16) and vocab <= 1 << 24 and self.tp_size <= 16 and all(
                s.temperature == 0 or s.top_k <= 0 or s.top_k >= vocab for s in sequences):
            if decode and not self._nuc_family_usable():
                return "gather"
            return "nuc"        # top_k off, some rows with top_p < 1 / min_p > 0: the ranks find the floor together
        return "gather"

    def _nuc_family_usable(self) -> bool:
        """False when a decode step in 'nuc' mode would run eagerly although a gathered-logits graph could serve it."""
        if self.enforce_eager or self.nuc_graphs or self._nuc_capture == "lazy":
            return True
        return not (self.graphs or self._general_capture == "lazy")

    def _cand_tokens(self, logits_local, t, k, p, m):
        """'cand' mode, eager: the sampler (one Philox step, like any other batch) on the gathered candidates."""
        return tp_candidate_tokens(logits_local, lambda cl: self.sampler(cl, temperatures=t, top_ks=k, top_ps=p, min_ps=m),
                                   temperatures=t.float().contiguous())

    def _mass_tokens(self, logits_local, t, k, p, m):
        """'mass' mode, eager: the sampler on this rank's shard (one Philox step), then the 16-byte-per-row exchange."""
        def local(ll):
            return self.sampler(ll, temperatures=t, top_ks=k, top_ps=p, min_ps=m, return_stats=True)
        toks = tp_mass_tokens(logits_local, local, t.float().contiguous(), self.sampler.seed, self.sampler.calls + 1,
                              self.sampler.offset_dev)
        return toks

    def _nuc_tokens(self, logits_local, t, k, p, m):
        """'nuc' mode, eager: the cross-rank floor, then the sampler (filters off, one Philox step) on the masked shard
        and the 16-byte-per-row exchange."""
        def local(ll):
            return self.sampler(ll, temperatures=t, top_ks=None, top_ps=None, min_ps=None, return_stats=True)
        return tp_nucleus_tokens(logits_local, local, t, p, m, self.sampler.seed, self.sampler.calls + 1,
                                 self.sampler.offset_dev)

    def _greedy_merge(self, logits_local, t, k, p, m):
        """Tokens of an all-greedy batch from this rank's vocab shard (temperature 0 rows: the sampler
        kernel returns the lowest-index argmax of the shard)."""
        local = self.sampler(logits_local, temperatures=t, top_ks=k, top_ps=p, min_ps=m)
        return tp_greedy_tokens(logits_local, local)

    def _sample_tokens(self, logits, sequences, mode: str = "gather") -> list[int]:
        if self.sampler is None:
            return []
        if len(sequences) > self._samp_i64.shape[0]:
            dev = logits.device
            t = torch.tensor([s.temperature for s in sequences], dtype=torch.float32, device=dev)
            k = torch.tensor([s.top_k for s in sequences], dtype=torch.int64, device=dev)
            p = torch.tensor([s.top_p for s in sequences], dtype=torch.float32, device=dev)
            m = torch.tensor([s.min_p for s in sequences], dtype=torch.float32, device=dev)
        else:
            t, k, p, m = self._ship_sampling(sequences)
        if mode == "greedy":
            return self._greedy_merge(logits, t, k, p, m).tolist()
        if mode == "cand":
            return self._cand_tokens(logits, t, k, p, m).tolist()
        if mode == "mass":
            return self._mass_tokens(logits, t, k, p, m).tolist()
        if mode == "nuc":
            return self._nuc_tokens(logits, t, k, p, m).tolist()
        return self.sampler(logits, temperatures=t, top_ks=k, top_ps=p, min_ps=m).tolist()

    def _update_metrics(self, sequences, prefill: bool) -> None:
        if prefill:
            self.total_prefill_tokens += sum(s.prefill_stop - s.num_cached_tokens for s in sequences)
        else:
            self.total_decode_tokens += len(sequences)
        self.total_tokens_generated += len(sequences)

    # ------------------------------------------------------------------ hipGraph decode
    def _graph_bucket(self, sequences) -> int | None:
        """Smallest captured bucket that holds the batch, in the graph family the batch decodes with."""
        if self.enforce_eager:
            return None
        family = self._graph_family(sequences, capture=True)[1]
        return next((bs for bs in self.graph_bs if bs >= len(sequences) and bs in family), None)

    def _graph_family(self, sequences, capture: bool = False):
        """-> (mode, graphs of the family this batch decodes with).  The candidate family and (under TP) the
        gathered-logits family are captured on first use."""
        mode = self._tp_mode(sequences, decode=True)
        if mode == "greedy" and self.greedy_graphs:
            return "greedy", self.greedy_graphs
        if mode == "cand" and self._cand_capture != "no":
            if self._cand_capture == "lazy" and capture:
                self._drop_speculation()
                torch.cuda.synchronize(self.device)
                self._capture_cand()
            if self.cand_graphs:
                return "cand", self.cand_graphs
        if mode == "mass" and self.mass_graphs:
            return "mass", self.mass_graphs
        if mode == "nuc":
            if self._nuc_capture == "lazy" and capture:
                self._drop_speculation()
                torch.cuda.synchronize(self.device)
                self._capture_nuc()
                if not self._nuc_family_usable():    # the capture just failed: these batches replay the gather graphs
                    return self._graph_family(sequences, capture)
            # (no graphs at all: the step runs eagerly in 'nuc' mode -- _graph_bucket finds no bucket in an empty family)
            return "nuc", self.nuc_graphs
        if capture and self._general_capture == "lazy":
            self._drop_speculation()
            torch.cuda.synchronize(self.device)
            self._capture_general()
        return "gather", self.graphs

    def capture_device_graphs(self, max_batch_size: int) -> None:
        """Capture one hipGraph per decode batch-size bucket (largest first, shared pool)."""
        if self.enforce_eager:
            return
        general_ok = True
        if self.tp_size > 1:
            import torch.distributed as dist

            if dist.get_backend() != "nccl":
                # only RCCL collectives can be captured (and a failed capture cannot be recovered from): the
                # gathered-logits graphs are skipped; the all-greedy graphs contain no backend collective when
                # the peer-to-peer all-reduce is up, and are captured
                general_ok = False
                hf_ = self.config.hf_config
                if not decode_step_is_native(self.dtype, hf_.hidden_size, hf_.vocab_size):
                    logger.warning("tensor-parallel backend %s is not capturable: decoding eagerly", dist.get_backend())
                    return
        self.graph_bs = graph_batch_sizes(min(max_batch_size, self.config.max_num_seqs))
        only = os.environ.get("MV_GRAPH_BATCH_SIZES")
        if only:   # diagnostic / bench knob: capture just these buckets (bench.py's secondary TP engines replay one)
            keep = {int(v) for v in only.split(",") if v.strip()}
            self.graph_bs = [b for b in self.graph_bs if b in keep] or self.graph_bs
        max_bs = self.graph_bs[-1]
        hidden = self.config.hf_config.hidden_size
        vocab = self.config.hf_config.vocab_size
        dev = self.device
        gv = self.graph_vars = {
            "input_ids": self._dev_ids[:max_bs], "positions": self._dev_pos[:max_bs],
            "slot_mapping": self._dev_slots[:max_bs], "context_lens": self._dev_ctx[:max_bs],
            "block_tables": self._dev_tables[:max_bs],
            "logits": torch.zeros((max_bs, vocab), dtype=self.dtype, device=dev),
            "logits_local": (torch.zeros((max_bs, vocab // self.tp_size), dtype=self.dtype, device=dev)
                             if self.tp_size > 1 else None),
            "tokens": torch.zeros(max_bs, dtype=torch.int64, device=dev),
            "rng_offset": self._dev_rng[:1],
        }
        # the split-KV workspace must exist before capture (no allocation inside a graph)
        hf = self.config.hf_config
        heads = hf.num_attention_heads // get_tensor_parallel_world_size()
        # ... sized for every batch an EAGER step can carry too (max_num_seqs may exceed the largest graph bucket)
        hip_ops.workspace(max(hip_ops.paged_decode_workspace_bytes(bs, heads, self.num_kv_heads, self.head_dim,
                                                                   self.block_size, self.max_blocks_per_seq)
                              for bs in list(self.graph_bs) + [self.config.max_num_seqs]), dev)
        self._general_capture = "no" if not general_ok else "done"
        try:
            if self.tp_size > 1:   # the family without a backend collective inside (peer-to-peer kernels only)
                for bs in reversed(self.graph_bs):
                    self._capture_one(bs, greedy=True)
                if self.greedy_graphs and decode_step_is_native(self.dtype, hf.hidden_size, hf.vocab_size):
                    # same property: only this library's kernels inside, so nothing argues for laziness -- captured
                    # here, out of the same pool, before any sequence is live (a capture in mid-serving would stall
                    # every rank for seconds and take its memory from whatever the KV pool left over)
                    self._capture_cand()
                    self._capture_mass()
                if general_ok and self.greedy_graphs and decode_step_is_native(self.dtype, hf.hidden_size, hf.vocab_size):
                    # the gathered-logits family needs RCCL nodes in its graphs: captured on the first batch with
                    # a sampled row (_graph_bucket), so that a greedy-only workload never captures a collective
                    self._general_capture = "lazy"
                    self._nuc_capture = "lazy"      # same reason: three small backend all-reduces inside
            if self._general_capture == "done":
                self._capture_general()
            logger.info("Captured %d decode graphs (batch sizes %s..%d)", len(self.graphs) + len(self.greedy_graphs),
                        self.graph_bs[:4], self.graph_bs[-1])
        except Exception as e:  # fall back to eager, loudly
            logger.warning("Failed to capture device graphs, decoding eagerly: %s", e)
            self.graphs.clear()
            self.greedy_graphs.clear()
            self.cand_graphs.clear()
            self.mass_graphs.clear()
            self.nuc_graphs.clear()
            self.graph_pool = None
            self._general_capture = self._cand_capture = self._mass_capture = self._nuc_capture = "no"
            self._recover_from_failed_capture()

    def _capture_general(self) -> None:
        """The gathered-logits graphs (every sampling mode).  Under TP a failure here only sends batches with
        sampled rows to eager decoding: the all-greedy graphs stay."""
        if self._general_capture == "lazy":
            self._neutralise_graph_inputs()
        try:
            for bs in reversed(self.graph_bs):
                self._capture_one(bs, greedy=False)
            self._general_capture = "done"
        except Exception as e:
            if not self.greedy_graphs:
                raise
            logger.warning("Failed to capture the gathered-logits graphs (sampled batches decode eagerly): %s", e)
            self.graphs.clear()
            self._general_capture = "no"
            self._recover_from_failed_capture()

    def _neutralise_graph_inputs(self) -> None:
        """Before a capture that happens while sequences are LIVE (the lazily captured families): the warm-up and
        capture runs execute the model on whatever the graph inputs hold, and a stale slot (0 after start-up) would
        overwrite a live sequence's K/V row.  Slot -1 rows store nothing and context 0 rows read nothing — the
        replay-padding convention (reference inference_executor.py:758-764); the next real replay rewrites the
        whole input pack from the host."""
        gv = self.graph_vars
        gv["slot_mapping"].fill_(-1)
        gv["context_lens"].zero_()
        gv["input_ids"].zero_()
        gv["positions"].zero_()
        torch.cuda.synchronize(self.device)

    def _capture_cand(self) -> None:
        """The per-rank-candidates graphs (batches of greedy / small-top-k rows under TP)."""
        self._neutralise_graph_inputs()
        try:
            for bs in reversed(self.graph_bs):
                self._capture_one(bs, mode="cand")
            self._cand_capture = "done"
        except Exception as e:
            logger.warning("Failed to capture the candidate-sampling graphs (such batches gather their logits): %s", e)
            self.cand_graphs.clear()
            self._cand_capture = "no"
            self._recover_from_failed_capture()

    def _capture_mass(self) -> None:
        """The two-stage-draw graphs (batches of greedy rows / rows sampled from the whole vocabulary under TP)."""
        self._neutralise_graph_inputs()
        try:
            for bs in reversed(self.graph_bs):
                self._capture_one(bs, mode="mass")
            self._mass_capture = "done"
        except Exception as e:
            logger.warning("Failed to capture the two-stage-draw graphs (such batches gather their logits): %s", e)
            self.mass_graphs.clear()
            self._mass_capture = "no"
            self._recover_from_failed_capture()

    def _capture_nuc(self) -> None:
        """The cross-rank-floor graphs (batches with nucleus / min-p rows and top_k off under TP), on first use."""
        self._neutralise_graph_inputs()
        try:
            for bs in reversed(self.graph_bs):
                self._capture_one(bs, mode="nuc")
            self._nuc_capture = "done"
        except Exception as e:
            logger.warning("Failed to capture the nucleus-floor graphs (such batches decode eagerly): %s", e)
            self.nuc_graphs.clear()
            self._nuc_capture = "no"
            self._recover_from_failed_capture()

    def _recover_from_failed_capture(self) -> None:
        reset_context()
        # a failed capture leaves torch's side capture stream current (torch.cuda.graph.__exit__
        # raises in capture_end before restoring it): go back to the default stream
        torch.cuda.set_stream(torch.cuda.default_stream(self.device))
        try:
            torch.cuda.synchronize(self.device)
        except Exception:  # the invalidated capture reports once more; the device itself is fine
            pass

    def _graph_forward(self, bs: int, greedy: bool = False, mode: str | None = None):
        mode = mode or ("greedy" if greedy else "gather")
        greedy = mode == "greedy"
        gv = self.graph_vars
        set_context(False, 0, 0, None, None, gv["slot_mapping"][:bs], gv["context_lens"][:bs],
                    gv["block_tables"][:bs])
        hidden = self.model(gv["input_ids"][:bs], gv["positions"][:bs])
        d = self._dev_samp_f32
        if greedy:  # TP > 1, every row at temperature 0: merge per-rank argmaxes, no logit gather
            out = gv["logits_local"][:bs]
            out.copy_(self.model.compute_logits(hidden, gather=False))
            local = hip_ops.sample(out, d[0, :bs], self._dev_samp_i64[:bs], d[1, :bs], d[2, :bs],
                                   seed=self.sampler.seed, offset=0, offset_dev=gv["rng_offset"])
            toks = tp_greedy_tokens(out, local)
        elif mode == "cand":  # TP > 1, rows greedy or with a small top_k: per-rank top-64 candidates, no logit gather
            out = gv["logits_local"][:bs]
            out.copy_(self.model.compute_logits(hidden, gather=False))
            toks = tp_candidate_tokens(out, lambda cl: hip_ops.sample(
                cl, d[0, :bs], self._dev_samp_i64[:bs], d[1, :bs], d[2, :bs], seed=self.sampler.seed, offset=0,
                offset_dev=gv["rng_offset"]), temperatures=d[0, :bs])
        elif mode == "mass":  # TP > 1, rows greedy or sampled from the whole vocabulary: two-stage draw, no logit gather
            out = gv["logits_local"][:bs]
            out.copy_(self.model.compute_logits(hidden, gather=False))
            toks = tp_mass_tokens(out, lambda ll: hip_ops.sample(
                ll, d[0, :bs], self._dev_samp_i64[:bs], d[1, :bs], d[2, :bs], seed=self.sampler.seed, offset=0,
                offset_dev=gv["rng_offset"], return_stats=True), d[0, :bs], self.sampler.seed, 0, gv["rng_offset"])
        elif mode == "nuc":   # TP > 1, top_k off, nucleus / min-p rows: cross-rank floor, masked two-stage draw
            out = gv["logits_local"][:bs]
            out.copy_(self.model.compute_logits(hidden, gather=False))
            toks = tp_nucleus_tokens(out, lambda ll: hip_ops.sample(
                ll, d[0, :bs], None, None, None, seed=self.sampler.seed, offset=0, offset_dev=gv["rng_offset"],
                return_stats=True), d[0, :bs], d[1, :bs], d[2, :bs], self.sampler.seed, 0, gv["rng_offset"])
        else:
            out = gv["logits"][:bs]
            out.copy_(self.model.compute_logits(hidden))
            toks = hip_ops.sample(out, d[0, :bs], self._dev_samp_i64[:bs], d[1, :bs], d[2, :bs],
                                  seed=self.sampler.seed, offset=0, offset_dev=gv["rng_offset"])
        gv["tokens"][:bs].copy_(toks)
        # last node: advance the device-side decode inputs so that the NEXT step can be replayed as is
        hip_ops.decode_advance(gv["input_ids"][:bs], gv["tokens"][:bs], gv["positions"][:bs],
                               gv["slot_mapping"][:bs], gv["context_lens"][:bs], gv["rng_offset"])

    @torch.inference_mode()
    def _capture_one(self, bs: int, greedy: bool = False, mode: str | None = None) -> None:
        mode = mode or ("greedy" if greedy else "gather")
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):  # warm-up on a side stream (RCCL / hipBLASLt lazy init)
            self._graph_forward(bs, mode=mode)
        torch.cuda.current_stream(self.device).wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=self.graph_pool):
            self._graph_forward(bs, mode=mode)
        if self.graph_pool is None:
            self.graph_pool = graph.pool()
        {"greedy": self.greedy_graphs, "cand": self.cand_graphs, "mass": self.mass_graphs, "nuc": self.nuc_graphs,
         "gather": self.graphs}[mode][bs] = graph
        torch.cuda.synchronize(self.device)
        reset_context()

    def _replay_graph(self, bs: int, inp: BatchInputs, sequences):
        """One pinned->device copy of the decode pack (+ block tables / sampling parameters only when they
        changed), then the captured graph: model, LM head, sampler, input advance.

        Speculative next step: the graph ends by advancing its own inputs on the device (ids <- sampled
        tokens, positions/context/slots + 1), so when the engine says nothing is waiting and no sequence
        of the batch is about to finish or to enter a new block, the NEXT step's graph is enqueued right
        behind this one, before the host has even read this step's tokens.  The host's scheduling,
        input building and bookkeeping then overlap GPU work instead of leaving it idle between steps.
        On the next call the speculation is validated (same sequences, same order, every length + 1,
        block tables untouched); a miss (EOS, preemption, new request) just overwrites the device state
        from the host as usual — the speculative step only re-wrote K/V rows that the real step rewrites
        with identical values or that belong to finished sequences."""
        b = inp.num_seqs
        gv = self.graph_vars
        mode, family = self._graph_family(sequences)
        greedy = mode != "gather"          # (this rank's vocab shard is what the step leaves in logits_local)
        graph = family[bs]
        spec = self._spec
        hit = False
        if spec is not None:
            batch, sbs, lens, slot, sgraph = spec
            hit = (sbs == bs and sgraph is graph and len(batch) == b and not inp.tables_dirty
                   and not inp.batch_changed
                   and all(x is y for x, y in zip(batch, sequences))
                   and all(s.num_tokens == n for s, n in zip(sequences, lens)))
            self.spec_hits += hit
            self.spec_misses += not hit
            if hit:
                self._spec = None
            else:
                self._drop_speculation()
        if hit:
            slot_now = slot  # this step is already executing (or done): its tokens land in that slot
        else:
            self.sampler.calls += 1
            self.builder.p_rng[0] = self.sampler.calls      # Philox offset of this step, inside the pack
            self._pack_dev.copy_(self.builder.decode_pack, non_blocking=True)
            w = inp.block_tables.shape[1]
            if w and inp.tables_dirty:
                self._dev_tables[:b, :w].copy_(inp.block_tables, non_blocking=True)
            self._ship_sampling(sequences, b_pad=bs, unchanged=not inp.batch_changed)
            graph.replay()
            slot_now = self._stash_tokens(b)
        if self.allow_speculation and self.speculate_next and self._can_speculate(sequences):
            self.sampler.calls += 1                          # the graph advanced the device-side offset itself
            graph.replay()
            self._spec = (list(sequences), bs, [s.num_tokens + 1 for s in sequences], self._stash_tokens(b), graph)
        self._tok_events[slot_now].synchronize()
        if self._ar_slots[slot_now] is not None:     # the all-reduce status block that travelled with these tokens
            get_custom_allreduce().verify(self._ar_slots[slot_now])
        return (gv["logits_local"] if greedy else gv["logits"])[:b], self._tok_host[slot_now][:b].tolist()

    def _drop_speculation(self) -> None:
        """Forget the speculative step in flight (its tokens are never read) and give its Philox step count
        back, so that the random stream of the real steps does not depend on what was speculated."""
        if self._spec is not None:
            self._spec = None
            self.sampler.calls -= 1

    def _stash_tokens(self, b: int) -> int:
        """Async D2H of the step that was just enqueued into a pinned ring slot + an event to wait on."""
        slot = self._tok_slot
        self._tok_slot = (slot + 1) % len(self._tok_host)
        self._tok_host[slot][:b].copy_(self.graph_vars["tokens"][:b], non_blocking=True)
        ar = get_custom_allreduce() if self.tp_size > 1 else None
        self._ar_slots[slot] = ar.post_status() if ar is not None and ar.ready else None
        self._tok_events[slot].record()
        return slot

    def _can_speculate(self, sequences) -> bool:
        bs = self.block_size
        for s in sequences:
            # after this step the sequence has num_tokens + 1 tokens; the step after that needs a new block
            # iff num_tokens % bs == 0, and there is no such step if this token exhausts max_tokens
            if (s.num_tokens % bs == 0 or s.num_completion_tokens + 1 >= s.max_tokens
                    or s.num_tokens + 1 >= self.config.max_model_len):
                return False
        return True

    def cleanup(self) -> None:
        self.graphs.clear()
        self.greedy_graphs.clear()
        self.cand_graphs.clear()
        self.mass_graphs.clear()
        self.nuc_graphs.clear()
        self.graph_vars.clear()
        self.graph_pool = None
        self.kv_cache = None
        self.model = None
        self.sampler = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
