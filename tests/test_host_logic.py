"""Host-side integer logic (block manager, scheduler, sequence, input builder) replayed against
traces recorded from the reference (tests/golden/host_traces.json, tools/gen_goldens.py):
block tables, slot mappings, cu_seqlens, positions, context lens, batch composition, preemption
order, prefix-cache counters and xxh64 digests must match BIT FOR BIT, dtypes included."""
import json
import os
import pickle
from types import SimpleNamespace

import pytest
import torch

from minivllm_amd.engine.block_manager import BlockManager
from minivllm_amd.engine.input_builder import InputBuilder
from minivllm_amd.engine.scheduler import Scheduler
from minivllm_amd.engine.sequence import Sequence, SequenceStatus
from minivllm_amd.sampling_params import SamplingParams


@pytest.fixture(scope="module")
def traces(golden_dir):
    with open(os.path.join(golden_dir, "host_traces.json")) as f:
        return json.load(f)


def fake_next_token(tokens, vocab):
    return (sum(tokens) * 31 + len(tokens) * 7 + 3) % vocab


def test_xxh64_digests(traces):
    for case in traces["hashes"]:
        assert BlockManager.compute_hash(case["tokens"], case["prefix"]) == case["hash"]
    # SURVEY §8 known answers
    assert BlockManager.compute_hash([1, 2, 3]) == 13048208712247349756
    assert BlockManager.compute_hash([1, 2, 3], prefix=5) == 17518828517520909647
    assert BlockManager.compute_hash(list(range(64))) == 4138455281752509893
    with pytest.raises(ValueError):
        BlockManager.compute_hash([])


def _check_tensor(name, got, want, want_dtype, step):
    if want is None:
        assert got is None, f"step {step}: {name} should be None"
        return
    assert got is not None, f"step {step}: {name} missing"
    assert str(got.dtype).replace("torch.", "") == want_dtype, f"step {step}: {name} dtype {got.dtype}"
    assert got.tolist() == want, f"step {step}: {name} differs"


@pytest.mark.parametrize("idx", range(6))
def test_replay_reference_trace(traces, idx):
    rec = traces["scenarios"][idx]
    sc = rec["scenario"]
    cfg = SimpleNamespace(max_num_seqs=sc["max_num_seqs"], max_num_batched_tokens=sc["max_num_batched_tokens"],
                          eos=sc.get("eos", -1), num_kvcache_blocks=sc["num_blocks"],
                          kvcache_block_size=sc["block_size"])
    sched = Scheduler(cfg)
    builder = InputBuilder(sc["block_size"], sc["max_num_seqs"], sc["max_model_len"], sc["max_num_batched_tokens"])
    seqs = [Sequence(r["tokens"], SamplingParams(temperature=r.get("temperature", 1.0), max_tokens=r["max_tokens"],
                                                 ignore_eos=r.get("ignore_eos", False)), block_size=sc["block_size"])
            for r in sc["requests"]]
    index_of = {id(s): i for i, s in enumerate(seqs)}
    add_at = sc.get("add_at", [0] * len(seqs))
    pending = list(range(len(seqs)))
    step_no, k = 0, 0
    while True:
        for i in list(pending):
            if add_at[i] <= step_no:
                sched.add(seqs[i])
                pending.remove(i)
        if sched.is_finished() and not pending:
            break
        if sched.is_finished():
            step_no += 1
            continue
        want = rec["steps"][k]
        batch, is_prefill = sched.schedule()
        assert is_prefill == want["is_prefill"], f"step {k}"
        assert [index_of[id(s)] for s in batch] == want["seq_index"], f"step {k}: batch composition"
        assert [list(s.block_table) for s in batch] == want["block_table_per_seq"], f"step {k}: block ids"
        assert [s.num_cached_tokens for s in batch] == want["num_cached_tokens"], f"step {k}"
        inp = builder.prefill(batch) if is_prefill else builder.decode(batch)
        assert (inp.max_seqlen_q, inp.max_seqlen_k) == (want["max_seqlen_q"], want["max_seqlen_k"])
        _check_tensor("input_ids", inp.input_ids, want["input_ids"], want["input_ids_dtype"], k)
        _check_tensor("positions", inp.positions, want["positions"], want["positions_dtype"], k)
        for name in ("cum_seqlens_q", "cum_seqlens_k", "slot_mapping", "context_lens", "block_tables"):
            _check_tensor(name, getattr(inp, name), want[name], want[name + "_dtype"], k)
        toks = [fake_next_token(s.token_ids, sc["vocab"]) for s in batch]
        assert toks == want["next_tokens"]
        sched.postprocess(batch, toks)
        assert [s.is_finished for s in batch] == want["finished"], f"step {k}"
        assert sched.block_manager.get_num_free_blocks() == want["free_blocks_after"], f"step {k}"
        assert [index_of[id(s)] for s in sched.waiting] == want["waiting_after"], f"step {k}"
        assert [index_of[id(s)] for s in sched.running] == want["running_after"], f"step {k}"
        k += 1
        step_no += 1
    assert k == len(rec["steps"])
    assert [dict(tokens=s.token_ids, num_completion=s.num_completion_tokens) for s in seqs] == rec["final"]
    assert sched.block_manager.stats == rec["stats"]


def test_survey_known_answer_trace():
    """two prompts (130 and 3 tokens), fresh BlockManager(16, 64)."""
    bm = BlockManager(16, 64)
    a, b = Sequence(list(range(1, 131))), Sequence(list(range(500, 503)))
    bm.allocate(a)
    bm.allocate(b)
    ib = InputBuilder(64, 8, 1024, 4096)
    p = ib.prefill([a, b])
    assert p.cum_seqlens_q.tolist() == p.cum_seqlens_k.tolist() == [0, 130, 133]
    assert p.slot_mapping.tolist() == list(range(130)) + [192, 193, 194]
    assert p.block_tables.tolist() == [[0, 1, 2], [3, -1, -1]] and p.context_lens is None
    for s in (a, b):
        s.append_token(7)
        bm.may_append(s)
    d = ib.decode([a, b])
    assert d.slot_mapping.tolist() == [130, 195] and d.context_lens.tolist() == [131, 4]
    assert d.positions.tolist() == [130, 3] and d.block_tables.tolist() == [[0, 1, 2], [3, -1, -1]]
    assert (d.max_seqlen_q, d.max_seqlen_k, d.cum_seqlens_q) == (0, 0, None)


def test_prefix_cache_hit_shares_block():
    bm = BlockManager(8, 64)
    a = Sequence(list(range(1, 65)) + [200, 201, 202])
    b = Sequence(list(range(1, 65)) + [300, 301, 302, 303, 304])
    bm.allocate(a)
    bm.allocate(b)
    assert (a.block_table, b.block_table, b.num_cached_tokens) == ([0, 1], [0, 2], 64)
    assert bm.blocks[0].ref_count == 2 and bm.stats["cache_hits"] == 1
    bm.deallocate(a)
    assert bm.blocks[0].ref_count == 1 and 0 in bm.used_block_ids
    bm.deallocate(b)
    c = Sequence(list(range(1, 65)) + [9])      # block 0 is free but still cached: revived, not reset
    bm.allocate(c)
    assert c.block_table[0] == 0 and c.num_cached_tokens == 64


def test_fully_cached_prompt_keeps_a_query_block():
    """Build semantics where the reference is broken: a prompt whose every block
    hits the prefix cache recomputes its last block so that a last-token logit exists."""
    bm = BlockManager(8, 64)
    toks = list(range(1, 129))
    a, b = Sequence(toks), Sequence(toks)
    bm.allocate(a)
    bm.allocate(b)
    assert b.block_table == a.block_table and b.num_cached_tokens == 64
    p = InputBuilder(64, 4, 512, 1024).prefill([b])
    assert p.cum_seqlens_q.tolist() == [0, 64] and p.cum_seqlens_k.tolist() == [0, 128]
    assert p.positions.tolist() == list(range(64, 128))
    assert p.slot_mapping.tolist() == list(range(b.block_table[1] * 64, b.block_table[1] * 64 + 64))


def test_block_pressure_preempts_instead_of_raising():
    """the reference raises 'No free blocks available' when a sequence needs its
    next block (len % bs == 1) and none is free; the build preempts the newest sequence instead."""
    cfg = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=512, eos=-1, num_kvcache_blocks=3,
                          kvcache_block_size=64)
    sched = Scheduler(cfg)
    a = Sequence(list(range(1, 65)), SamplingParams(max_tokens=40, ignore_eos=True))
    b = Sequence(list(range(100, 164)), SamplingParams(max_tokens=40, ignore_eos=True))
    sched.add(a)
    sched.add(b)
    steps = 0
    while not sched.is_finished():
        batch, _ = sched.schedule()
        sched.postprocess(batch, [1] * len(batch))
        steps += 1
        assert steps < 400
    assert a.num_completion_tokens == 40 and b.num_completion_tokens == 40


def test_generation_stops_at_max_model_len():
    """Beyond the reference (it runs off the end of its block-table staging buffer): a sequence that has filled
    max_model_len finishes there, whatever max_tokens says; shorter ones are unaffected."""
    cfg = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=512, eos=-1, num_kvcache_blocks=8,
                          kvcache_block_size=64, max_model_len=128)
    sched = Scheduler(cfg)
    long_, short = (Sequence([5] * 120, SamplingParams(max_tokens=40, ignore_eos=True)),
                    Sequence([7] * 10, SamplingParams(max_tokens=40, ignore_eos=True)))
    sched.add(long_)
    sched.add(short)
    while not sched.is_finished():
        batch, _ = sched.schedule()
        sched.postprocess(batch, [1] * len(batch))
    assert (len(long_), long_.num_completion_tokens) == (128, 8) and short.num_completion_tokens == 40
    assert long_.status == SequenceStatus.FINISHED and len(long_.block_table) == 0


def test_sequence_alone_in_a_full_pool_finishes_instead_of_preempting_itself_forever():
    cfg = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=512, eos=-1, num_kvcache_blocks=2,
                          kvcache_block_size=64, max_model_len=4096)
    sched = Scheduler(cfg)
    seq = Sequence([5] * 100, SamplingParams(max_tokens=100, ignore_eos=True))
    sched.add(seq)
    steps = 0
    while not sched.is_finished():
        batch, _ = sched.schedule()
        sched.postprocess(batch, [1] * len(batch))
        steps += 1
        assert steps < 100
    assert len(seq) == 128 and sched.num_preemptions == 0


def test_sequence_block_math_and_pickle():
    s = Sequence(list(range(130)), SamplingParams(temperature=0.5, top_k=7, max_tokens=9), block_size=64)
    assert (s.num_blocks, s.last_block_num_tokens, s.num_cached_blocks) == (3, 2, 0)
    assert s.block(2) == [128, 129] and len(s.block(0)) == 64
    with pytest.raises(IndexError):
        s.block(3)
    with pytest.raises(ValueError):
        Sequence([])
    with pytest.raises(ValueError):
        s.append_token(-1)
    r = pickle.loads(pickle.dumps(s))            # prompt phase: whole prompt travels
    assert r.token_ids == s.token_ids and r.status == SequenceStatus.WAITING and r.top_k == 7
    s.append_token(5)
    s.block_table = [4, 2, 9]
    r = pickle.loads(pickle.dumps(s))            # decode phase: only the last token travels
    assert r.last_token == 5 and len(r) == 131 and r.token_ids[-1] == 5 and r.token_ids[0] == 0
    assert r.block_table == [4, 2, 9] and r.status == SequenceStatus.RUNNING
    # what a tensor-parallel worker rebuilds must give the input builder the same token ranges as rank 0's object:
    # a chunk of a long prompt (prefill_end), and a decode row of a mixed batch (num_cached_tokens = len - 1)
    from minivllm_amd.engine.input_builder import InputBuilder
    c = Sequence(list(range(1, 301)), SamplingParams(max_tokens=3), block_size=64)
    c.block_table, c.num_cached_tokens, c.prefill_end = [7, 8, 9, 10, 11], 128, 256
    s.num_cached_tokens = len(s) - 1
    mine = InputBuilder(64, 4, 512, 512).prefill([s, c])
    theirs = InputBuilder(64, 4, 512, 512).prefill(pickle.loads(pickle.dumps([s, c])))
    for name in ("input_ids", "positions", "cum_seqlens_q", "cum_seqlens_k", "slot_mapping", "block_tables",
                 "context_lens", "rest_cum_seqlens_q", "rest_cum_seqlens_k"):
        assert torch.equal(getattr(mine, name), getattr(theirs, name)), name
    assert (mine.decode_rows, mine.max_seqlen_q, mine.rest_max_seqlen_k) == (theirs.decode_rows, theirs.max_seqlen_q, theirs.rest_max_seqlen_k) == (1, 128, 256)
    s.status = SequenceStatus.FINISHED
    with pytest.raises(RuntimeError):
        s.append_token(1)


def test_sampling_params_validation():
    SamplingParams(temperature=0.0)
    for bad in (dict(temperature=-0.1), dict(top_p=0.0), dict(top_p=1.5), dict(top_k=0), dict(top_k=-2),
                dict(min_p=1.5), dict(max_tokens=0)):
        with pytest.raises(ValueError):
            SamplingParams(**bad)


def test_block_manager_errors_and_may_append_boundaries():
    with pytest.raises(ValueError):
        BlockManager(0, 64)
    with pytest.raises(ValueError):
        BlockManager(4, 0)
    bm = BlockManager(2, 4)
    s = Sequence([1, 2, 3], block_size=4)
    bm.allocate(s)
    with pytest.raises(ValueError):
        bm.allocate(s)                              # already has blocks
    assert bm.blocks[s.block_table[0]].hash == -1   # partial block: no digest
    s.append_token(4)
    bm.may_append(s)                                # block filled: digest published
    blk = bm.blocks[s.block_table[0]]
    assert blk.hash == BlockManager.compute_hash([1, 2, 3, 4]) and bm.hash_to_block_id[blk.hash] == blk.block_id
    s.append_token(5)
    bm.may_append(s)                                # first token of a new block: block taken
    assert len(s.block_table) == 2 and bm.get_num_free_blocks() == 0
    big = Sequence(list(range(20)), block_size=4)
    assert not bm.can_allocate(big)
    with pytest.raises(ValueError):
        bm.allocate(big)
    bm.deallocate(s)
    assert bm.get_num_free_blocks() == 2 and s.block_table == [] and s.num_cached_tokens == 0
    # digest is dropped when the block is handed out for new content
    t = Sequence([9, 9, 9, 9, 9], block_size=4)
    bm.allocate(t)
    assert blk.hash != BlockManager.compute_hash([1, 2, 3, 4]) or blk.block_id not in t.block_table


# ---------------------------------------------------------------- chunked prefill (beyond the reference, §8 f4)
def test_chunked_prefill_schedule_and_inputs():
    """A 300-token prompt with a 128-token step budget: three prefill steps of 128 / 128 / 44 tokens, each naming
    the whole prefix as keys (cu_k) and only its own tokens as queries / slots; nothing else is scheduled in
    between; only the last chunk's token becomes a token.  Short prompts schedule exactly as without the flag."""
    cfg = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=128, eos=-1, num_kvcache_blocks=16,
                          kvcache_block_size=64, max_model_len=512, enable_chunked_prefill=True)
    sched = Scheduler(cfg)
    builder = InputBuilder(64, 4, 512, 128)
    long_seq = Sequence(list(range(1, 301)), SamplingParams(max_tokens=3, ignore_eos=True), block_size=64)
    short = Sequence([7, 8, 9], SamplingParams(max_tokens=2, ignore_eos=True), block_size=64)
    sched.add(long_seq)
    sched.add(short)
    spans = []
    for step in range(3):
        batch, is_prefill = sched.schedule()
        assert is_prefill and batch == [long_seq]
        inp = builder.prefill(batch)
        lo, hi = long_seq.num_cached_tokens, long_seq.prefill_stop
        spans.append((lo, hi))
        assert inp.input_ids.tolist() == list(range(1 + lo, 1 + hi))
        assert inp.positions.tolist() == list(range(lo, hi))
        assert inp.cum_seqlens_q.tolist() == [0, hi - lo] and inp.cum_seqlens_k.tolist() == [0, hi]
        assert (inp.max_seqlen_q, inp.max_seqlen_k) == (hi - lo, hi)
        want_slots = [long_seq.block_table[p // 64] * 64 + p % 64 for p in range(lo, hi)]
        assert inp.slot_mapping.tolist() == want_slots
        assert inp.block_tables.tolist() == [long_seq.block_table]
        sched.postprocess(batch, [11])
    assert spans == [(0, 128), (128, 256), (256, 300)]
    assert long_seq.token_ids[-1] == 11 and len(long_seq) == 301 and long_seq.prefill_end is None
    assert sched.prefilling is None and list(sched.running) == [long_seq]
    batch, is_prefill = sched.schedule()              # now the short prompt, as usual
    assert is_prefill and batch == [short]
    sched.postprocess(batch, [5])
    batch, is_prefill = sched.schedule()
    assert not is_prefill and batch == [long_seq, short]
    # without the flag the same prompt is what the reference makes of it: never schedulable
    cfg.enable_chunked_prefill = False
    s2 = Scheduler(cfg)
    s2.add(Sequence(list(range(1, 301)), block_size=64))
    with pytest.raises(RuntimeError, match="No sequences scheduled"):
        s2.schedule()


def test_mixed_batches_schedule_and_inputs():
    """enable_mixed_batches (beyond the reference, §8 f4): while a 300-token prompt is prefilled in chunks, the two
    running sequences keep decoding — every prefill step carries one token of each of them, in front, as rows whose
    all-but-last tokens are cached; the builder names them decode rows (context lengths + their table rows) and
    re-bases the prompt's cumulative lengths; middle chunks yield no token."""
    cfg = SimpleNamespace(max_num_seqs=8, max_num_batched_tokens=130, eos=-1, num_kvcache_blocks=16,
                          kvcache_block_size=64, max_model_len=512, enable_chunked_prefill=True,
                          enable_mixed_batches=True)
    sched = Scheduler(cfg)
    builder = InputBuilder(64, 8, 512, 130)
    sp = SamplingParams(max_tokens=20, ignore_eos=True)
    a = Sequence(list(range(10, 73)), sp, block_size=64)      # 63 tokens: crosses a block edge while decoding
    b = Sequence([5, 6, 7], sp, block_size=64)
    sched.add(a)
    sched.add(b)
    batch, is_prefill = sched.schedule()                      # nothing is running yet: the reference's prefill
    assert is_prefill and batch == [a, b]
    assert builder.prefill(batch).decode_rows == 0
    sched.postprocess(batch, [100, 200])
    long_seq = Sequence(list(range(1000, 1300)), SamplingParams(max_tokens=3, ignore_eos=True), block_size=64)
    sched.add(long_seq)
    spans, next_tok = [], 300
    while sched.prefilling is not None or long_seq.num_completion_tokens == 0:
        la, lb = len(a), len(b)
        batch, is_prefill = sched.schedule()
        assert is_prefill and batch == [a, b, long_seq]
        inp = builder.prefill(batch)
        lo, hi = long_seq.num_cached_tokens, long_seq.prefill_stop
        spans.append((lo, hi))
        assert inp.decode_rows == 2 and inp.context_lens.tolist() == [la, lb]
        assert inp.input_ids.tolist() == [a.last_token, b.last_token] + list(range(1000 + lo, 1000 + hi))
        assert inp.positions.tolist() == [la - 1, lb - 1] + list(range(lo, hi))
        assert inp.cum_seqlens_q.tolist() == [0, 1, 2, 2 + hi - lo] and inp.cum_seqlens_k.tolist() == [0, la, la + lb, la + lb + hi]
        assert inp.rest_cum_seqlens_q.tolist() == [0, hi - lo] and inp.rest_cum_seqlens_k.tolist() == [0, hi]
        assert (inp.rest_max_seqlen_q, inp.rest_max_seqlen_k) == (hi - lo, hi)
        slot = lambda s, p: s.block_table[p // 64] * 64 + p % 64   # noqa: E731
        assert inp.slot_mapping.tolist() == [slot(a, la - 1), slot(b, lb - 1)] + [slot(long_seq, p) for p in range(lo, hi)]
        assert inp.block_tables[:2].tolist() == [a.block_table + [-1] * (inp.block_tables.shape[1] - len(a.block_table)),
                                                 b.block_table + [-1] * (inp.block_tables.shape[1] - len(b.block_table))]
        sched.postprocess(batch, [next_tok, next_tok + 1, next_tok + 2])
        assert len(a) == la + 1 and len(b) == lb + 1 and a.last_token == next_tok and b.last_token == next_tok + 1
        next_tok += 3
    # 130-token budget minus two decode rows = 128 -> whole blocks: 128 / 128 / 44
    assert spans == [(0, 128), (128, 256), (256, 300)]
    assert len(long_seq) == 301 and long_seq.last_token == next_tok - 1 and sched.prefilling is None
    assert list(sched.running) == [a, b, long_seq]
    assert len(a.block_table) == 2                            # a grew past its first block during the mixed steps
    batch, is_prefill = sched.schedule()                      # no prompt left: an ordinary decode step
    assert not is_prefill and batch == [a, b, long_seq]
    # a prompt that only fits the step alone is prefilled alone first (the reference's order), not starved
    cfg2 = SimpleNamespace(max_num_seqs=8, max_num_batched_tokens=64, eos=-1, num_kvcache_blocks=16,
                           kvcache_block_size=64, max_model_len=512, enable_mixed_batches=True)
    s2 = Scheduler(cfg2)
    r = Sequence([1, 2, 3], sp, block_size=64)
    s2.add(r)
    s2.postprocess(s2.schedule()[0], [9])
    full = Sequence(list(range(64)), sp, block_size=64)       # 64 tokens + 1 decode row > 64
    s2.add(full)
    batch, is_prefill = s2.schedule()
    assert is_prefill and batch == [full]


def test_chunked_prefill_config_gate(tmp_path):
    from minivllm_amd.config import Config

    (tmp_path / "config.json").write_text(json.dumps(dict(
        model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=64, num_hidden_layers=1,
        num_attention_heads=2, num_key_value_heads=1, head_dim=32, intermediate_size=64, vocab_size=100,
        max_position_embeddings=4096, rms_norm_eps=1e-6, tie_word_embeddings=True, hidden_act="silu")))
    with pytest.raises(ValueError, match="max_num_batched_tokens"):      # the reference's rule, by default
        Config(str(tmp_path), max_num_batched_tokens=1024, max_model_len=4096)
    cfg = Config(str(tmp_path), max_num_batched_tokens=1024, max_model_len=4096, enable_chunked_prefill=True)
    assert cfg.max_model_len == 4096


def test_mixed_chunked_steps_never_exceed_max_num_seqs():
    """A chunk in flight must not ride on a step that already holds max_num_seqs decode rows (the staging buffers,
    the sampler planes and the graph buckets are sized for max_num_seqs rows): bs=8, 32-token budget, max_num_seqs=4,
    four running sequences and a 100-token prompt — the chunks then run alone, like the first one does."""
    cfg = SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=32, eos=-1, num_kvcache_blocks=64,
                          kvcache_block_size=8, max_model_len=256, enable_chunked_prefill=True,
                          enable_mixed_batches=True)
    sched = Scheduler(cfg)
    sp = SamplingParams(max_tokens=64, ignore_eos=True)
    for i in range(4):
        sched.add(Sequence([i + 1, i + 2, i + 3], sp, block_size=8))
    batch, is_prefill = sched.schedule()
    assert is_prefill and len(batch) == 4
    sched.postprocess(batch, [7] * 4)
    long_seq = Sequence(list(range(100, 200)), SamplingParams(max_tokens=2, ignore_eos=True), block_size=8)
    sched.add(long_seq)
    sizes = []
    for _ in range(12):
        batch, is_prefill = sched.schedule()
        sizes.append(len(batch))
        assert len(batch) <= cfg.max_num_seqs, sizes
        sched.postprocess(batch, [9] * len(batch))
        if long_seq.num_completion_tokens:
            break
    assert long_seq.num_completion_tokens == 1 and sched.prefilling is None
    assert 1 in sizes                                          # the chunks ran in steps of their own


def test_warmup_runs_when_the_token_budget_is_below_half_the_model_length():
    from minivllm_amd.engine.inference_executor import InferenceExecutor

    assert InferenceExecutor.warmup_sequences(16384, 256, 40960) == 1     # chunked prefill as intended: was 0
    assert InferenceExecutor.warmup_sequences(16384, 256, 4096) == 2      # the reference's count where it is >= 1
    assert InferenceExecutor.warmup_sequences(16384, 1, 4096) == 1


def test_nucleus_batches_fall_back_to_the_gather_graphs_when_their_own_family_cannot_exist():
    """Tensor parallel, a batch with top-p rows and top_k off ('nuc' mode).  Its graph family holds backend
    collectives and is captured lazily, only where the peer-to-peer all-reduce is up.  On an RCCL-only run (or after
    a failed capture) the family stays empty while the gathered-logits graphs exist: such a batch must replay THOSE,
    not drop to an eager step.  The eager path and the replay ask the same helper, so they agree."""
    import torch

    from minivllm_amd.engine.inference_executor import InferenceExecutor

    ex = object.__new__(InferenceExecutor)
    ex.tp_size, ex.sampler, ex.dtype, ex.enforce_eager = 2, object(), torch.bfloat16, False
    ex.config = SimpleNamespace(hf_config=SimpleNamespace(vocab_size=151936))
    ex.greedy_graphs, ex.cand_graphs, ex.mass_graphs, ex.nuc_graphs = {8: "g"}, {}, {}, {}
    ex.graphs, ex.graph_bs = {8: "gather-graph"}, [8]
    ex._general_capture, ex._nuc_capture, ex._cand_capture, ex._mass_capture = "done", "no", "no", "no"
    sp = SamplingParams(temperature=0.8, top_p=0.9, max_tokens=4)
    seqs = [Sequence([1, 2, 3], sp, block_size=64) for _ in range(3)]
    assert ex._tp_mode(seqs) == "nuc"                          # prefill / no graph in play: the cross-rank floor
    assert ex._tp_mode(seqs, decode=True) == "gather"
    assert ex._graph_family(seqs) == ("gather", ex.graphs)
    assert ex._graph_bucket(seqs) == 8                         # was None: the step decoded eagerly
    # the family exists (or can still be captured): 'nuc' stays
    ex.nuc_graphs = {8: "nuc-graph"}
    assert ex._graph_family(seqs) == ("nuc", ex.nuc_graphs) and ex._tp_mode(seqs, decode=True) == "nuc"
    ex.nuc_graphs, ex._nuc_capture = {}, "lazy"
    assert ex._tp_mode(seqs, decode=True) == "nuc"
    # no gather graphs either (gloo ranks: nothing with a backend collective is captured): eager 'nuc', as before
    ex._nuc_capture, ex.graphs, ex._general_capture = "no", {}, "no"
    assert ex._tp_mode(seqs, decode=True) == "nuc" and ex._graph_bucket(seqs) is None
    # enforce_eager never changes the mode
    ex.enforce_eager, ex.graphs = True, {8: "x"}
    assert ex._tp_mode(seqs, decode=True) == "nuc"
    # 'mass' batches (default SamplingParams) without their family already took the gather graphs
    ex.enforce_eager = False
    mass = [Sequence([1, 2, 3], SamplingParams(max_tokens=4), block_size=64)]
    assert ex._graph_family(mass) == ("gather", ex.graphs)
