"""End-to-end parity on the GPU: tiny Qwen3 (qk-norm, tied, G=2, D=128), Qwen2 (qkv bias, untied,
G=7, D=64) and OPT (MHA, learned positions) driven through the build's own
BlockManager + InferenceExecutor, against logits recorded from the REFERENCE running the same
weights / prompts / forced continuation on the CPU (tests/golden/e2e_*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"e2e_{name}.npz"))


def _build(tmp_path, z, dtype, enforce_eager=True, max_num_seqs=8, block_size=64):
    from minivllm_amd.config import Config
    from minivllm_amd.engine.inference_executor import InferenceExecutor
    from minivllm_amd.glue import create_model

    d = tmp_path / "model"
    d.mkdir(exist_ok=True)
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    cfg = Config(str(d), num_kvcache_blocks=16, kvcache_block_size=block_size, max_model_len=512,
                 max_num_batched_tokens=1024, max_num_seqs=max_num_seqs, enforce_eager=enforce_eager,
                 dtype={torch.float32: "float32", torch.float16: "float16"}.get(dtype, "bfloat16"))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(DEV):
            model = create_model(cfg.hf_config)
    finally:
        torch.set_default_dtype(prev)
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    model.load_weights(weights)          # HF names -> fused qkv / gate_up through the packed mapping
    model = model.to(device=DEV, dtype=dtype)
    ex = InferenceExecutor(cfg, model, torch.device(DEV))
    ex._allocate_kv_cache()
    ex._initialize_sampler()
    ex._optimize_model()
    return cfg, model, ex


def _prompts(z):
    lens = z["prompt_lens"].tolist()
    flat = z["prompts"].tolist()
    out, o = [], 0
    for n in lens:
        out.append(flat[o:o + n])
        o += n
    return out


def _run(ex, z, warmup=False, block_size=64):
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.engine.sequence import Sequence
    from minivllm_amd.sampling_params import SamplingParams

    if warmup:
        ex._warmup_model(1024, 8)
    bm = BlockManager(16, block_size)
    seqs = [Sequence(p, SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True), block_size=block_size)
            for p in _prompts(z)]
    for s in seqs:
        bm.allocate(s)
    logits, toks = ex.execute_batch(seqs, prefill=True)
    outs = [(logits.float().cpu(), toks)]
    for step in range(int(z["n_decode"])):
        for s, t in zip(seqs, z["forced_tokens"][step].tolist()):
            s.append_token(int(t))
            bm.may_append(s)
        logits, toks = ex.execute_batch(seqs, prefill=False)
        outs.append((logits.float().cpu(), toks))
    return outs


def _ref(z, tag):
    return [torch.from_numpy(z[f"{tag}_prefill_logits"])] + \
           [torch.from_numpy(z[f"{tag}_decode_logits_{i}"]) for i in range(int(z["n_decode"]))]


@pytest.mark.parametrize("name", ["qwen3", "qwen2", "opt"])
def test_fp32_logits_match_reference(golden_dir, tmp_path, name):
    z = _load(golden_dir, name)
    _, _, ex = _build(tmp_path, z, torch.float32)
    outs = _run(ex, z, warmup=True)
    for i, ((got, toks), want) in enumerate(zip(outs, _ref(z, "f32"))):
        scale = want.abs().max().item()
        # fp32 end to end; only the GEMM / softmax accumulation order differs from the CPU run
        assert (got - want).abs().max().item() <= 2e-4 * scale, f"{name} step {i}"
        assert toks == want.argmax(-1).tolist(), f"{name} step {i}: greedy tokens"


@pytest.mark.parametrize("name", ["qwen3", "qwen2", "opt"])
def test_bf16_logits_within_reference_noise(golden_dir, tmp_path, name):
    """north_star: 'matching the reference eager path logits within 1e-3 bf16'.  The reference's own
    bf16 run differs from its fp32 run by more than that on these models, so the
    bound is max(1e-3 * scale, 1.5 x the reference's bf16-vs-fp32 error), per SURVEY §8c."""
    z = _load(golden_dir, name)
    _, _, ex = _build(tmp_path, z, torch.bfloat16)
    outs = _run(ex, z)
    for i, ((got, toks), want32, want16) in enumerate(zip(outs, _ref(z, "f32"), _ref(z, "bf16"))):
        scale = want32.abs().max().item()
        ref_noise = (want16 - want32).abs().max().item()
        err = (got - want32).abs().max().item()
        assert err <= max(1e-3 * scale, 1.5 * ref_noise), f"{name} step {i}: err {err:.3e} noise {ref_noise:.3e}"
        # greedy token identical wherever the fp32 top-2 margin exceeds the error
        top2 = want32.topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2 * err
        assert torch.equal(torch.tensor(toks)[safe], want32.argmax(-1)[safe])


@pytest.mark.parametrize("name", ["qwen3", "qwen2"])
def test_f16_logits_close_to_reference_fp32(golden_dir, tmp_path, name):
    """float16 end to end (the third dtype of the C-ABI): every kernel in its f16 instantiation.  f16 carries
    3 more mantissa bits than bf16, so the logits must sit inside the bf16 bound of the test above."""
    z = _load(golden_dir, name)
    _, _, ex = _build(tmp_path, z, torch.float16)
    outs = _run(ex, z)
    for i, ((got, toks), want32, want16) in enumerate(zip(outs, _ref(z, "f32"), _ref(z, "bf16"))):
        scale = want32.abs().max().item()
        err = (got - want32).abs().max().item()
        assert err <= max(1e-3 * scale, 1.5 * (want16 - want32).abs().max().item()), f"{name} step {i}: err {err:.3e}"
        top2 = want32.topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2 * err
        assert torch.equal(torch.tensor(toks)[safe], want32.argmax(-1)[safe])


def test_graph_replay_equals_eager(golden_dir, tmp_path):
    """hipGraph decode (model + LM head + sampler in one graph) must reproduce eager bit for bit."""
    z = _load(golden_dir, "qwen3")
    _, _, ex = _build(tmp_path, z, torch.bfloat16)
    eager = _run(ex, z)
    ex2_dir = tmp_path / "g"
    ex2_dir.mkdir()
    _, _, exg = _build(ex2_dir, z, torch.bfloat16, enforce_eager=False)
    exg.capture_device_graphs(8)
    assert exg.graphs and set(exg.graphs) == {1, 2, 4, 8}
    graph = _run(exg, z)
    for i, ((a, ta), (b, tb)) in enumerate(zip(eager, graph)):
        assert torch.equal(a, b), f"step {i}: logits differ between eager and graph replay"
        assert ta == tb


@pytest.mark.parametrize("bs", [128, 256])
def test_other_legal_block_sizes_end_to_end(golden_dir, tmp_path, bs):
    """``Config`` accepts every multiple of 64 as ``kvcache_block_size`` (reference config.py:178-185; its message
    names 64, 128, 256) and the block size only changes where a token's K/V row lives (sequence.py:151-201), never the
    math: the golden logits of the block-64 reference run are the expectation at 128 and 256 too.  fp32 against the
    reference's fp32 logits, bf16 inside the reference's own noise, hipGraph replay == eager bit for bit; then a
    prompt whose first block comes from the prefix cache (the paged form of the MFMA prefill kernel walking blocks of
    this size) followed by decode steps across the block edge, against the same flow at block size 64."""
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.engine.sequence import Sequence

    z = _load(golden_dir, "qwen3")
    d32, d16, dg = tmp_path / "f32", tmp_path / "bf16", tmp_path / "graph"
    for d in (d32, d16, dg):
        d.mkdir()
    _, _, ex32 = _build(d32, z, torch.float32, block_size=bs)
    for i, ((got, toks), want) in enumerate(zip(_run(ex32, z, warmup=True, block_size=bs), _ref(z, "f32"))):
        assert (got - want).abs().max().item() <= 2e-4 * want.abs().max().item(), f"bs {bs} step {i}"
        assert toks == want.argmax(-1).tolist()
    _, _, ex16 = _build(d16, z, torch.bfloat16, block_size=bs)
    eager = _run(ex16, z, block_size=bs)
    for i, ((got, _), want32, want16) in enumerate(zip(eager, _ref(z, "f32"), _ref(z, "bf16"))):
        err = (got - want32).abs().max().item()
        assert err <= max(1e-3 * want32.abs().max().item(), 1.5 * (want16 - want32).abs().max().item()), f"bs {bs} step {i}"
    _, _, exg = _build(dg, z, torch.bfloat16, enforce_eager=False, block_size=bs)
    exg.capture_device_graphs(8)
    assert exg.graphs
    for i, ((a, ta), (b, tb)) in enumerate(zip(eager, _run(exg, z, block_size=bs))):
        assert torch.equal(a, b) and ta == tb, f"bs {bs} step {i}: graph replay differs from eager"

    def cached_flow(ex, block):
        """prefill A (block + 3 tokens), prefill B (same first block + 5 others: a prefix-cache hit), 4 decode steps of B"""
        bm = BlockManager(16, block)
        shared = [(7 * i + 3) % 300 for i in range(block)]
        a = Sequence(shared + [200, 201, 202], block_size=block)
        b = Sequence(shared + [210, 211, 212, 213, 214], block_size=block)
        bm.allocate(a)
        ex.execute_batch([a], prefill=True)
        bm.allocate(b)
        assert b.num_cached_tokens == block and b.block_table[0] == a.block_table[0]
        outs = [ex.execute_batch([b], prefill=True)[0].float().cpu()]
        for t in (11, 12, 13, 14):
            b.append_token(t)
            bm.may_append(b)
            outs.append(ex.execute_batch([b], prefill=False)[0].float().cpu())
        return outs

    ref_dir = tmp_path / "ref64"
    ref_dir.mkdir()
    _, _, ex64 = _build(ref_dir, z, torch.float32, block_size=64)
    # at block 64 the same token list spans bs / 64 cached blocks: same cached-prefix length only when it is whole
    # blocks in both layouts -- compare against a clean (uncached) block-64 run of the same token lists instead
    bm = BlockManager(16, 64)
    shared = [(7 * i + 3) % 300 for i in range(bs)]
    b64 = Sequence(shared + [210, 211, 212, 213, 214], block_size=64)
    bm.allocate(b64)
    assert b64.num_cached_tokens == 0
    want = [ex64.execute_batch([b64], prefill=True)[0].float().cpu()]
    for t in (11, 12, 13, 14):
        b64.append_token(t)
        bm.may_append(b64)
        want.append(ex64.execute_batch([b64], prefill=False)[0].float().cpu())
    for i, (g_, w_) in enumerate(zip(cached_flow(ex32, bs), want)):
        torch.testing.assert_close(g_, w_, rtol=1e-4, atol=1e-5 * w_.abs().max().item(), msg=lambda m: f"bs {bs} cached flow step {i}: {m}")


def test_prefix_cached_prefill_matches_full_prefill(golden_dir, tmp_path):
    """A prompt whose first block is served by the prefix cache must give the logits of a clean
    full prefill (the reference's fallback gets this wrong)."""
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.engine.sequence import Sequence

    z = _load(golden_dir, "qwen3")
    _, _, ex = _build(tmp_path, z, torch.float32)
    bm = BlockManager(16, 64)
    shared = list(range(5, 69))
    a = Sequence(shared + [200, 201, 202], block_size=64)
    b = Sequence(shared + [210, 211, 212, 213, 214], block_size=64)
    bm.allocate(a)
    la, _ = ex.execute_batch([a], prefill=True)
    bm.allocate(b)
    assert b.num_cached_tokens == 64
    lb, _ = ex.execute_batch([b], prefill=True)           # keys 0..63 come from the paged cache
    bm2 = BlockManager(16, 64)
    for blk in (0, 1, 2):  # occupy the blocks used above so that nothing is shared
        bm2._take(blk)
    b2 = Sequence(shared + [210, 211, 212, 213, 214], block_size=64)
    bm2.allocate(b2)
    assert b2.num_cached_tokens == 0
    lb2, _ = ex.execute_batch([b2], prefill=True)
    torch.testing.assert_close(lb.float(), lb2.float(), rtol=1e-4, atol=1e-5)


def test_engine_generate_greedy_eager_equals_graphs(golden_dir, tmp_path):
    """LLM(Config).generate over the real ModelRunner: prompts as token-id lists (the directory
    has no tokenizer), greedy decoding; eager execution and hipGraph replay give the same tokens."""
    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    results = []
    for eager in (True, False):
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=1024, max_num_seqs=4, enforce_eager=eager,
                     dtype="bfloat16", num_kvcache_blocks=32, load_format="dummy", seed=0)
        engine = LLM(cfg)
        engine.model_runner.model_manager.model.load_weights(weights)   # golden weights over the dummy init
        assert bool(engine.model_runner.inference_executor.graphs) == (not eager)
        outs = engine.generate(_prompts(z) + [[7, 7, 7]],
                               SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True), use_tqdm=False)
        results.append([o["token_ids"] for o in outs])
        assert len(outs) == 4 and all(len(t) == 12 for t in results[-1])
        engine.exit()
    assert results[0] == results[1]


def test_full_batch_off_the_bucket_grid_runs_on_a_graph(golden_dir, tmp_path):
    """max_num_seqs = 24 is not a multiple of 16: the bucket list ends with 24 itself, so a full batch of 24
    decodes on a captured graph (speculatively, too) and produces the eager tokens for nearly every request."""
    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    prompts = [[3 + (5 * i + j) % 290 for j in range(4 + i)] for i in range(24)]
    results, stats = [], []
    for eager in (True, False):
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=1024, max_num_seqs=24, enforce_eager=eager,
                     dtype="bfloat16", num_kvcache_blocks=40, load_format="dummy", seed=0)
        engine = LLM(cfg)
        engine.model_runner.model_manager.model.load_weights(weights)
        ex = engine.model_runner.inference_executor
        outs = engine.generate(prompts, SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True), use_tqdm=False)
        results.append([o["token_ids"] for o in outs])
        stats.append((sorted(ex.graphs), ex.spec_hits))
        engine.exit()
    assert stats[1][0] == [1, 2, 4, 8, 12, 16, 24] and stats[1][1] > 5 and stats[0] == ([], 0)
    # bucket 24 for 24 rows: no padding, the same split-KV plan as eager -> bit-identical step inputs
    assert results[0] == results[1]


def test_llm_from_checkpoint_dir_with_tokenizer(golden_dir, tmp_path):
    """The full user path: a model directory with config.json + model.safetensors (HF names) + a
    tokenizer; ``LLM(Config(dir))`` loads it (load_format="auto"), string prompts are tokenised,
    EOS comes from the tokenizer, text comes back."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).contiguous()
               for k in z.files if k.startswith("w:")}
    save_file(weights, str(d / "model.safetensors"))
    vocab = {f"w{i}": i for i in range(318)}
    vocab.update({"<eos>": 318, "<unk>": 319})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<eos>", unk_token="<unk>").save_pretrained(str(d))

    cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=1024, max_num_seqs=4, dtype="bfloat16",
                 num_kvcache_blocks=32)
    llm = LLM(cfg)
    assert llm.tokenizer is not None and cfg.eos == 318
    prompts = ["w3 w4 w5 w6 w7", "w100 w101", "w9 " * 70]
    outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=10), use_tqdm=False)
    assert len(outs) == 3 and all(isinstance(o["text"], str) for o in outs)
    assert all(1 <= len(o["token_ids"]) <= 10 for o in outs)
    # the same prompts as token ids give the same continuation (greedy); the prefix cache now holds them
    ids = [llm.tokenizer.encode(p) for p in prompts]
    again = llm.generate(ids, SamplingParams(temperature=0.0, max_tokens=10), use_tqdm=False)
    assert [o["token_ids"] for o in again] == [o["token_ids"] for o in outs]
    # sampling path through the engine (reference preset of examples/cpu_inference_opt.py)
    sampled = llm.generate(prompts, SamplingParams(temperature=0.6, top_p=0.95, top_k=40, max_tokens=8, ignore_eos=True),
                           use_tqdm=False)
    assert all(len(o["token_ids"]) == 8 and all(0 <= t < 320 for t in o["token_ids"]) for o in sampled)
    ex = llm.model_runner.inference_executor
    assert ex.total_decode_tokens > 0 and ex.total_prefill_tokens > 0 and ex.graphs
    llm.exit()


def test_speculative_replay_matches_eager_across_finishes(golden_dir, tmp_path):
    """Requests finishing at different steps (max_tokens 3..40, one EOS hit), a block boundary inside the
    run and a late-arriving request: the graph path with speculative next-step replay must produce
    exactly the eager tokens, and must actually have speculated."""
    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    prompts = [list(range(3, 3 + n)) for n in (60, 5, 63, 20, 64)]      # 63/64: block edges while decoding
    sps = [SamplingParams(temperature=0.0, max_tokens=m, ignore_eos=(i != 1)) for i, m in enumerate((3, 40, 9, 17, 30))]
    results, stats = [], []
    for eager in (True, False):
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=1024, max_num_seqs=8, enforce_eager=eager,
                     dtype="bfloat16", num_kvcache_blocks=32, load_format="dummy", seed=0)
        engine = LLM(cfg)
        engine.model_runner.model_manager.model.load_weights(weights)
        if not results:  # pick an EOS id that request 1 really emits (from a dry greedy run)
            dry = engine.generate([prompts[1]], SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True), use_tqdm=False)
            eos = dry[0]["token_ids"][12]
        engine.scheduler.eos = eos
        outs = engine.generate(prompts, sps, use_tqdm=False)
        results.append([o["token_ids"] for o in outs])
        ex = engine.model_runner.inference_executor
        stats.append((ex.spec_hits, ex.spec_misses))
        engine.exit()
    assert results[0] == results[1]
    assert len(results[0][1]) <= 13 and results[0][1][-1] == eos          # stopped at the EOS
    assert [len(t) for i, t in enumerate(results[0]) if i != 1] == [3, 9, 17, 30]
    assert stats[0] == (0, 0) and stats[1][0] > 5                          # eager never speculates; graphs did


def test_kv_cache_auto_sizing_and_graph_buckets(golden_dir, tmp_path):
    """a10: with num_kvcache_blocks=-1 the pool is min(memory budget // block bytes, max_num_seqs *
    ceil(max_model_len / block)) and is written back into the Config (the scheduler sizes itself from it)."""
    from minivllm_amd.engine.inference_executor import graph_batch_sizes, kv_block_bytes

    z = _load(golden_dir, "qwen3")
    cfg, model, ex = _build(tmp_path, z, torch.bfloat16)
    cfg.num_kvcache_blocks = -1
    ex.kv_cache = None
    ex._allocate_kv_cache()
    assert cfg.num_kvcache_blocks == 8 * (512 // 64)               # the cap: memory is not the limit here
    L, n, bs, hkv, d = ex.kv_cache.shape[1:]
    assert (L, bs, hkv, d) == (2, 64, 2, 128)
    assert ex.kv_cache.numel() * 2 == n * kv_block_bytes(2, 64, 2, 128, 2)
    att = model.model.layers[1].self_attn.attn
    assert att.k_cache.data_ptr() == ex.kv_cache[0, 1].data_ptr() and att.v_cache.data_ptr() == ex.kv_cache[1, 1].data_ptr()
    assert graph_batch_sizes(256) == [1, 2, 4, 8, 12] + list(range(16, 257, 16))
    assert graph_batch_sizes(8) == [1, 2, 4, 8] and graph_batch_sizes(1000)[-1] == 512
    assert graph_batch_sizes(24) == [1, 2, 4, 8, 12, 16, 24] and graph_batch_sizes(3) == [1, 2, 3]
    assert graph_batch_sizes(1) == [1]


def test_config5_per_gpu_layer_shape_qwen2_72b_tp8(tmp_path):
    """BASELINE config 5 as far as one GPU goes: the PER-GPU shard of Qwen2-72B at TP = 8 — hidden 8192, 8 query
    heads / 1 kv head of 128, intermediate 29568 / 8 = 3696, qkv bias — as a 2-layer model (no collectives: what a
    rank computes between its all-reduces), bf16, against the fp32 oracle on the same seeded weights: prefill of
    three prompts, then decode steps at a batch that takes the decode-batch projection kernels (40 rows; down_proj's
    K = 3696 is not a multiple of 64 and stays on the library GEMM).  Bound: max(1e-3 x scale, 1.5 x the error of the
    oracle's OWN bf16 run against its fp32 run) — the bar of the other end-to-end tests; both errors are printed."""
    import golden_inputs as GI
    from oracle.model import OracleModel

    from minivllm_amd.config import Config
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.engine.inference_executor import InferenceExecutor
    from minivllm_amd.engine.input_builder import InputBuilder
    from minivllm_amd.engine.sequence import Sequence
    from minivllm_amd.glue import create_model
    from minivllm_amd.sampling_params import SamplingParams

    cfg_d = dict(model_type="qwen2", architectures=["Qwen2ForCausalLM"], hidden_size=8192, num_hidden_layers=2,
                 num_attention_heads=8, num_key_value_heads=1, head_dim=128, intermediate_size=3696, vocab_size=2048,
                 max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False,
                 hidden_act="silu", torch_dtype="bfloat16")
    (tmp_path / "config.json").write_text(json.dumps(cfg_d))
    hfw = GI.seeded_hf_weights(cfg_d, seed=5)
    nseq = 40
    cfg = Config(str(tmp_path), num_kvcache_blocks=3 * nseq, max_model_len=192, max_num_batched_tokens=4096,
                 max_num_seqs=64, enforce_eager=True, dtype="bfloat16")
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(DEV):
            model = create_model(cfg.hf_config)
    finally:
        torch.set_default_dtype(prev)
    model.load_weights({k: v.float() for k, v in hfw.items()})
    model = model.to(device=DEV, dtype=torch.bfloat16)
    for mod in model.modules():                      # what ModelManager does after loading
        if hasattr(mod, "pack_for_decode"):
            mod.pack_for_decode()
    att = model.model.layers[0].self_attn
    assert (att.num_heads, att.num_kv_heads, att.head_dim) == (8, 1, 128) and att.qkv_proj.weight.shape == (1280, 8192)
    assert att.qkv_proj.weight_packed is not None and model.model.layers[0].mlp.down_proj.weight_packed is None
    ex = InferenceExecutor(cfg, model, torch.device(DEV))
    ex._allocate_kv_cache(); ex._initialize_sampler(); ex._optimize_model()
    ref = OracleModel(cfg_d, {k: v.float() for k, v in hfw.items()}, torch.float32, num_blocks=3 * nseq)
    ref16 = OracleModel(cfg_d, {k: v.float() for k, v in hfw.items()}, torch.bfloat16, num_blocks=3 * nseq)
    g = torch.Generator().manual_seed(0)
    greedy = SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True)
    seqs = [Sequence(torch.randint(0, 2048, (int(n),), generator=g).tolist(), greedy, block_size=64)
            for n in torch.randint(3, 100, (nseq,), generator=g)]
    bm, ib = BlockManager(3 * nseq, 64), InputBuilder(64, 64, 192, 4096)
    for s in seqs:
        bm.allocate(s)
    worst = worst_noise = 0.0
    for step in range(3):
        prefill = step == 0
        inp = ib.prefill(seqs) if prefill else ib.decode(seqs)
        meta = dict(is_prefill=prefill, slot_mapping=inp.slot_mapping, block_tables=inp.block_tables,
                    cu_q=inp.cum_seqlens_q, cu_k=inp.cum_seqlens_k, context_lens=inp.context_lens)
        want = ref.forward_logits(inp.input_ids.clone(), inp.positions.clone(), meta).float()
        # the noise floor of bf16 on THIS model: the oracle's own bf16 run (reference op order and rounding points,
        # same tokens, its own bf16 KV cache) against its fp32 run
        noise = (ref16.forward_logits(inp.input_ids.clone(), inp.positions.clone(), meta).float() - want).abs().max().item()
        got, _ = ex.execute_batch(seqs, prefill=prefill)
        got = got.float().cpu()
        scale = want.abs().max().item()
        err = (got - want).abs().max().item()
        worst = max(worst, err / scale)
        worst_noise = max(worst_noise, noise / scale)
        # the bar of every other e2e test: max(1e-3 * scale, 1.5 x the bf16 noise of the eager op order)
        assert err <= max(1e-3 * scale, 1.5 * noise), f"step {step}: err {err:.4f} noise {noise:.4f} scale {scale:.3f}"
        assert torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0).item() > 0.9995
        for s, t in zip(seqs, want.argmax(-1).tolist()):
            s.append_token(int(t))
            bm.may_append(s)
    print(f"\nconfig 5 per-GPU shard (hidden 8192, 8/1 heads, inter 3696), bf16 vs fp32 oracle: worst error {worst:.4%} of the "
          f"logit scale (the oracle's own bf16 run: {worst_noise:.4%})")


def test_chunked_prefill_equals_one_shot_prefill(golden_dir, tmp_path):
    """SURVEY §8 f4 / BASELINE config 4: a 300-token prompt prefilled in chunks of 128 (each chunk attends the
    earlier ones through the paged cache) must produce the logits of the one-shot prefill, and the same greedy
    continuation."""
    from safetensors.torch import save_file

    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    save_file({k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).contiguous()
               for k in z.files if k.startswith("w:")}, str(d / "model.safetensors"))
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, 320, (n,), generator=g).tolist() for n in (300, 5, 129)]
    sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
    outs, logits = [], []
    for budget, chunked in ((1024, False), (128, True)):
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=budget, max_num_seqs=4, num_kvcache_blocks=24,
                     enforce_eager=True, dtype="bfloat16", enable_chunked_prefill=chunked)
        engine = LLM(cfg)
        ex = engine.model_runner.inference_executor
        seen = []
        real = ex.execute_batch
        ex.execute_batch = lambda seqs, prefill=True, _r=real, _s=seen: (lambda r: (_s.append((prefill, [s.prefill_stop - s.num_cached_tokens for s in seqs], r[0].float().cpu())), r)[1])(_r(seqs, prefill))
        outs.append([o["token_ids"] for o in engine.generate(prompts, sp, use_tqdm=False)])
        logits.append(seen)
        engine.exit()
    assert outs[0] == outs[1]
    one_shot = [x for x in logits[0] if x[0]]
    chunked = [x for x in logits[1] if x[0]]
    assert [x[1] for x in chunked][:3] == [[128], [128], [44]]            # the long prompt: three chunks, alone
    # last-token logits of the long prompt: final chunk vs the one-shot prefill row of the same sequence
    torch.testing.assert_close(chunked[2][2][0], one_shot[0][2][0], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunked_prefill_matches_the_reference_one_shot_logits(golden_dir, tmp_path, dtype):
    """An INDEPENDENT pin for the chunked / paged form of the prefill kernel (the self-consistency test above would pass a
    bug shared by both runs): the golden prompts (70, 5 and 65 tokens) prefilled with a 64-token budget — the 70- and
    65-token prompts in two chunks each, the second chunk attending the first through the paged cache — must give the
    REFERENCE's one-shot prefill logits (tests/golden/e2e_qwen3.npz, recorded from the reference in fp32 and bf16),
    and the forced decode steps behind them the reference's decode logits."""
    from minivllm_amd.engine.scheduler import Scheduler
    from minivllm_amd.engine.sequence import Sequence
    from minivllm_amd.sampling_params import SamplingParams

    z = _load(golden_dir, "qwen3")
    cfg, _, ex = _build(tmp_path, z, dtype)
    cfg.max_num_batched_tokens, cfg.enable_chunked_prefill = 64, True
    sched = Scheduler(cfg)
    sp = SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True)
    seqs = [Sequence(p, sp, block_size=64) for p in _prompts(z)]
    for s in seqs:
        sched.add(s)
    forced = z["forced_tokens"]
    rows, chunks = {}, []
    while sched.waiting or sched.prefilling is not None:
        batch, is_prefill = sched.schedule()
        assert is_prefill
        chunks.append([s.prefill_stop - s.num_cached_tokens for s in batch])
        logits, _ = ex.execute_batch(batch, prefill=True)
        logits = logits.float().cpu()
        toks = []
        for i, s in enumerate(batch):
            if s.prefill_end is None:                      # a last (or only) chunk: its row is the prompt's logits
                rows[seqs.index(s)] = logits[i]
            toks.append(int(forced[0][seqs.index(s)]))
        sched.postprocess(batch, toks)
    assert chunks == [[64], [6], [5], [64], [1]]            # 70 = 64 + 6, 5, 65 = 64 + 1
    got = torch.stack([rows[i] for i in range(3)])
    want32, want16 = torch.from_numpy(z["f32_prefill_logits"]), torch.from_numpy(z["bf16_prefill_logits"])
    scale = want32.abs().max().item()
    err = (got - want32).abs().max().item()
    if dtype == torch.float32:
        assert err <= 2e-4 * scale, f"chunked prefill vs reference fp32: {err:.3e}"
        assert got.argmax(-1).tolist() == want32.argmax(-1).tolist()
    else:
        noise = (want16 - want32).abs().max().item()
        assert err <= max(1e-3 * scale, 1.5 * noise), f"chunked prefill vs reference: err {err:.3e} noise {noise:.3e}"
    # the decode steps behind the chunked prefill read the K/V rows the chunks stored
    by_id = {s.seq_id: i for i, s in enumerate(seqs)}
    for step in range(2):
        batch, is_prefill = sched.schedule()
        assert not is_prefill and len(batch) == 3
        logits, _ = ex.execute_batch(batch, prefill=False)
        order = [by_id[s.seq_id] for s in batch]
        got = logits.float().cpu()
        want = torch.from_numpy(z[f"f32_decode_logits_{step}"])[order]
        err = (got - want).abs().max().item()
        if dtype == torch.float32:
            assert err <= 2e-4 * scale, f"decode step {step} after chunked prefill: {err:.3e}"
        else:
            noise = (torch.from_numpy(z[f"bf16_decode_logits_{step}"])[order] - want).abs().max().item()
            assert err <= max(1e-3 * scale, 1.5 * noise)
        sched.postprocess(batch, [int(forced[step + 1][i]) for i in order])


def test_mixed_batches_equal_separate_steps(golden_dir, tmp_path):
    """SURVEY §8 f4, mixed prefill/decode batches: with enable_mixed_batches the running sequences decode inside the
    prefill steps (paged-decode kernel for their rows, prefill kernel for the prompt chunk, one pass of the
    projections for both).  Greedy continuations must equal those of the separate-step schedule, the mixed steps
    must really have happened, and the decode rows' logits must match the pure decode step's."""
    from safetensors.torch import save_file

    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    save_file({k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).contiguous()
               for k in z.files if k.startswith("w:")}, str(d / "model.safetensors"))
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 320, (n,), generator=g).tolist() for n in (5, 40, 300, 70)]
    sp = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    outs, steps = [], []
    for mixed in (False, True):
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=130, max_num_seqs=8, num_kvcache_blocks=24,
                     enforce_eager=True, dtype="bfloat16", enable_chunked_prefill=True, enable_mixed_batches=mixed)
        engine = LLM(cfg)
        ex = engine.model_runner.inference_executor
        seen = []
        real = ex.execute_batch

        def spy(seqs, prefill=True, _r=real, _s=seen):
            shape = [s.prefill_stop - s.num_cached_tokens for s in seqs] if prefill else [1] * len(seqs)
            r = _r(seqs, prefill)
            _s.append((prefill, shape, [s.seq_id for s in seqs], r[0].float().cpu()))
            return r
        ex.execute_batch = spy
        outs.append([o["token_ids"] for o in engine.generate(prompts, sp, use_tqdm=False)])
        steps.append(seen)
        engine.exit()
    assert outs[0] == outs[1]
    plain = [x[1] for x in steps[0] if x[0]]
    mixed = [x[1] for x in steps[1] if x[0]]
    assert plain == [[5, 40], [128], [128], [44], [70]]               # separate steps: the chunks run alone
    assert mixed == [[5, 40], [1, 1, 128], [1, 1, 128], [1, 1, 44], [1, 1, 1, 70]]
    # the first sequence is done four steps earlier: it decoded inside the four mixed steps
    last = [max(i for i, x in enumerate(st) if x[2][0] == st[0][2][0]) for st in steps]
    assert last[1] == last[0] - 4
    # the decode rows of the first mixed step against the first pure decode step of the other schedule (same rows)
    first_mixed = next(x for x in steps[1] if x[0] and x[1][0] == 1)
    first_decode = next(x for x in steps[0] if not x[0])
    scale = float(first_decode[3].abs().max())
    torch.testing.assert_close(first_mixed[3][:2], first_decode[3][:2], rtol=0, atol=2e-2 * scale)


def test_kv_pool_sized_by_hbm_not_by_the_cap(tmp_path):
    """a10 on the real part: num_kvcache_blocks=-1 with limits so large (max_num_seqs 512 x max_model_len 40960 =
    327 680 blocks = 2.2 TiB of Qwen3-1.7B KV) that the 288 GB of HBM, not the cap, decide.  The pool must fit the
    budget util * total - used, be allocated, serve a request, and be addressable to its LAST block (7 MiB blocks:
    byte offsets beyond 2^37).  Formula: minivllm/engine/inference_executor.py:245-318 with the local kv heads."""
    import gc

    import bench
    from oracle import ops as O

    from minivllm_amd import LLM, Config, SamplingParams, hip_ops
    from minivllm_amd.engine.inference_executor import kv_block_bytes

    (tmp_path / "config.json").write_text(json.dumps(bench.MODELS["qwen3-1.7b"]))
    free0, total = torch.cuda.mem_get_info()
    cfg = Config(str(tmp_path), dtype="bfloat16", kvcache_block_size=64, max_num_seqs=512, max_model_len=40960,
                 max_num_batched_tokens=40960, enforce_eager=True, load_format="dummy", seed=0, num_kvcache_blocks=-1)
    engine = LLM(cfg)
    try:
        ex = engine.model_runner.inference_executor
        n = cfg.num_kvcache_blocks
        bb = kv_block_bytes(28, 64, 8, 128, 2)
        assert bb == 7 << 20
        assert 0 < n < 512 * (40960 // 64), "the cap decided: this GPU is not memory-limited for the test's limits"
        assert ex.kv_cache.shape == (2, 28, n, 64, 8, 128)
        used_before_kv = total - free0 + 0   # other processes + this context before the engine existed
        assert n * bb <= cfg.device_memory_utilization * total - used_before_kv
        # not wasteful either: what is left of the budget after weights + pool is less than 4 GiB of slack
        free1, _ = torch.cuda.mem_get_info()
        budget_left = cfg.device_memory_utilization * total - (total - free1)
        print(f"\nKV pool on this GPU: {n} blocks x 7 MiB = {n * bb / 2**30:.1f} GiB of {total / 2**30:.1f} GiB "
              f"(util {cfg.device_memory_utilization}); budget left {budget_left / 2**30:.2f} GiB")
        assert -1 * 2**30 <= budget_left <= 4 * 2**30, budget_left
        # the engine serves with it
        outs = engine.generate([[5, 6, 7, 8, 9], list(range(100, 170))], SamplingParams(temperature=0.0, max_tokens=5,
                                                                                         ignore_eos=True), use_tqdm=False)
        assert [len(o["token_ids"]) for o in outs] == [5, 5]
        # the last block of the last layer is real memory at the right address: write it, read it through the kernel
        kc, vc = ex.kv_cache[0, 27], ex.kv_cache[1, 27]
        kc[n - 1].normal_(); vc[n - 1].normal_(); kc[0].normal_(); vc[0].normal_()
        q = torch.randn(2, 16, 128, device=DEV).bfloat16()
        table = torch.tensor([[n - 1, 0], [0, n - 1]], dtype=torch.int32, device=DEV)
        ctx = torch.tensor([64, 100], dtype=torch.int32, device=DEV)
        out = hip_ops.paged_decode(q, kc, vc, table, ctx, 128 ** -0.5)
        small_k, small_v = torch.stack([kc[0], kc[n - 1]]).cpu(), torch.stack([vc[0], vc[n - 1]]).cpu()
        ref = O.paged_decode(q.cpu(), small_k, small_v, torch.tensor([[1, 0], [0, 1]], dtype=torch.int32), ctx.cpu(), 128 ** -0.5)
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2 ** -7, atol=2 ** -7)
    finally:
        engine.exit()
        del engine
        gc.collect()
        torch.cuda.empty_cache()


def test_randomized_serving_speculation_changes_nothing(golden_dir, tmp_path, monkeypatch):
    """Continuous batching under pressure: 40 requests (greedy, nucleus and top-k sampled rows mixed in one
    batch) with random lengths / budgets / shared prefixes arriving while others decode, a KV pool small enough to force preemption and recomputation, prefix-cache
    hits, block edges, EOS stops.  The captured-graph path is run twice, with and without the speculative
    replay of the next step: every request must get exactly the same tokens (same buckets, same kernels; the
    speculation only changes WHEN a step is enqueued).  Against eager decoding the batch is padded to a
    bucket, which may move a split-KV boundary, so only agreement of nearly all requests is asked there."""
    import random

    from minivllm_amd import LLM, Config, SamplingParams

    z = _load(golden_dir, "qwen3")
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    rng = random.Random(7)
    shared = [rng.randrange(3, 300) for _ in range(70)]                   # > one block: prefix-cache hits
    reqs = []
    for i in range(40):
        body = [rng.randrange(3, 300) for _ in range(rng.choice((1, 5, 20, 62, 63, 64, 65, 100)))]
        prompt = (shared + body) if i % 3 == 0 else body
        temp, top_p, top_k = ((0.0, 1.0, -1), (0.8, 0.9, -1), (1.0, 1.0, 20))[i % 3 if i % 2 else 0]
        reqs.append((prompt, SamplingParams(temperature=temp, top_p=top_p, top_k=top_k,
                                            max_tokens=rng.choice((1, 2, 7, 20, 33, 70)),
                                            ignore_eos=(i % 4 != 0)), rng.randrange(0, 60)))

    def serve(eager, speculate, mixed=False):
        monkeypatch.setenv("MV_DECODE_SPECULATE", "1" if speculate else "0")
        cfg = Config(str(d), max_model_len=512, max_num_batched_tokens=128 if mixed else 512, max_num_seqs=8,
                     enforce_eager=eager, dtype="bfloat16", num_kvcache_blocks=14, load_format="dummy", seed=0,
                     enable_chunked_prefill=mixed, enable_mixed_batches=mixed)
        engine = LLM(cfg)
        engine.model_runner.model_manager.model.load_weights(weights)
        engine.scheduler.eos = 17
        pending = sorted(range(len(reqs)), key=lambda i: reqs[i][2])
        ids, done, step = {}, {}, 0
        while pending or not engine.is_finished():
            while pending and reqs[pending[0]][2] <= step:
                i = pending.pop(0)
                ids[engine.add_request(reqs[i][0], reqs[i][1])] = i
            if not engine.is_finished():
                outs, _ = engine.step()
                for sid, toks in outs:
                    done[ids[sid]] = toks
            step += 1
            assert step < 5000
        ex = engine.model_runner.inference_executor
        stats = (ex.spec_hits, ex.spec_misses, engine.scheduler.num_preemptions, step)
        engine.exit()
        return [done[i] for i in range(len(reqs))], stats

    plain, st0 = serve(eager=False, speculate=False)
    spec, st1 = serve(eager=False, speculate=True)
    assert st0[0] == 0 and st1[0] > 20, (st0, st1)                      # really speculated (arrivals discard some)
    assert st0[2] == st1[2] > 0, (st0, st1)                             # the pool did run dry: preemptions
    assert spec == plain
    # (the Philox stream of a step is keyed by the step count, which the device-side advance reproduces: sampled
    # rows are identical too.)  Eager: same scheduler, same seeds -- only the split-KV padding differs
    eager, _ = serve(eager=True, speculate=False)
    agree = sum(a == b for a, b in zip(eager, plain))
    print(f"eager vs graph: {agree} of {len(reqs)} requests identical")
    # measured: 40 of 40.  Eager steps run unpadded batches, graph steps the next bucket: a different split-KV plan
    # can move a logit by an ulp, and a sampled row sitting on a probability edge then draws another token and
    # diverges for good — a slack of 2 requests for that, no more (a real regression loses most of them)
    assert agree >= len(reqs) - 2
    # mixed prefill/decode batches + chunked prefill with a 128-token step budget (graphs on): another schedule
    # altogether (decode rows ride in the prefill steps, 170-token prompts come in two chunks), so sampled rows draw
    # from other Philox steps — the greedy requests must still say the same thing
    mixed, st3 = serve(eager=False, speculate=True, mixed=True)
    greedy = [i for i, r in enumerate(reqs) if r[1].temperature == 0.0]
    agree_mixed = sum(mixed[i] == plain[i] for i in greedy)
    print(f"mixed batches vs separate steps: {agree_mixed} of {len(greedy)} greedy requests identical; "
          f"{st3[3]} engine steps against {st1[3]}, {st3[2]} preemptions")
    assert all(len(mixed[i]) == len(plain[i]) for i in greedy)
    assert agree_mixed >= len(greedy) - 2
