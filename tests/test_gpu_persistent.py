"""The persistent small-batch decode launch (csrc/decode_persistent.hip: every decoder layer of a step of <= 16 rows in ONE
launch, phases separated by fence-free grid barriers, the next projection's weights streaming into LDS across them) against
the multi-launch path it replaces — BIT for bit: same K slices, same summation order, same rounding points, the attention
and combine bodies shared (csrc/paged_decode_body.h).  Op order under test: reference minivllm/models/qwen_base.py:173-190
(attention block), :232-236 (MLP), :281-298 (decoder layer)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

QWEN3_17B_3L = dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=2048, num_hidden_layers=3,
                    num_attention_heads=16, num_key_value_heads=8, head_dim=128, intermediate_size=6144, vocab_size=4096,
                    max_position_embeddings=8192, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=True,
                    hidden_act="silu", attention_bias=False, torch_dtype="bfloat16", eos_token_id=1)
# Qwen2 flavour at head_dim 128: qkv bias, no head norms, 2 query heads per kv head, kv heads not a multiple of 4 (the
# per-head attention mapping), hidden 512 (2 K slices; the gated form: 4 slices of 4 k-steps -> the 4-step chunks),
# intermediate 1536 (6 K slices for down_proj)
QWEN2_LIKE = dict(model_type="qwen2", architectures=["Qwen2ForCausalLM"], hidden_size=512, num_hidden_layers=2,
                  num_attention_heads=4, num_key_value_heads=2, intermediate_size=1536, vocab_size=2048,
                  max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=10000.0, tie_word_embeddings=False,
                  hidden_act="silu", torch_dtype="bfloat16", eos_token_id=1)


def _executor(tmp_path, cfg_dict, dtype=torch.bfloat16, blocks=160, max_num_seqs=16, eager=True, block_size=64, seed=0):
    from minivllm_amd.config import Config
    from minivllm_amd.engine.inference_executor import InferenceExecutor
    from minivllm_amd.engine.model_manager import ModelManager
    from minivllm_amd.glue import create_model

    d = tmp_path / f"m{seed}_{len(os.listdir(tmp_path))}"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg_dict))
    cfg = Config(str(d), num_kvcache_blocks=blocks, kvcache_block_size=block_size, max_model_len=2048,
                 max_num_batched_tokens=4096, max_num_seqs=max_num_seqs, enforce_eager=eager,
                 dtype={torch.float16: "float16"}.get(dtype, "bfloat16"))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(DEV):
            model = create_model(cfg.hf_config)
    finally:
        torch.set_default_dtype(prev)
    ModelManager._dummy_init(model, seed)
    with torch.no_grad():   # norm weights and biases away from their neutral values
        g = torch.Generator(device=DEV).manual_seed(seed + 1)
        for name, p in model.named_parameters():
            if "norm" in name:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g, device=DEV))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g, device=DEV))
    model = model.to(device=DEV, dtype=dtype)
    for mod in model.modules():
        if hasattr(mod, "pack_for_decode"):
            mod.pack_for_decode()
    ex = InferenceExecutor(cfg, model, torch.device(DEV))
    ex._allocate_kv_cache()
    ex._initialize_sampler()
    ex._optimize_model()
    return cfg, model, ex


def _sequences(lens, vocab, block_size, bm, seed=3):
    from minivllm_amd.engine.sequence import Sequence
    from minivllm_amd.sampling_params import SamplingParams

    g = torch.Generator().manual_seed(seed)
    seqs = [Sequence(torch.randint(2, vocab, (n,), generator=g).tolist(),
                     SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True), block_size=block_size) for n in lens]
    for s in seqs:
        bm.allocate(s)
    return seqs


def _both_ways(ex, seqs, monkeypatch):
    """One decode step of the same state twice — launches, then the persistent kernel — (logits, kv copies, launched?)."""
    from minivllm_amd.glue import qwen

    out = []
    for on in (False, True):
        monkeypatch.setattr(qwen, "PERSISTENT", on)
        logits, toks = ex.execute_batch(seqs, prefill=False)
        torch.cuda.synchronize()
        out.append((logits.clone(), toks, ex.kv_cache.clone()))
    return out


@pytest.mark.parametrize("cfg_dict,lens", [
    (QWEN3_17B_3L, [1023]),                                    # B = 1, the sweep's cell
    (QWEN3_17B_3L, [1, 63, 64]),                               # block edges, a context of one token
    (QWEN3_17B_3L, [5, 200, 65, 700]),                         # <= 4 rows: the launches fold their norms too
    (QWEN3_17B_3L, [300, 17, 128, 129, 1000]),                 # 5 rows: the launches run the norms on their own
    (QWEN3_17B_3L, [40 + 37 * i for i in range(8)]),
    (QWEN3_17B_3L, [7 + 61 * i for i in range(16)]),           # 16 rows: the most the kernel takes
    (QWEN2_LIKE, [100]),
    (QWEN2_LIKE, [3, 64, 500, 129, 77, 1, 255]),
])
def test_persistent_launch_is_bit_identical_to_the_launches(tmp_path, monkeypatch, cfg_dict, lens):
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.glue import qwen

    cfg, model, ex = _executor(tmp_path, cfg_dict)
    bm = BlockManager(cfg.num_kvcache_blocks, 64)
    seqs = _sequences(lens, cfg_dict["vocab_size"], 64, bm)
    monkeypatch.setattr(qwen, "PERSISTENT", False)
    _, toks = ex.execute_batch(seqs, prefill=True)
    calls = []
    real = qwen.hip_ops.decode_layers_persistent
    monkeypatch.setattr(qwen.hip_ops, "decode_layers_persistent", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for step in range(3):
        for s, t in zip(seqs, toks):
            s.append_token(int(t))
            bm.may_append(s)
        (la, ta, kva), (lb, tb, kvb) = _both_ways(ex, seqs, monkeypatch)
        assert len(calls) == step + 1, "the persistent kernel declined this shape"
        assert model.model.persistent_give_ups() == 0
        assert torch.equal(la, lb), f"step {step}: logits differ ({(la.float() - lb.float()).abs().max().item():.3e})"
        assert ta == tb
        assert torch.equal(kva, kvb), f"step {step}: K/V cache contents differ"
        toks = ta


def test_persistent_launch_soak_every_step_compared(tmp_path, monkeypatch):
    """300 consecutive decode steps (contexts growing across block edges, 3 rows), each run both ways on the same state and
    compared bit for bit — 300 x 17 grid barriers' worth of fence-free hand-offs with warm caches and whatever arrival skew
    the chip produces; the status word stays clean."""
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.glue import qwen

    cfg, model, ex = _executor(tmp_path, QWEN3_17B_3L)
    bm = BlockManager(cfg.num_kvcache_blocks, 64)
    seqs = _sequences([60, 1, 125], 4096, 64, bm)
    monkeypatch.setattr(qwen, "PERSISTENT", False)
    _, toks = ex.execute_batch(seqs, prefill=True)
    for step in range(300):
        for s, t in zip(seqs, toks):
            s.append_token(int(t))
            bm.may_append(s)
        monkeypatch.setattr(qwen, "PERSISTENT", False)
        la, ta = ex.execute_batch(seqs, prefill=False)
        la = la.clone()
        monkeypatch.setattr(qwen, "PERSISTENT", True)
        lb, tb = ex.execute_batch(seqs, prefill=False)
        assert torch.equal(la, lb) and ta == tb, f"step {step}"
        assert model.model.persistent_give_ups() == 0
        toks = ta
    assert not model.model._persist.get("declined")


def test_persistent_launch_without_prefetch_and_in_f16(tmp_path, monkeypatch):
    """The same bits with no weight chunk in flight across the barriers (flag 0xff: every chunk loaded behind the
    barrier), with one chunk, and in float16."""
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.glue import qwen

    for dtype, flags in ((torch.bfloat16, 0xff), (torch.bfloat16, 1), (torch.float16, 0)):
        cfg, model, ex = _executor(tmp_path, QWEN3_17B_3L, dtype=dtype)
        bm = BlockManager(cfg.num_kvcache_blocks, 64)
        seqs = _sequences([90, 513, 64], 4096, 64, bm)
        monkeypatch.setattr(qwen, "PERSISTENT", False)
        monkeypatch.setattr(qwen, "PERSISTENT_FLAGS", flags)
        _, toks = ex.execute_batch(seqs, prefill=True)
        for s, t in zip(seqs, toks):
            s.append_token(int(t))
            bm.may_append(s)
        (la, ta, kva), (lb, tb, kvb) = _both_ways(ex, seqs, monkeypatch)
        assert getattr(model.model, "_persist", None) is not None and not model.model._persist.get("declined")
        assert torch.equal(la, lb) and torch.equal(kva, kvb), (dtype, flags)


def test_persistent_launch_in_a_graph_and_through_the_engine(tmp_path, monkeypatch):
    """hipGraph capture (memset node + launch) replays to the eager bits, speculative replay included; LLM.generate gives
    the same greedy tokens with and without the persistent launch."""
    from minivllm_amd import LLM, Config, SamplingParams
    from minivllm_amd.glue import qwen

    d = tmp_path / "eng"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(QWEN3_17B_3L))
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(2, 4096, (n,), generator=g).tolist() for n in (30, 200, 65)]
    sp = SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True)
    results = {}
    for on in (False, True):
        for eager in (True, False):
            monkeypatch.setattr(qwen, "PERSISTENT", on)
            llm = LLM(Config(str(d), dtype="bfloat16", max_model_len=1024, max_num_batched_tokens=2048, max_num_seqs=4,
                             num_kvcache_blocks=64, enforce_eager=eager, load_format="dummy", seed=5))
            try:
                results[(on, eager)] = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
                inner = llm.model_runner.inference_executor.model.model
                assert (getattr(inner, "_persist", None) is not None) == on   # (off is the default: MV_PERSISTENT=1 selects it)
                assert inner.persistent_give_ups() == 0
                if not eager:
                    assert llm.model_runner.inference_executor.spec_hits > 0
            finally:
                llm.exit()
    assert results[(True, True)] == results[(False, True)] == results[(True, False)] == results[(False, False)]


def test_shapes_the_persistent_kernel_declines_fall_back(golden_dir, tmp_path, monkeypatch):
    """head_dim 64 (the tiny Qwen2 golden), 17 rows, 7 query heads per kv head: MV_E_SHAPE -> the launches run, once asked
    never asked again."""
    from minivllm_amd.glue import qwen
    from test_gpu_e2e import _build, _load, _run

    z = _load(golden_dir, "qwen2")
    monkeypatch.setattr(qwen, "PERSISTENT", True)
    _, model, ex = _build(tmp_path, z, torch.bfloat16)
    for mod in model.modules():
        if hasattr(mod, "pack_for_decode"):
            mod.pack_for_decode()
    on = _run(ex, z)
    assert getattr(model.model, "_persist", None) is None or model.model._persist.get("declined")
    monkeypatch.setattr(qwen, "PERSISTENT", False)
    off = _run(ex, z)
    for (a, ta), (b, tb) in zip(on, off):
        assert torch.equal(a, b) and ta == tb
