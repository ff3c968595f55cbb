"""Pins oracle/model.py (the CPU restatement of the reference's eager model path) to logits
recorded from the reference itself, and runs the HOST engine end to end on the CPU through it
(BASELINE config 1: the reference's examples/cpu_inference_opt.py plumbing run)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.model import OracleModel, OracleRunner


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"e2e_{name}.npz"))
    cfg = json.loads(bytes(z["config_json"]).decode())
    weights = {k[2:]: torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
               for k in z.files if k.startswith("w:")}
    lens, flat = z["prompt_lens"].tolist(), z["prompts"].tolist()
    prompts, o = [], 0
    for n in lens:
        prompts.append(flat[o:o + n])
        o += n
    return z, cfg, weights, prompts


def _drive(model, prompts, forced):
    """Same teacher-forced schedule as tools/gen_goldens.py, on the build's host logic."""
    from minivllm_amd.engine.block_manager import BlockManager
    from minivllm_amd.engine.input_builder import InputBuilder
    from minivllm_amd.engine.sequence import Sequence

    bm, ib = BlockManager(16, 64), InputBuilder(64, 8, 512, 1024)
    seqs = [Sequence(p, block_size=64) for p in prompts]
    for s in seqs:
        bm.allocate(s)
    inp = ib.prefill(seqs)
    meta = dict(is_prefill=True, slot_mapping=inp.slot_mapping, block_tables=inp.block_tables,
                cu_q=inp.cum_seqlens_q, cu_k=inp.cum_seqlens_k)
    outs = [model.forward_logits(inp.input_ids.clone(), inp.positions.clone(), meta).float()]
    for toks in forced:
        for s, t in zip(seqs, toks):
            s.append_token(int(t))
            bm.may_append(s)
        inp = ib.decode(seqs)
        meta = dict(is_prefill=False, slot_mapping=inp.slot_mapping, block_tables=inp.block_tables,
                    context_lens=inp.context_lens)
        outs.append(model.forward_logits(inp.input_ids.clone(), inp.positions.clone(), meta).float())
    return outs


@pytest.mark.parametrize("name", ["qwen3", "qwen2", "opt"])
def test_oracle_model_fp32_matches_reference(golden_dir, name):
    z, cfg, weights, prompts = _load(golden_dir, name)
    outs = _drive(OracleModel(cfg, weights, torch.float32), prompts, z["forced_tokens"].tolist())
    refs = [z["f32_prefill_logits"]] + [z[f"f32_decode_logits_{i}"] for i in range(int(z["n_decode"]))]
    for i, (got, want) in enumerate(zip(outs, refs)):
        want = torch.from_numpy(want)
        # identical op order on the same CPU; attention is fp32 in both
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5 * want.abs().max().item(), msg=f"{name} step {i}")


@pytest.mark.parametrize("name", ["qwen3", "qwen2", "opt"])
def test_oracle_model_bf16_within_reference_noise(golden_dir, name):
    z, cfg, weights, prompts = _load(golden_dir, name)
    outs = _drive(OracleModel(cfg, weights, torch.bfloat16), prompts, z["forced_tokens"].tolist())
    for i, got in enumerate(outs):
        key = "prefill_logits" if i == 0 else f"decode_logits_{i - 1}"
        w32, w16 = torch.from_numpy(z["f32_" + key]), torch.from_numpy(z["bf16_" + key])
        noise = (w16 - w32).abs().max().item()
        assert (got - w32).abs().max().item() <= max(1e-3 * w32.abs().max().item(), 1.5 * noise), f"{name} {i}"


def test_cpu_plumbing_config1_opt_through_host_engine(golden_dir, tmp_path):
    """BASELINE config 1: OPT, fp32, max_num_seqs=8, max_model_len=1024, enforce_eager, 5 prompts,
    T=0.6 / top_p=0.95 / top_k=40, 50 new tokens (examples/cpu_inference_opt.py:143-158) — the
    build's LLMEngine + Scheduler + InputBuilder on the CPU with the oracle as the runner."""
    from minivllm_amd import Config, LLMEngine, SamplingParams
    from minivllm_amd.engine.input_builder import InputBuilder

    z, cfg_d, weights, _ = _load(golden_dir, "opt")
    d = tmp_path / "opt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg_d))
    cfg = Config(str(d), max_num_seqs=8, max_model_len=1024, enforce_eager=True, dtype="float32",
                 num_kvcache_blocks=32)
    model = OracleModel(cfg_d, weights, torch.float32, num_blocks=32)
    runner = OracleRunner(model, InputBuilder(64, 8, 1024, cfg.max_num_batched_tokens), seed=0)
    engine = LLMEngine(cfg, model_runner=runner)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(3, 300, (16,), generator=g).tolist() for _ in range(5)]
    sp = SamplingParams(temperature=0.6, top_p=0.95, top_k=40, max_tokens=50, ignore_eos=True)
    outs = engine.generate(prompts, sp, use_tqdm=False)
    assert len(outs) == 5 and all(len(o["token_ids"]) == 50 for o in outs)
    assert all(0 <= t < cfg_d["vocab_size"] for o in outs for t in o["token_ids"])
    assert engine.scheduler.block_manager.get_num_free_blocks() == 32     # everything released
    # greedy through the same plumbing is deterministic and reproducible
    greedy = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    a = engine.generate(prompts[:2], greedy, use_tqdm=False)
    b = engine.generate(prompts[:2], greedy, use_tqdm=False)
    assert [o["token_ids"] for o in a] == [o["token_ids"] for o in b]
    with pytest.raises(TypeError):
        LLMEngine("not a config")
