#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE (jianzhnie/mini-vllm) on the CPU.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
The reference imports ``colorama`` unconditionally for log colours and the image does
not ship it, so a no-op stand-in module is put on sys.path first (SURVEY.md §8c recipe);
nothing under /root/reference is modified and nothing from it is copied into the repo —
the fixtures are data only: seeded inputs and the outputs the reference produced.

    python tools/gen_goldens.py            # rewrites tests/golden/*.npz / *.json
    python tools/gen_goldens.py wide config2   # only the SURVEY §8(c) wide cases / the true-dimension config 2 run
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

REF = os.environ.get("MINIVLLM_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

os.environ["MINIVLLM_DEVICE"] = "cpu"
os.environ["CUDA_VISIBLE_DEVICES"] = ""

_blank = type("_Blank", (), {"__getattr__": lambda self, name: ""})()
_colorama = types.ModuleType("colorama")
_colorama.Fore = _colorama.Back = _colorama.Style = _blank
_colorama.init = lambda *a, **k: None
sys.modules.setdefault("colorama", _colorama)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(4)

import minivllm.engine.inference_executor as ref_exec_mod  # noqa: E402
from minivllm.config import Config  # noqa: E402
from minivllm.engine.block_manager import BlockManager  # noqa: E402
from minivllm.engine.inference_executor import InferenceExecutor  # noqa: E402
from minivllm.engine.scheduler import Scheduler  # noqa: E402
from minivllm.engine.sequence import Sequence  # noqa: E402
from minivllm.models.layers.activation import SiluAndMul  # noqa: E402
from minivllm.models.layers.attention import Attention  # noqa: E402
from minivllm.models.layers.attention_backend import StandardAttentionBackend  # noqa: E402
from minivllm.models.layers.attention_gather import BufferedPageAttention  # noqa: E402
from minivllm.models.layers.layernorm import RMSNorm  # noqa: E402
from minivllm.models.layers.page_attention import PageAttention  # noqa: E402
from minivllm.models.layers.rotary_embedding import RotaryEmbedding  # noqa: E402
from minivllm.models.registry import create_model  # noqa: E402
from minivllm.sampling import functional as SF  # noqa: E402
from minivllm.sampling_params import SamplingParams  # noqa: E402
from minivllm.utils.context import get_context, reset_context, set_context  # noqa: E402


def f32(t):
    return t.detach().float().cpu().numpy()


def i64(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.int64)


def randn(*shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


# ----------------------------------------------------------------------------- model dirs
TINY = {
    "qwen3": dict(model_type="qwen3", architectures=["Qwen3ForCausalLM"], hidden_size=256,
                  num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                  intermediate_size=512, vocab_size=320, max_position_embeddings=1024, rms_norm_eps=1e-6,
                  rope_theta=10000.0, tie_word_embeddings=True, hidden_act="silu", attention_bias=False,
                  torch_dtype="float32"),
    "qwen2": dict(model_type="qwen2", architectures=["Qwen2ForCausalLM"], hidden_size=448,
                  num_hidden_layers=2, num_attention_heads=7, num_key_value_heads=1,
                  intermediate_size=384, vocab_size=320, max_position_embeddings=1024, rms_norm_eps=1e-6,
                  rope_theta=1000000.0, tie_word_embeddings=False, hidden_act="silu",
                  torch_dtype="float32"),
    "opt": dict(model_type="opt", architectures=["OPTForCausalLM"], hidden_size=128, num_hidden_layers=2,
                num_attention_heads=2, ffn_dim=256, vocab_size=320, max_position_embeddings=1024,
                word_embed_proj_dim=128, do_layer_norm_before=True, enable_bias=True,
                tie_word_embeddings=True, activation_function="relu", torch_dtype="float32"),
}


def model_dir(tmp, name):
    d = os.path.join(tmp, name)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(TINY[name], f)
    return d


# ----------------------------------------------------------------------------- host traces
def fake_next_token(seq_tokens, vocab):
    """Deterministic stand-in for the model: depends only on the sequence content."""
    return (sum(seq_tokens) * 31 + len(seq_tokens) * 7 + 3) % vocab


def ctx_dump(input_ids, positions):
    c = get_context()
    d = dict(is_prefill=bool(c.is_prefill), max_seqlen_q=int(c.max_seqlen_q), max_seqlen_k=int(c.max_seqlen_k),
             input_ids=input_ids.tolist(), positions=positions.tolist())
    for name in ("cum_seqlens_q", "cum_seqlens_k", "slot_mapping", "context_lens", "block_tables"):
        t = getattr(c, name)
        d[name] = None if t is None else t.tolist()
        d[name + "_dtype"] = None if t is None else str(t.dtype).replace("torch.", "")
    d["input_ids_dtype"] = str(input_ids.dtype).replace("torch.", "")
    d["positions_dtype"] = str(positions.dtype).replace("torch.", "")
    return d


def run_scenario(cfg_dir, scenario):
    """Drive the reference Scheduler + BlockManager + InferenceExecutor input prep."""
    cfg = Config(cfg_dir, num_kvcache_blocks=scenario["num_blocks"], kvcache_block_size=scenario["block_size"],
                 max_model_len=scenario["max_model_len"], max_num_batched_tokens=scenario["max_num_batched_tokens"],
                 max_num_seqs=scenario["max_num_seqs"], enforce_eager=True, dtype="float32")
    cfg.eos = scenario.get("eos", -1)
    ex = InferenceExecutor(cfg, torch.nn.Linear(2, 2))
    sched = Scheduler(cfg)
    seqs = []
    for req in scenario["requests"]:
        sp = SamplingParams(temperature=req.get("temperature", 1.0), max_tokens=req["max_tokens"],
                            ignore_eos=req.get("ignore_eos", False))
        s = Sequence(req["tokens"], sp, block_size=scenario["block_size"])
        seqs.append(s)
    index_of = {id(s): i for i, s in enumerate(seqs)}
    add_at = scenario.get("add_at", [0] * len(seqs))
    steps = []
    step_no = 0
    pending = list(range(len(seqs)))
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
        batch, is_prefill = sched.schedule()
        if is_prefill:
            ids, pos = ex._prepare_prefill_input(batch)
        else:
            ids, pos = ex._prepare_decode_input(batch)
        rec = ctx_dump(ids, pos)
        reset_context()
        rec["seq_index"] = [index_of[id(s)] for s in batch]
        rec["block_table_per_seq"] = [list(s.block_table) for s in batch]
        rec["num_cached_tokens"] = [int(s.num_cached_tokens) for s in batch]
        toks = [fake_next_token(s.token_ids, scenario["vocab"]) for s in batch]
        rec["next_tokens"] = toks
        sched.postprocess(batch, toks)
        rec["finished"] = [bool(s.is_finished) for s in batch]
        rec["free_blocks_after"] = sched.block_manager.get_num_free_blocks()
        rec["waiting_after"] = [index_of[id(s)] for s in sched.waiting]
        rec["running_after"] = [index_of[id(s)] for s in sched.running]
        steps.append(rec)
        step_no += 1
        if step_no > scenario.get("max_steps", 400):
            raise RuntimeError("scenario did not terminate")
    final = [dict(tokens=list(s.token_ids), num_completion=int(s.num_completion_tokens)) for s in seqs]
    stats = dict(sched.block_manager.stats)
    return dict(scenario=scenario, steps=steps, final=final, stats=stats)


def preempt_scenario(name, seed, num_blocks, lens, max_tokens):
    g = torch.Generator().manual_seed(seed)
    reqs = [dict(tokens=torch.randint(1, 300, (n,), generator=g).tolist(), max_tokens=m, ignore_eos=True)
            for n, m in zip(lens, max_tokens)]
    return dict(name=name, num_blocks=num_blocks, block_size=64, max_model_len=512, max_num_batched_tokens=512,
                max_num_seqs=4, vocab=320, requests=reqs)


def host_traces(tmp):
    cfg_dir = model_dir(tmp, "qwen3")
    g = torch.Generator().manual_seed(1234)

    def rnd(n, hi=300):
        return torch.randint(1, hi, (n,), generator=g).tolist()

    shared = rnd(128)
    scenarios = [
        # the SURVEY §8 known-answer trace
        dict(name="survey_two_prompts", num_blocks=16, block_size=64, max_model_len=1024,
             max_num_batched_tokens=4096, max_num_seqs=8, vocab=320,
             requests=[dict(tokens=list(range(1, 131)), max_tokens=3, ignore_eos=True),
                       dict(tokens=list(range(500, 503)), max_tokens=3, ignore_eos=True)]),
        # prefix-cache hits: shared 128-token prefix, staggered arrival, boundary-length prompts
        dict(name="prefix_cache", num_blocks=24, block_size=64, max_model_len=1024,
             max_num_batched_tokens=4096, max_num_seqs=8, vocab=320, add_at=[0, 0, 3, 6],
             requests=[dict(tokens=shared + rnd(5), max_tokens=6, ignore_eos=True),
                       dict(tokens=shared + rnd(70), max_tokens=4, ignore_eos=True),
                       dict(tokens=shared[:64] + rnd(64), max_tokens=5, ignore_eos=True),
                       # (a fully cached prompt, len % 64 == 0 with every block hit, is NOT pinned: the
                       #  reference then prefills zero query tokens — SURVEY §9 Q3 — the build recomputes
                       #  the last block instead; covered by tests/test_host_logic.py)
                       dict(tokens=rnd(63), max_tokens=70, ignore_eos=True)]),      # crosses a block while decoding
        # memory pressure: preemption of the newest sequence, re-prefill, FIFO order.  (Parameters found
        # by search: under block pressure the reference often raises "No free blocks available"
        # because can_append tests len%bs==0 while may_append allocates at len%bs==1 — DESIGN.md Q10 —
        # so only scenarios the reference itself survives can be pinned.)
        preempt_scenario("preemption_a", 25, 4, [28, 48, 29], [48, 54, 65]),
        preempt_scenario("preemption_b", 106, 8, [108, 39, 121, 109], [67, 61, 33, 63]),
        # limits: max_num_seqs and max_num_batched_tokens cut the prefill batch; eos finishes early
        dict(name="limits_and_eos", num_blocks=64, block_size=64, max_model_len=256,
             max_num_batched_tokens=256, max_num_seqs=3, vocab=320, eos=17,
             requests=[dict(tokens=rnd(n), max_tokens=12, ignore_eos=(n % 2 == 0))
                       for n in (100, 90, 80, 1, 64, 65, 127, 128, 129, 7)]),
        # larger block size
        dict(name="block128", num_blocks=10, block_size=128, max_model_len=1024,
             max_num_batched_tokens=2048, max_num_seqs=8, vocab=320,
             requests=[dict(tokens=rnd(n), max_tokens=5, ignore_eos=True) for n in (127, 128, 129, 300)]),
    ]
    results = []
    for s in scenarios:
        try:
            results.append(run_scenario(cfg_dir, s))
        except Exception as e:  # a reference crash is itself a finding: report which scenario
            raise RuntimeError(f"reference failed on scenario {s['name']}: {e}") from e
    out = dict(scenarios=results)
    out["hashes"] = [
        dict(tokens=[1, 2, 3], prefix=-1, hash=BlockManager.compute_hash([1, 2, 3])),
        dict(tokens=[1, 2, 3], prefix=5, hash=BlockManager.compute_hash([1, 2, 3], 5)),
        dict(tokens=list(range(64)), prefix=-1, hash=BlockManager.compute_hash(list(range(64)))),
        dict(tokens=[151935, 0, 2**31 - 1], prefix=2**64 - 1,
             hash=BlockManager.compute_hash([151935, 0, 2**31 - 1], 2**64 - 1)),
    ]
    with open(os.path.join(OUT, "host_traces.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    n = sum(len(s["steps"]) for s in out["scenarios"])
    print(f"host_traces.json: {len(scenarios)} scenarios, {n} steps")


# ----------------------------------------------------------------------------- per-op goldens
def make_paged(B, Hkv, D, bs, ctxs, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    nblk = [(c + bs - 1) // bs for c in ctxs]
    total = sum(nblk) + 2
    perm = torch.randperm(total, generator=g).tolist()
    width = max(nblk) + 1
    table = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for j in range(nblk[b]):
            table[b, j] = next(it)
    kc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(total, bs, Hkv, D, generator=g).to(dtype)
    return kc, vc, table


def op_goldens():
    z = {}
    # --- K3: PageAttention and BufferedPageAttention
    cases = [(3, 4, 2, 64, 16, [1, 16, 40]), (2, 14, 2, 128, 64, [130, 64]), (2, 4, 4, 64, 4, [9, 3])]
    for ci, (B, H, Hkv, D, bs, ctxs) in enumerate(cases):
        kc, vc, table = make_paged(B, Hkv, D, bs, ctxs, seed=100 + ci)
        q = randn(B, H, D, seed=200 + ci)
        ctx = torch.tensor(ctxs, dtype=torch.int32)
        scale = D ** -0.5
        out = PageAttention()(q, kc, vc, table, ctx, scale)
        out_b = BufferedPageAttention()(q, kc, vc, table, ctx, scale)
        assert torch.allclose(out, out_b, atol=1e-6)
        p = f"decode{ci}_"
        z.update({p + "q": f32(q), p + "k_cache": f32(kc), p + "v_cache": f32(vc), p + "block_tables": i64(table),
                  p + "context_lens": i64(ctx), p + "scale": np.float32(scale), p + "out": f32(out)})
        # bf16 run of the reference itself (its own eager dtype behaviour)
        out16 = PageAttention()(q.bfloat16(), kc.bfloat16(), vc.bfloat16(), table, ctx, scale)
        z[p + "out_bf16"] = f32(out16)
    # --- K2: fallback prefill through Attention.forward (no KV cache -> no store)
    for ci, (H, Hkv, D, lens) in enumerate([(4, 2, 64, [5, 33, 1]), (14, 2, 128, [70, 9])]):
        T = sum(lens)
        q, k, v = randn(T, H, D, seed=300 + ci), randn(T, Hkv, D, seed=310 + ci), randn(T, Hkv, D, seed=320 + ci)
        cu = torch.tensor([0] + np.cumsum(lens).tolist(), dtype=torch.int32)
        attn = Attention(H, D, D ** -0.5, Hkv)
        set_context(True, max(lens), max(lens), cu, cu, None, None, None)
        out = attn(q, k, v)
        out16 = attn(q.bfloat16(), k.bfloat16(), v.bfloat16())
        reset_context()
        p = f"prefill{ci}_"
        z.update({p + "q": f32(q), p + "k": f32(k), p + "v": f32(v), p + "cu": i64(cu), p + "out": f32(out),
                  p + "out_bf16": f32(out16), p + "scale": np.float32(D ** -0.5)})
    # --- K1: store_kv_cache (torch scatter path on CPU), strided k/v views, -1 slots
    n, Hq, Hkv, D, nb, bs = 20, 4, 2, 64, 3, 16
    qkv = randn(n, (Hq + 2 * Hkv) * D, seed=400)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].view(n, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(n, Hkv, D)
    slots = torch.randperm(nb * bs, generator=torch.Generator().manual_seed(401))[:n].to(torch.int32)
    slots[2] = -1
    kc, vc = randn(nb, bs, Hkv, D, seed=402), randn(nb, bs, Hkv, D, seed=403)
    z.update({"store_qkv": f32(qkv), "store_slots": i64(slots), "store_kc_in": f32(kc), "store_vc_in": f32(vc)})
    StandardAttentionBackend().store_kv_cache(k, v, kc, vc, slots)
    z.update({"store_kc_out": f32(kc), "store_vc_out": f32(vc),
              "store_dims": np.array([n, Hq, Hkv, D, nb, bs], dtype=np.int64)})
    # --- K6: RMSNorm / add-RMSNorm, fp32 and bf16
    for cols in (128, 448):
        x, r = randn(6, cols, seed=500 + cols), randn(6, cols, seed=501 + cols)
        norm = RMSNorm(cols, eps=1e-6)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.1 * randn(cols, seed=502 + cols))
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            nd = RMSNorm(cols, eps=1e-6).to(dt)
            with torch.no_grad():
                nd.weight.copy_(norm.weight.to(dt))
            xd, rd = x.to(dt), r.to(dt)
            y, none = nd(xd)
            assert none is None
            y2, res = nd(xd, rd)
            p = f"rms{cols}_{tag}_"
            z.update({p + "x": f32(xd), p + "r": f32(rd), p + "w": f32(nd.weight), p + "y": f32(y),
                      p + "y_add": f32(y2), p + "res": f32(res)})
    # --- K7: rotary tables and application
    for ci, (D, base, scaling) in enumerate([(128, 1e6, None), (64, 1e4, None), (128, 1e4, {"type": "linear", "factor": 4.0}),
                                             (128, 1e4, {"type": "dynamic", "factor": 2.0})]):
        max_pos = 2048
        rope = RotaryEmbedding(D, D, max_pos, base, scaling)
        pos = torch.tensor([0, 1, 2, 63, 64, 1000, 2047, 5], dtype=torch.int64)
        q, k = randn(8, 3, D, seed=600 + ci), randn(8, 1, D, seed=610 + ci)
        qo, ko = rope(pos, q, k)
        qo16, ko16 = rope(pos, q.bfloat16(), k.bfloat16())
        p = f"rope{ci}_"
        rows = np.array([0, 1, 2, 63, 64, 1000, 2047], dtype=np.int64)
        z.update({p + "cos_rows": f32(rope.cos_cache[rows]), p + "sin_rows": f32(rope.sin_cache[rows]), p + "rows": rows,
                  p + "cfg": np.array([D, max_pos, base, 0 if not scaling else (1 if scaling["type"] == "linear" else 2),
                                       1.0 if not scaling else scaling["factor"]], dtype=np.float64),
                  p + "pos": i64(pos), p + "q": f32(q), p + "k": f32(k), p + "q_out": f32(qo), p + "k_out": f32(ko),
                  p + "q_out_bf16": f32(qo16), p + "k_out_bf16": f32(ko16)})
    # --- K9: SiluAndMul
    x = randn(5, 2 * 96, seed=700)
    z.update({"silu_x": f32(x), "silu_y": f32(SiluAndMul()(x)), "silu_y_bf16": f32(SiluAndMul()(x.bfloat16()))})
    # --- K8: sampler filter chain on fp32 logits (support sets are deterministic)
    B, V = 6, 200
    g = torch.Generator().manual_seed(800)
    logits = torch.stack([torch.randperm(V, generator=g).float() for _ in range(B)]) * 0.05 - 5.0
    temps = torch.tensor([1.0, 0.6, 0.6, 1.5, 0.8, 1.0])
    top_ks = torch.tensor([-1, 40, -1, 5, V + 3, 1], dtype=torch.int64)
    top_ps = torch.tensor([1.0, 0.95, 0.9, 1.0, 0.7, 1.0])
    min_ps = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.05, 0.0])
    s1 = SF.apply_temperature(logits, temps)
    s2 = SF.apply_top_k(s1, top_ks)
    s3 = SF.apply_top_p(s2, top_ps)
    s4 = SF.apply_min_p(s3, min_ps)
    z.update({"samp_logits": f32(logits), "samp_temps": f32(temps), "samp_top_ks": i64(top_ks), "samp_top_ps": f32(top_ps),
              "samp_min_ps": f32(min_ps), "samp_after_temp": f32(s1),
              "samp_keep_topk": np.isfinite(f32(s2)), "samp_keep_topp": np.isfinite(f32(s3)),
              "samp_keep_minp": np.isfinite(f32(s4))})
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **z)
    print(f"ops.npz: {len(z)} arrays")


# ----------------------------------------------------------------------------- end-to-end tiny models
def hf_style_state_dict(model, name):
    """Split the reference's packed parameters back into HF checkpoint names so that the
    build's own weight loader (packed-module remap) is what the e2e parity test exercises."""
    cfg = TINY[name]
    out = {}
    for k, v in model.state_dict().items():
        v = v.detach().float()
        if k.endswith("qkv_proj.weight") or k.endswith("qkv_proj.bias"):
            if name == "opt":
                hq = hkv = cfg["num_attention_heads"]
                hd = cfg["hidden_size"] // hq
            else:
                hq, hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
                hd = cfg.get("head_dim", cfg["hidden_size"] // hq)
            qs, ks = hq * hd, hkv * hd
            suffix = k.rsplit(".", 1)[1]
            base = k[: -len("qkv_proj." + suffix)]
            out[base + "q_proj." + suffix] = v[:qs]
            out[base + "k_proj." + suffix] = v[qs:qs + ks]
            out[base + "v_proj." + suffix] = v[qs + ks:]
        elif k.endswith("gate_up_proj.weight"):
            inter = cfg["intermediate_size"]
            base = k[: -len("gate_up_proj.weight")]
            out[base + "gate_proj.weight"] = v[:inter]
            out[base + "up_proj.weight"] = v[inter:]
        elif k == "lm_head.weight" and cfg.get("tie_word_embeddings"):
            continue
        else:
            out[k] = v
    return out


def e2e(tmp):
    prompts = [list(range(3, 3 + 70)), [9, 8, 7, 6, 5], list(range(100, 100 + 64)) + [1]]
    n_decode = 6
    for name in ("qwen3", "qwen2", "opt"):
        d = model_dir(tmp, name)
        z = {"prompt_lens": np.array([len(p) for p in prompts]), "prompts": np.array(sum(prompts, [])),
             "n_decode": np.int64(n_decode), "config_json": np.frombuffer(json.dumps(TINY[name]).encode(), dtype=np.uint8)}
        weights = None
        for tag, dt in (("f32", "float32"), ("bf16", "bfloat16")):
            cfg = Config(d, num_kvcache_blocks=16, kvcache_block_size=64, max_model_len=512,
                         max_num_batched_tokens=1024, max_num_seqs=8, enforce_eager=True, dtype=dt)
            torch.manual_seed(0)
            model = create_model(cfg.hf_config)
            g = torch.Generator().manual_seed(42)
            with torch.no_grad():
                for pname, p in model.named_parameters():
                    if "norm" in pname and pname.endswith("weight"):
                        p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
                    elif pname.endswith("bias"):
                        p.copy_(0.05 * torch.randn(p.shape, generator=g))
                    else:
                        p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5))
                    p.copy_(p.bfloat16().float())  # fp32 and bf16 runs share bf16-representable weights
            if weights is None:
                weights = hf_style_state_dict(model, name)
                for k, v in weights.items():  # bf16-representable by construction: store the 16 bits
                    z["w:" + k] = v.bfloat16().view(torch.int16).numpy().view(np.uint16)
            model = model.to(getattr(torch, dt))
            ex = InferenceExecutor(cfg, model)
            ex._allocate_kv_cache()
            ex._initialize_sampler()
            ex._optimize_model()
            bm = BlockManager(16, 64)
            seqs = [Sequence(p, SamplingParams(temperature=1.0, max_tokens=64, ignore_eos=True), block_size=64)
                    for p in prompts]
            for s in seqs:
                bm.allocate(s)
            with torch.inference_mode():
                logits, _ = ex.execute_batch(seqs, prefill=True)
            z[f"{tag}_prefill_logits"] = f32(logits)
            forced = []
            for step in range(n_decode):
                # teacher-forced continuation: the fp32 argmax, shared by both dtypes via z
                if tag == "f32":
                    toks = torch.argmax(logits.float(), dim=-1).tolist()
                    forced.append(toks)
                else:
                    toks = z["forced_tokens"][step].tolist()
                for s, t in zip(seqs, toks):
                    s.append_token(int(t))
                    bm.may_append(s)
                with torch.inference_mode():
                    logits, _ = ex.execute_batch(seqs, prefill=False)
                z[f"{tag}_decode_logits_{step}"] = f32(logits)
            if tag == "f32":
                z["forced_tokens"] = np.array(forced, dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, f"e2e_{name}.npz"), **z)
        print(f"e2e_{name}.npz: {len(z)} arrays")


# ----------------------------------------------------------------------------- the wide cases
def wide_goldens():
    """Decode at the head layouts 16/8, 28/4, 8/1, 12/12 with contexts up to 4096, and prefill up to 1024 tokens per
    sequence (the same outputs pin the prefix-cached prefill: its rows are the new-token rows of the full no-cache
    prefill).  Inputs are regenerated from their seeds on the test side (tests/golden_inputs.py); stored: the
    reference's fp32 outputs and a checksum of the inputs it saw."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_inputs as GI

    z = {}
    for case in GI.WIDE_DECODE:
        q, kc, vc, table, ctx = GI.wide_decode_inputs(case)
        scale = case["D"] ** -0.5
        out = PageAttention()(q.float(), kc.float(), vc.float(), table, ctx, scale)
        z[f"decode_{case['name']}_out"] = f32(out)
        z[f"decode_{case['name']}_checksum"] = np.float64(GI.checksum(q, kc, vc, table, ctx))
    for case in GI.WIDE_PREFILL:
        q, k, v, cu = GI.wide_prefill_inputs(case)
        attn = Attention(case["H"], case["D"], case["D"] ** -0.5, case["Hkv"])
        set_context(True, max(case["lens"]), max(case["lens"]), cu, cu, None, None, None)
        out = attn(q.float(), k.float(), v.float())
        reset_context()
        z[f"prefill_{case['name']}_out_rows"] = f32(out.reshape(q.shape)[GI.wide_prefill_rows(case)])
        z[f"prefill_{case['name']}_checksum"] = np.float64(GI.checksum(q, k, v, cu))
    np.savez_compressed(os.path.join(OUT, "ops_wide.npz"), **z)
    print(f"ops_wide.npz: {len(z)} arrays")


def config2_golden(tmp):
    """BASELINE config 2 at its true dimensions: Qwen3-0.6B, all 28 layers, seeded bf16-representable weights
    (tests/golden_inputs.py::seeded_hf_weights), prefill of two prompts + 3 teacher-forced decode steps through the
    reference's InferenceExecutor in fp32 and in bf16.  Stored per step: the fp32 logits at 512 fixed vocabulary
    columns and their top-8, the logit scale, and the reference's OWN bf16-vs-fp32 error (the noise floor the
    bf16 parity bound is taken from)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_inputs as GI

    d = os.path.join(tmp, "config2")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(GI.CONFIG2, f)
    hfw = GI.seeded_hf_weights(GI.CONFIG2)
    prompts, cols = GI.config2_prompts(), GI.config2_columns()
    z = {"weights_checksum": np.float64(GI.checksum(hfw["model.embed_tokens.weight"][:64],
                                                    hfw["model.layers.27.mlp.down_proj.weight"][:8],
                                                    hfw["model.layers.0.self_attn.q_norm.weight"])),
         "columns": cols.numpy()}
    full = {}
    for tag, dt in (("f32", "float32"), ("bf16", "bfloat16")):
        cfg = Config(d, num_kvcache_blocks=8, kvcache_block_size=64, max_model_len=512, max_num_batched_tokens=1024,
                     max_num_seqs=4, enforce_eager=True, dtype=dt)
        model = create_model(cfg.hf_config)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if name.endswith("qkv_proj.weight"):
                    b = name[: -len("qkv_proj.weight")]
                    src = torch.cat([hfw[b + "q_proj.weight"], hfw[b + "k_proj.weight"], hfw[b + "v_proj.weight"]])
                elif name.endswith("gate_up_proj.weight"):
                    b = name[: -len("gate_up_proj.weight")]
                    src = torch.cat([hfw[b + "gate_proj.weight"], hfw[b + "up_proj.weight"]])
                elif name == "lm_head.weight":
                    src = hfw["model.embed_tokens.weight"]
                else:
                    src = hfw[name]
                p.copy_(src.float())
        model = model.to(getattr(torch, dt))
        ex = InferenceExecutor(cfg, model)
        ex._allocate_kv_cache(); ex._initialize_sampler(); ex._optimize_model()
        bm = BlockManager(8, 64)
        seqs = [Sequence(p, SamplingParams(temperature=1.0, max_tokens=64, ignore_eos=True), block_size=64) for p in prompts]
        for s in seqs:
            bm.allocate(s)
        forced = []
        for step in range(GI.CONFIG2_DECODE_STEPS + 1):
            with torch.inference_mode():
                logits, _ = ex.execute_batch(seqs, prefill=(step == 0))
            logits = logits.float()
            if tag == "f32":
                full[step] = logits.clone()
                top = logits.topk(8, dim=-1)
                z[f"cols_{step}"] = f32(logits[:, cols])
                z[f"top_idx_{step}"] = top.indices.numpy().astype(np.int64)
                z[f"top_val_{step}"] = f32(top.values)
                z[f"scale_{step}"] = np.float32(logits.abs().max())
                toks = logits.argmax(-1).tolist()
                forced.append(toks)
            else:
                z[f"ref_bf16_noise_{step}"] = np.float32((logits - full[step]).abs().max())
                toks = z["forced_tokens"][step].tolist()
            for s, t in zip(seqs, toks):
                s.append_token(int(t))
                bm.may_append(s)
        if tag == "f32":
            z["forced_tokens"] = np.array(forced, dtype=np.int64)
        del model, ex
    np.savez_compressed(os.path.join(OUT, "e2e_config2_qwen3_0p6b.npz"), **z)
    print("e2e_config2_qwen3_0p6b.npz:", {k: (float(v) if v.ndim == 0 else v.shape) for k, v in z.items() if "noise" in k or "scale" in k})


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        if not only or "base" in only:
            host_traces(tmp)
            op_goldens()
            e2e(tmp)
        if not only or "wide" in only:
            wide_goldens()
        if not only or "config2" in only:
            config2_golden(tmp)


if __name__ == "__main__":
    main()
