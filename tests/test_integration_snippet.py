"""INTEGRATION.md shows the binding a reference maintainer would add (a ctypes stub implementing the reference's
AttentionBackend on libmvhip.so).  This test EXECUTES that code block as written — with stand-ins for the two
reference modules it imports (the ABC and the context accessor; nothing of the reference travels to the GPU box) —
and checks store / prefill / paged decode through it against the oracle."""
import abc
import os
import re
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert len(blocks) == 1
    return blocks[0]


def _load(ctx_holder):
    """Execute the snippet in a fresh module namespace with stub `minivllm` modules on sys.modules."""
    class AttentionBackend(abc.ABC):      # the interface of minivllm/models/layers/attention_backend.py:81-128
        @abc.abstractmethod
        def forward(self, query, key, value, attention_mask=None, is_causal=True): ...

        @abc.abstractmethod
        def store_kv_cache(self, key, value, k_cache, v_cache, slot_mapping): ...

    stubs = {}
    for name in ("minivllm", "minivllm.models", "minivllm.models.layers", "minivllm.models.layers.attention_backend",
                 "minivllm.utils", "minivllm.utils.context"):
        stubs[name] = types.ModuleType(name)
    stubs["minivllm.models.layers.attention_backend"].AttentionBackend = AttentionBackend
    stubs["minivllm.utils.context"].get_context = lambda: ctx_holder["ctx"]
    from minivllm_amd import hip_ops

    code = _snippet().replace('"libmvhip.so"', repr(hip_ops.LIB_PATH))
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        ns = {"__name__": "mvhip_backend"}
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ns


def test_snippet_compiles_and_names_only_exported_symbols():
    from minivllm_amd import hip_ops

    code = _snippet()
    compile(code, "INTEGRATION.md", "exec")
    for sym in set(re.findall(r"_lib\.(mv_[a-z0-9_]+)", code)):
        assert sym in hip_ops.EXPORTED_SYMBOLS, sym


@pytest.mark.gpu
def test_snippet_runs_against_the_oracle():
    from oracle import ops as O

    dev = "cuda:0"
    holder = {}
    ns = _load(holder)
    backend = ns["MvHipAttentionBackend"]()
    torch.manual_seed(0)
    H, Hkv, D, bs = 16, 8, 128, 64
    lens = [70, 5]
    T = sum(lens)
    qkv = torch.randn(T, (H + 2 * Hkv) * D).bfloat16().to(dev)
    q = qkv[:, :H * D].view(T, H, D)                                   # views of the fused GEMM output, as in the model
    k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
    cu = torch.tensor([0, 70, 75], dtype=torch.int32, device=dev)
    holder["ctx"] = types.SimpleNamespace(cum_seqlens_q=cu, cum_seqlens_k=cu, max_seqlen_q=70, max_seqlen_k=70)
    # store into a paged cache through the snippet's backend
    table = torch.tensor([[2, 0], [1, -1]], dtype=torch.int32)
    slots = torch.tensor([int(table[i, p // bs]) * bs + p % bs for i, n in enumerate(lens) for p in range(n)], dtype=torch.int32)
    kc = torch.zeros(3, bs, Hkv, D, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    backend.store_kv_cache(k, v, kc, vc, slots.to(dev))
    kc_ref, vc_ref = torch.zeros(3, bs, Hkv, D, dtype=torch.bfloat16), torch.zeros(3, bs, Hkv, D, dtype=torch.bfloat16)
    O.store_kvcache(k.cpu(), v.cpu(), kc_ref, vc_ref, slots)
    assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref)
    # prefill
    out = backend.forward(q, k, v)
    ref = O.prefill_varlen(q.cpu(), k.cpu(), v.cpu(), cu.cpu(), cu.cpu(), D ** -0.5)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=2 ** -7, atol=2 ** -7)
    # decode over what was stored
    qd = torch.randn(2, H, D).bfloat16().to(dev)
    ctx_lens = torch.tensor(lens, dtype=torch.int64, device=dev)       # the reference's unit tests pass int64 here
    got = ns["mvhip_page_attention"](qd, kc, vc, table.to(dev).long(), ctx_lens, D ** -0.5)
    want = O.paged_decode(qd.cpu(), kc_ref, vc_ref, table, torch.tensor(lens, dtype=torch.int32), D ** -0.5)
    torch.testing.assert_close(got.float().cpu(), want, rtol=2 ** -7, atol=2 ** -7)
