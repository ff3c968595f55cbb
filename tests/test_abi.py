"""The C-ABI library loads (no GPU needed) and exports exactly what include/mvhip.h declares; the
product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mvhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from minivllm_amd import hip_ops

    names = _declared()
    assert len(names) >= 12
    lib = ctypes.CDLL(hip_ops.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mvhip.h but not exported"
    assert sorted(hip_ops.EXPORTED_SYMBOLS) == names      # the ctypes table covers the whole header
    assert hip_ops.version() == 1
    assert hip_ops.lib().mv_strerror(-5).decode().startswith("pointer or stride")
    assert b"ok" == hip_ops.lib().mv_strerror(0)


def test_workspace_size_queries_are_host_only():
    from minivllm_amd import hip_ops

    assert hip_ops.paged_decode_workspace_bytes(256, 16, 8, 128, 64, 64) == 256      # one split: no workspace
    small = hip_ops.paged_decode_workspace_bytes(1, 16, 8, 128, 64, 64)
    assert small > 256 and small % 256 == 0                                          # split-KV partials
    assert hip_ops.paged_decode_workspace_bytes(1, 16, 8, 96, 64, 64) == 256         # unsupported D: nothing


def test_struct_layouts_of_the_binding_match_the_library():
    """mv_decode_layers_persistent takes two structs by pointer: the ctypes declarations must have the library's layout."""
    from minivllm_amd import hip_ops

    assert hip_ops.lib().mv_persist_struct_bytes(1) == ctypes.sizeof(hip_ops.MvPersistArgs)
    assert hip_ops.lib().mv_persist_struct_bytes(0) == 8 * len(hip_ops.PERSIST_LAYER_FIELDS)   # a row of the int64 layer table
    assert hip_ops.lib().mv_decode_layers_persistent_workspace_bytes(1, 2048, 16, 8, 128, 6144, 64, 64) > 4096


def test_no_cpu_fallback():
    from minivllm_amd import hip_ops

    x = torch.zeros(4, 128, dtype=torch.bfloat16)
    with pytest.raises(hip_ops.MvHipError, match="no CPU fallback"):
        hip_ops.rmsnorm(x, torch.ones(128, dtype=torch.bfloat16), 1e-6)
    with pytest.raises(hip_ops.MvHipError):
        hip_ops.silu_and_mul(x)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mini-vllm_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"
