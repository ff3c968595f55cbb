"""mini-vllm on MI355X: the paged-attention inference hot path of
jianzhnie/mini-vllm rebuilt for gfx950 (hand-written HIP kernels behind the
C-ABI in include/mvhip.h) under the reference's LLM / Config / SamplingParams
API.  Import as ``minivllm_amd`` (see minivllm_amd.py at the repo root)."""

__version__ = "0.1.0"

_LAZY = {
    "Config": ("config", "Config"),
    "SamplingParams": ("sampling_params", "SamplingParams"),
    "LLM": ("llm", "LLM"),
    "LLMEngine": ("engine.llm_engine", "LLMEngine"),
    "Sequence": ("engine.sequence", "Sequence"),
    "SequenceStatus": ("engine.sequence", "SequenceStatus"),
}

__all__ = list(_LAZY) + ["__version__"]


def __getattr__(name):
    # lazy so that `import minivllm_amd.hip_ops` does not drag in transformers
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
