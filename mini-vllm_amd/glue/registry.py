"""Architecture registry (mirror of minivllm/models/registry.py:30-52).  GPT-2 is not built:
it is in none of the benchmark configurations."""
from typing import Any

from .opt import OPTForCausalLM
from .qwen import Qwen2ForCausalLM, Qwen3ForCausalLM

SUPPORTED_MODELS = {"Qwen2ForCausalLM": Qwen2ForCausalLM, "Qwen3ForCausalLM": Qwen3ForCausalLM,
                    "OPTForCausalLM": OPTForCausalLM}
TYPE_TO_ARCH = {"qwen2": "Qwen2ForCausalLM", "qwen3": "Qwen3ForCausalLM", "opt": "OPTForCausalLM"}


def create_model(hf_config: Any) -> Any:
    architectures = getattr(hf_config, "architectures", None) or []
    model_type = getattr(hf_config, "model_type", "").lower()
    name = next((a for a in architectures if a in SUPPORTED_MODELS), None) or TYPE_TO_ARCH.get(model_type)
    if name is None:
        raise ValueError(f"Unsupported model: model_type={model_type!r}, architectures={architectures}")
    return SUPPORTED_MODELS[name](hf_config)
