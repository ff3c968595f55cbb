"""Checkpoint loading (mirror of minivllm/utils/loader.py:90-239): safetensors / .bin files of a
local directory, HF names remapped onto the fused parameters (q_proj|k_proj|v_proj -> qkv_proj,
gate_proj|up_proj -> gate_up_proj) and sliced per tensor-parallel rank by each parameter's
``weight_loader``.  Local paths only: there is no network on the benchmark machines."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any

import torch
from torch import nn

logger = logging.getLogger(__name__)

__all__ = ["load_weight", "load_model", "has_weight_files"]


def _default_loader(param: nn.Parameter, tensor: torch.Tensor, shard_id: Any = None) -> None:
    if param.data.shape == tensor.shape:
        param.data.copy_(tensor)
    else:
        logger.warning("Shape mismatch for parameter: expected %s, got %s. Skipping.",
                       tuple(param.data.shape), tuple(tensor.shape))


def load_weight(model: nn.Module, weight_name: str, tensor: torch.Tensor,
                packed_modules_mapping: dict[str, tuple[str, Any]]) -> bool:
    target, shard_id = weight_name, None
    for fragment, (fused, sid) in packed_modules_mapping.items():
        if fragment in weight_name:
            target, shard_id = weight_name.replace(fragment, fused), sid
            break
    try:
        param = model.get_parameter(target)
    except AttributeError:
        logger.warning("Parameter '%s' not found, skipping weight '%s'", target, weight_name)
        return False
    loader = getattr(param, "weight_loader", _default_loader)
    try:
        if shard_id is None:
            loader(param, tensor)
        else:
            loader(param, tensor, shard_id)
        return True
    except Exception as e:  # keep going like the reference does: a bad tensor must not abort the load
        logger.warning("Failed to load weight '%s': %s", weight_name, e)
        return False


def has_weight_files(model_path: str | Path) -> bool:
    p = Path(model_path)
    return p.is_dir() and (any(p.glob("*.safetensors")) or any(p.glob("*.bin")))


def load_model(model: nn.Module, model_path: str | Path) -> None:
    base = Path(model_path)
    if not base.is_dir():
        raise FileNotFoundError(f"Model weight directory not found: {base} (local directories only)")
    st_files, bin_files = sorted(base.glob("*.safetensors")), sorted(base.glob("*.bin"))
    if not st_files and not bin_files:
        raise ValueError(f"No .safetensors or .bin files found in {base}")
    mapping = getattr(model, "packed_modules_mapping", {})
    total = loaded = 0
    for path in st_files:
        from safetensors import safe_open

        with safe_open(str(path), framework="pt", device="cpu") as f:
            for name in f.keys():  # noqa: SIM118
                total += 1
                loaded += load_weight(model, name, f.get_tensor(name), mapping)
    for path in bin_files:
        state = torch.load(str(path), map_location="cpu", weights_only=True)
        for name, tensor in state.items():
            total += 1
            loaded += load_weight(model, name, tensor, mapping)
        del state
    if loaded < total:
        logger.warning("Weight loading completed with %d missing parameter(s) out of %d", total - loaded, total)
    logger.info("Loaded %d/%d weights from %s", loaded, total, base)
