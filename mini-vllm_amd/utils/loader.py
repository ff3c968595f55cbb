"""Checkpoint loading (mirror of minivllm/utils/loader.py:90-239): safetensors / .bin files of a
local directory, HF names remapped onto the fused parameters (q_proj|k_proj|v_proj -> qkv_proj,
gate_proj|up_proj -> gate_up_proj) and sliced per tensor-parallel rank by each parameter's
``weight_loader``.  Local paths only: there is no network on the benchmark machines.

Sharded, direct-to-device (the reference reads every tensor whole onto the CPU and slices it
there): a safetensors tensor reaches the ``weight_loader`` as a ``LazyTensor`` — shape and dtype known, nothing
read yet.  ``narrow`` / ``chunk`` (the only things the loaders do before ``copy_``) turn into a slice of the file,
read straight onto the parameter's device: at TP = 8 a rank reads 1/8 of every sharded matrix (18 of 145 GB for
Qwen2-72B) and nothing is staged in host memory."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any

import torch
from torch import nn

logger = logging.getLogger(__name__)

__all__ = ["load_weight", "load_model", "has_weight_files", "LazyTensor", "materialize"]


class LazyTensor:
    """A (slice of a) checkpoint tensor that is read only when it is materialised.  Supports what the
    ``weight_loader`` hooks call on a loaded weight: ``shape`` / ``dim()`` / ``size()``, ``narrow``, ``chunk``,
    indexing a chunk list, and being the source of ``param.data.copy_`` (through ``materialize``)."""
    bytes_read = 0     # class-wide counter (tests: a rank must not read the other ranks' shards)

    def __init__(self, st_slice, device, ranges=None, dtype=None):
        self._sl, self._device = st_slice, device
        full = tuple(st_slice.get_shape())
        self._ranges = list(ranges) if ranges is not None else [(0, n) for n in full]
        self._dtype = dtype

    @property
    def shape(self):
        return torch.Size(hi - lo for lo, hi in self._ranges)

    def dim(self) -> int:
        return len(self._ranges)

    def size(self, d: int | None = None):
        return self.shape if d is None else self.shape[d]

    def narrow(self, dim: int, start: int, length: int) -> "LazyTensor":
        lo, hi = self._ranges[dim]
        if start < 0 or lo + start + length > hi:
            raise IndexError(f"narrow({dim}, {start}, {length}) outside a dimension of size {hi - lo}")
        r = list(self._ranges)
        r[dim] = (lo + start, lo + start + length)
        return LazyTensor(self._sl, self._device, r, self._dtype)

    def chunk(self, n: int, dim: int = 0) -> list["LazyTensor"]:
        size = self.shape[dim]
        step = -(-size // n)
        return [self.narrow(dim, i, min(step, size - i)) for i in range(0, size, step)]

    def materialize(self) -> torch.Tensor:
        idx = tuple(slice(lo, hi) for lo, hi in self._ranges)
        t = self._sl[idx] if idx else self._sl[...]
        LazyTensor.bytes_read += t.numel() * t.element_size()
        return t if t.device == torch.device(self._device) else t.to(self._device)


def materialize(t):
    """A real tensor from what a weight_loader was handed (LazyTensor or tensor)."""
    return t.materialize() if isinstance(t, LazyTensor) else t


def _default_loader(param: nn.Parameter, tensor, shard_id: Any = None) -> None:
    if param.data.shape == tensor.shape:
        param.data.copy_(materialize(tensor))
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

        # tensors go straight to the device the parameters live on, and only the slice a rank owns is read
        try:
            dev = next(model.parameters()).device
        except StopIteration:
            dev = torch.device("cpu")
        dev_s = "cpu" if dev.type == "cpu" else f"{dev.type}:{dev.index if dev.index is not None else 0}"
        with safe_open(str(path), framework="pt", device=dev_s) as f:
            for name in f.keys():  # noqa: SIM118
                total += 1
                loaded += load_weight(model, name, LazyTensor(f.get_slice(name), dev), mapping)
    for path in bin_files:
        state = torch.load(str(path), map_location="cpu", weights_only=True)
        for name, tensor in state.items():
            total += 1
            loaded += load_weight(model, name, tensor, mapping)
        del state
    if loaded < total:
        logger.warning("Weight loading completed with %d missing parameter(s) out of %d", total - loaded, total)
    logger.info("Loaded %d/%d weights from %s", loaded, total, base)
