"""Model + tokenizer lifecycle (mirror of minivllm/models/manager.py:31-242) for one GPU of the
tensor-parallel group.  Parameters are created directly on the device in the serving dtype
(a 72B shard never exists in fp32 on the host), then filled from the checkpoint or — with
``Config.load_format == "dummy"`` — random-initialised at the true architecture dimensions."""
from __future__ import annotations

import logging
import os

import torch

from ..config import Config
from ..glue import create_model
from ..utils.loader import has_weight_files, load_model

logger = logging.getLogger(__name__)
__all__ = ["ModelManager", "resolve_dtype"]

_DTYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}


def resolve_dtype(config: Config) -> torch.dtype:
    if config.dtype != "auto":
        return _DTYPES[config.dtype]
    # the reference leaves "auto" at fp32 whatever the checkpoint says; here it
    # means what it says: the checkpoint's own dtype
    hf = getattr(config.hf_config, "torch_dtype", None) or getattr(config.hf_config, "dtype", None)
    if isinstance(hf, str):
        hf = _DTYPES.get(hf.replace("torch.", ""))
    return hf if hf in _DTYPES.values() else torch.float32


class ModelManager:
    def __init__(self, config: Config) -> None:
        self.config = config
        self.device: torch.device | None = None
        self.model = None
        self.tokenizer = None
        self.model_type: str | None = None

    def initialize(self) -> None:
        self._setup_device()
        if not self.config.model:
            raise ValueError("Model path cannot be empty")
        self._load_tokenizer()
        self._load_model()
        self._validate()

    def _setup_device(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("minivllm_amd needs a ROCm GPU (MI355X): the inference hot path runs on "
                               "hand-written HIP kernels and has no CPU fallback")
        # one process per GPU; modulo only matters when ranks outnumber GPUs (single-GPU TP tests)
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.device)

    def _load_tokenizer(self) -> None:
        from transformers import AutoTokenizer

        try:
            # a directory without vocabulary files still yields a tokenizer OBJECT (one special token, every
            # string encodes to []): insist on the files
            files = ("tokenizer.json", "tokenizer.model", "vocab.json", "tokenizer_config.json")
            if os.path.isdir(self.config.model) and not any(
                    os.path.exists(os.path.join(self.config.model, f)) for f in files):
                raise FileNotFoundError(f"no tokenizer files in {self.config.model}")
            # custom tokenizer code in the model directory runs only if the user asked for it (the reference
            # hard-codes True, models/manager.py:99)
            tok = AutoTokenizer.from_pretrained(self.config.model,
                                                trust_remote_code=bool(self.config.trust_remote_code),
                                                use_fast=True, local_files_only=True)
            if tok.pad_token is None:
                tok.pad_token = tok.eos_token
            self.tokenizer = tok
        except Exception as e:
            # the reference refuses to start without a tokenizer (models/manager.py:111-117); here a model
            # directory without one still serves token-id prompts (string prompts then raise in add_request)
            logger.warning("no usable tokenizer in %s (%s): prompts must be token-id lists", self.config.model, e)
            self.tokenizer = None

    def _load_model(self) -> None:
        cfg = self.config
        dtype = resolve_dtype(cfg)
        prev = torch.get_default_dtype()
        try:
            torch.set_default_dtype(dtype)
            with torch.device(self.device):
                model = create_model(cfg.hf_config)
        finally:
            torch.set_default_dtype(prev)
        self.model_type = type(model).__name__.replace("ForCausalLM", "").lower()
        if cfg.load_format == "dummy" or (cfg.load_format == "auto" and not has_weight_files(cfg.model)
                                          and os.environ.get("MINIVLLM_DUMMY_WEIGHTS") == "1"):
            self._dummy_init(model, cfg.seed if cfg.seed is not None else 0)
        else:
            load_model(model, cfg.model)
        # params already have `dtype`; this also casts the RoPE tables like the reference's
        # model.to(device, dtype) does (its bf16 runs rotate with bf16-rounded cos/sin)
        self.model = model.to(device=self.device, dtype=dtype)
        # decode steps stream a fragment-ordered copy of every projection weight and of the LM head
        # (layers/linear.py); MV_PACK_WEIGHTS=0 keeps only the row-major weights (library GEMMs above 16-32 rows)
        if os.environ.get("MV_PACK_WEIGHTS", "1") != "0":
            mods = [m for m in self.model.modules() if hasattr(m, "pack_for_decode")]
            # the copies double the projection weights in HBM: only while they stay a modest share of the device
            # (Qwen2-72B on ONE GPU: 135 GiB of weights — a second copy would leave no room for the KV pool; it then
            # decodes on the row-major weights, like MV_PACK_WEIGHTS=0)
            extra = sum(m.weight.numel() * m.weight.element_size() for m in mods)
            total = torch.cuda.get_device_properties(self.device).total_memory if self.device.type == "cuda" else 0
            if total and extra > 0.25 * total:
                logger.warning("not packing the projection weights for decode: %.1f GiB of copies on a %.0f GiB device",
                               extra / 2 ** 30, total / 2 ** 30)
            else:
                for mod in mods:
                    mod.pack_for_decode()

    @staticmethod
    def _dummy_init(model, seed: int) -> None:
        gen = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                elif name.endswith("bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, 0.02, generator=gen)

    def _validate(self) -> None:
        c = self.model.config
        for field in ("vocab_size", "hidden_size", "num_attention_heads"):
            if getattr(c, field, 0) <= 0:
                raise ValueError(f"Invalid {field}: {getattr(c, field, 0)}")

    def get_model_info(self) -> dict:
        if self.model is None:
            return {}
        c = self.model.config
        return {"model_type": self.model_type, "device": str(self.device), "vocab_size": c.vocab_size,
                "hidden_size": c.hidden_size, "num_layers": c.num_hidden_layers,
                "num_heads": c.num_attention_heads, "dtype": str(self.config.dtype),
                "tensor_parallel_size": self.config.tensor_parallel_size}

    def cleanup(self) -> None:
        self.model = None
        self.tokenizer = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
