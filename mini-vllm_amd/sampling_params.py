"""Per-request sampling knobs (mirror of minivllm/sampling_params.py:12-67)."""
from dataclasses import dataclass

__all__ = ["SamplingParams"]


@dataclass
class SamplingParams:
    temperature: float = 1.0   # 0 selects greedy decoding (argmax)
    top_p: float = 1.0         # nucleus mass, (0, 1]
    top_k: int = -1            # -1 disables
    min_p: float = 0.0         # relative-probability floor, [0, 1]
    max_tokens: int = 64
    ignore_eos: bool = False

    def __post_init__(self) -> None:
        checks = (
            (self.temperature >= 0, f"temperature must be >= 0, got {self.temperature}"),
            (0.0 < self.top_p <= 1.0, f"top_p must be in (0, 1], got {self.top_p}"),
            (self.top_k == -1 or self.top_k > 0, f"top_k must be -1 (disable) or > 0, got {self.top_k}."),
            (0.0 <= self.min_p <= 1.0, f"min_p must be in [0, 1], got {self.min_p}"),
            (self.max_tokens > 0, f"max_tokens must be > 0, got {self.max_tokens}"),
        )
        for ok, msg in checks:
            if not ok:
                raise ValueError(msg)
