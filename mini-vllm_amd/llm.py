"""Public entry point (mirror of minivllm/llm.py:51-106): ``LLM(Config(...)).generate(...)``."""
from .engine.llm_engine import LLMEngine

__all__ = ["LLM"]


class LLM(LLMEngine):
    """Same constructor contract as the reference: takes a ``Config`` (not keyword arguments)."""
