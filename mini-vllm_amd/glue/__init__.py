from .registry import SUPPORTED_MODELS, create_model

__all__ = ["SUPPORTED_MODELS", "create_model"]
