"""MI355X-native paged-attention + radix KV-cache hot path (drop-in for nanovllmx on that path)."""
from .core.common import SamplingParams  # noqa: F401

__all__ = ["SamplingParams", "LLM"]


def __getattr__(name):
    if name == "LLM":
        from .llm import LLM

        return LLM
    raise AttributeError(name)
