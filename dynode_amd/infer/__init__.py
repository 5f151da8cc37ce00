"""Inference-side glue of the simulate path (mirrors dynode.infer)."""

from . import distributions, handlers  # noqa: F401
from .sample import resolve_deterministic, sample_distributions, sample_then_resolve  # noqa: F401

__all__ = ["distributions", "handlers", "resolve_deterministic", "sample_distributions", "sample_then_resolve"]
