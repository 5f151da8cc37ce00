"""Inference-side glue of the simulate path (mirrors dynode.infer)."""

from . import distributions, handlers  # noqa: F401
from .checkpointing import checkpoint_compartment_sizes  # noqa: F401
from .predictive import Predictive  # noqa: F401
from .sample import resolve_deterministic, sample_distributions, sample_then_resolve  # noqa: F401

__all__ = ["Predictive", "checkpoint_compartment_sizes", "distributions", "handlers", "resolve_deterministic", "sample_distributions", "sample_then_resolve"]
