"""Inference side of the simulate path (the names ``dynode.infer`` exports, plus the handlers /
distributions that stand in for numpyro here)."""

import importlib as _importlib

from . import distributions, handlers  # noqa: F401

_PUBLIC = {
    "sample": ("sample_distributions", "resolve_deterministic", "sample_then_resolve"),
    "checkpointing": ("checkpoint_compartment_sizes",),
    "predictive": ("Predictive",),
    "inference": ("InferenceProcess", "MCMCProcess", "SVIProcess"),
    "ensemble": ("EnsembleSampler",),
}

__all__ = ["distributions", "handlers"]
for _module, _names in _PUBLIC.items():
    _loaded = _importlib.import_module(f"{__name__}.{_module}")
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
        __all__.append(_name)
del _module, _names, _loaded, _name
