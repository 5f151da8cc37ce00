"""Prior / posterior predictive: the model evaluated for a whole batch of parameter draws at once.

Counterpart of ``numpyro.infer.Predictive`` as the reference uses it
(src/dynode/infer/inference.py:225-237, examples/sir_infer_parameters.py:159-168).  numpyro maps
the model over the samples one by one; here every sample site receives a ``[num_samples]`` tensor,
so the ``simulate`` call inside the model becomes ONE batched launch (the "4096 batched parameter
samples" of BASELINE cfg 2).  The model must reduce with negative axes (see inference.py).
"""

from __future__ import annotations

from typing import Callable, Optional

import torch

from . import handlers


class Predictive:
    def __init__(self, model: Callable, posterior_samples: Optional[dict] = None, num_samples: Optional[int] = None,
                 exclude_deterministic: bool = True, return_observed: bool = False):
        if posterior_samples is None and num_samples is None:
            raise ValueError("either posterior_samples or num_samples must be given")
        self.model = model
        self.posterior_samples = posterior_samples
        if posterior_samples:
            sizes = {int(v.shape[0]) for v in posterior_samples.values()}
            if len(sizes) != 1:
                raise ValueError(f"posterior samples have inconsistent leading sizes {sizes}")
            num_samples = sizes.pop()
        self.num_samples = int(num_samples)
        self.exclude_deterministic = exclude_deterministic
        # numpyro's Predictive also reports observed sites (their value is the observation); off by
        # default here because callers normally pass ``obs_data=None`` to get predictive draws
        self.return_observed = return_observed

    def __call__(self, rng_key: int = 0, **model_kwargs) -> dict:
        """Run the model; unobserved sample sites not in ``posterior_samples`` are drawn from their
        priors / likelihoods (``obs_data=None`` turns the likelihood into a predictive draw)."""
        data = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in (self.posterior_samples or {}).items()}
        with torch.no_grad(), handlers.seed(rng_key, batch=self.num_samples), handlers.substitute(data), \
                handlers.trace() as tr:
            self.model(**model_kwargs)
        out = {}
        for name, site in tr.sites.items():
            if site["type"] == "deterministic" and self.exclude_deterministic:
                continue
            if site["type"] == "sample" and site["is_observed"] and not self.return_observed:
                continue
            if name in data:
                continue
            out[name] = site["value"]
        return out
