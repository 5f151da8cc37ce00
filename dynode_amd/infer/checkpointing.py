"""Record compartment sizes at key time steps as deterministic sites (reference
src/dynode/infer/checkpointing.py:12-47).  Works for one trajectory or a batch: the time axis is
located from the right, so ``final_timestep_<c>`` has shape ``compartment.shape`` or
``(batch, *compartment.shape)``.
"""

from __future__ import annotations

import datetime

from ..utils import date_to_sim_day
from . import handlers


def checkpoint_compartment_sizes(config, solution, save_final_timesteps: bool = True,
                                 compartment_save_dates: list = ()):
    assert solution.ys is not None, "solution.ys returned None, odes failed."
    comps = config.idx.__dict__.items()

    def at(arr, comp_idx, day):
        time_axis = arr.dim() - 1 - len(config.compartments[comp_idx].shape)
        return arr.select(time_axis, day)

    if save_final_timesteps:
        for name, idx in comps:
            handlers.deterministic("final_timestep_%s" % name, at(solution.ys[idx], idx, -1))
    for date in compartment_save_dates:
        sim_day = (date_to_sim_day(date, config.initializer.initialize_date) if isinstance(date, datetime.date)
                   else int(date))
        n_days = solution.ts.shape[0]
        if 0 <= sim_day < n_days:
            tag = date.strftime("%Y_%m_%d") if isinstance(date, datetime.date) else f"day_{sim_day}"
            for name, idx in comps:
                handlers.deterministic(f"{tag}_timestep_{name}", at(solution.ys[idx], idx, sim_day))
