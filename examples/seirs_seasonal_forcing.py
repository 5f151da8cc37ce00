"""SEIRS with a sinusoidally forced transmission rate -- counterpart of the reference's
examples/seirs_seasonal_forcing.py, under the same module and function names
(``get_config``, ``get_seirs_odeparams``, ``seirs_ode_seasonal``) so that scripts and tests written
against the reference's example import unchanged.

beta(t) = beta * (1 + forcing_amp * sin(2 pi t / forcing_period + forcing_phase)); the three
forcing numbers travel in the parameter vector, so they can be batched or sampled like any other
parameter (BASELINE cfg 5 draws amplitude and phase per trajectory).
"""

from dynode_amd import SimulationConfig, simulate
from dynode_amd.rhs import SEIRS_Seasonal_ODEParams as SEIRS_ODEParams  # noqa: F401  (the reference's name here)
from dynode_amd.rhs import SeasonalityParams, seirs_ode_seasonal  # noqa: F401
from examples.seirs import get_config, get_seasonal_odeparams  # noqa: F401


def get_seirs_odeparams(config: SimulationConfig, forcing_amp=0.2, forcing_phase=0.0, forcing_period=365.0) -> SEIRS_ODEParams:
    return get_seasonal_odeparams(config, forcing_amp=forcing_amp, forcing_phase=forcing_phase,
                                  forcing_period=forcing_period)


if __name__ == "__main__":
    config = get_config()
    sol = simulate(ode=seirs_ode_seasonal, duration_days=1500, initial_state=config.initializer.get_initial_state(),
                   ode_parameters=get_seirs_odeparams(config, forcing_amp=0.2, forcing_phase=0.0, forcing_period=365.0),
                   solver_parameters=config.parameters.solver_params)
    s, e, i, r = [a.squeeze().cpu().numpy() for a in sol.ys]
    for day in (0, 365, 730, 1095, 1500):
        print(f"day {day:5d}  s {s[day]:.4f}  e {e[day]:.4f}  i {i[day]:.4f}  r {r[day]:.4f}")
    print("std over the last 100 days:", [float(x[-100:].std()) for x in (s, e, i, r)])
