"""Config / plugin schema of the simulate path (the names ``dynode.config`` exports).

Restated from the reference's behaviour (tests/test_config.py mirrors what its tests pin), not
copied: shapes and the recursive ``idx`` namespace, strain vectors, solver knobs, and the bin /
dimension builders the reference's larger models are described with.
"""

import importlib as _importlib

# module -> public names, in dependency order
_PUBLIC = {
    "bins": ("Bin", "DiscretizedPositiveIntBin", "AgeBin", "WaneBin"),
    "strains": ("Strain",),
    "dimension": ("Dimension", "VaccinationDimension", "ImmuneHistoryDimension",
                  "FullStratifiedImmuneHistoryDimension", "LastStrainImmuneHistoryDimension", "WaneDimension"),
    "deterministic_parameter": ("DeterministicParameter",),
    "placeholder_sample": ("PlaceholderSample", "SamplePlaceholderError"),
    "simulation_date": ("set_dynode_init_date_flag", "get_dynode_init_date_flag", "simulation_day"),
    "initializer": ("Initializer",),
    "params": ("Tsit5", "Dopri5", "SolverParams", "TransmissionParams", "Params"),
    "simulation_config": ("Compartment", "SimulationConfig"),
}

__all__ = []
for _module, _names in _PUBLIC.items():
    _loaded = _importlib.import_module(f"{__name__}.{_module}")
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
        __all__.append(_name)
del _module, _names, _loaded, _name
