"""dynode_amd -- MI355X-native batched ODE engine behind DynODE's simulate/infer surface."""

from ._abi import ModelDesc  # noqa: F401

__version__ = "0.1.0"
