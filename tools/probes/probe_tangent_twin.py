"""dev probe: a gradient-solve no lean instance applies to (the 2-age x 3-strain model scored on the VALUES of c), general
tangent instance against its static / adaptive-only twin (FEAT bits 10 + 11): HIP-event us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynode_amd import PoissonObservation, _abi, engine, simulate
from dynode_amd.infer import folded, handlers, sample_then_resolve
from dynode_amd.infer.inference import Potential, init_to_median
from dynode_amd.rhs import seirs_multi_strain_ode
from examples import infer_multi_strain as ex_m

cfg = ex_m.base.get_config(**ex_m.TRUTH)
values = ex_m._solve(cfg, 120).ys[cfg.idx.c].cpu()


def model(config, tf, obs_data):
    config = config.model_copy(deep=False)
    config.parameters = config.parameters.model_copy(deep=False)
    config.parameters.transmission_params = sample_then_resolve(config.parameters.transmission_params)
    sol = ex_m._solve(config, tf, observe=PoissonObservation(compartment=config.idx.c, data=obs_data, increments=False, floor=1e-6))
    handlers.factor("cumulative", sol.log_likelihood)


chains = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pot = Potential(model, dict(config=ex_m.get_config(6), tf=120, obs_data=values), 0, torch.device("cuda"))
f = folded.discover(pot)
z0 = pot.initial(chains, init_to_median, 3)
f.map_now(z0)
for rep in range(2):
    for hints in ({"general_instance": 1}, {}):
        with engine.dispatch_hints(**hints):
            for _ in range(5):
                f.solve_current(chains)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f.solve_current(chains)
            e1.record()
            torch.cuda.synchronize()
            print(hints, f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch, {_abi.lib().dyn_last_kernel_name().decode()}", flush=True)
