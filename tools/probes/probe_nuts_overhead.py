"""Dev probe: where does a NUTS iteration spend its time on cfg 4?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynode_amd.infer.inference import Potential
from examples import sir_infer_parameters as ex
data = ex.synthetic_incidence(100)
pot = Potential(ex.model, dict(config=ex.get_config(), tf=100, obs_data=data), 0, torch.device("cuda"))
z = pot.initial(128, ("median", 15), 0)
for _ in range(5): pot.potential_and_grad(z)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): u, g = pot.potential_and_grad(z)
torch.cuda.synchronize(); el = (time.perf_counter() - t) / 50
print("potential_and_grad (eager): %.3f ms" % (el * 1e3))
# CUDA-graph capture of the whole potential + gradient
static_z = z.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        pot.potential_and_grad(static_z)
torch.cuda.current_stream().wait_stream(s)
g_ = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g_):
        su, sg = pot.potential_and_grad(static_z)
    torch.cuda.synchronize()
    static_z.copy_(z + 0.01); g_.replay(); torch.cuda.synchronize()
    u2, g2 = pot.potential_and_grad(z + 0.01)
    print("graph == eager:", torch.allclose(su, u2, rtol=1e-6), torch.allclose(sg, g2, rtol=1e-4, atol=1e-6))
    t = time.perf_counter()
    for _ in range(200): g_.replay()
    torch.cuda.synchronize(); print("potential_and_grad (graph replay): %.3f ms" % ((time.perf_counter() - t) / 200 * 1e3))
except Exception as e:
    import traceback; traceback.print_exc()
