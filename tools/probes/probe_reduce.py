import os, sys, time, torch
x = torch.randn(16384, 366*136, device="cuda")
def T(f, n=3):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("sum dim0 f32      %.2f ms" % T(lambda: x.sum(0)))
print("sum dim0 ->f64    %.2f ms" % T(lambda: x.sum(0, dtype=torch.float64)))
ones = torch.ones(16384, device="cuda")
print("mv x.T @ ones     %.2f ms" % T(lambda: torch.mv(x.T, ones)))
print("ones @ x (mm)     %.2f ms" % T(lambda: ones[None] @ x))
print("square+sum dim0   %.2f ms" % T(lambda: x.square().sum(0)))
print("var_mean dim0     %.2f ms" % T(lambda: torch.var_mean(x, 0)))
xx = x.view(16, 1024, -1)
print("chunked sum(1).sum(0) %.2f ms" % T(lambda: xx.sum(1).sum(0)))
