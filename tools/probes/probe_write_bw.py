"""dev probe: pure-write, pure-read and copy bandwidth of the device for buffers of the size of one cfg-3 launch
(3.26 GB), to put the solve kernel's 3.2 TB/s of compulsory writes next to what the memory system sustains."""
import torch
n = 3_272_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: x.fill_(1.0)); print(f"fill   {n * 4 / ms / 1e6:8.0f} GB/s written ({ms:.3f} ms)")
ms = timed(lambda: x.zero_()); print(f"memset {n * 4 / ms / 1e6:8.0f} GB/s written ({ms:.3f} ms)")
ms = timed(lambda: y.copy_(x)); print(f"copy   {2 * n * 4 / ms / 1e6:8.0f} GB/s read + written ({ms:.3f} ms)")
ms = timed(lambda: x.sum()); print(f"sum    {n * 4 / ms / 1e6:8.0f} GB/s read ({ms:.3f} ms)")
ms = timed(lambda: torch.add(x, 1.0, out=y)); print(f"add    {2 * n * 4 / ms / 1e6:8.0f} GB/s read + written ({ms:.3f} ms)")
