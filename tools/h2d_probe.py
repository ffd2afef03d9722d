import torch, time
dev = torch.device("cuda:0")
n = 2 + 2048 * 32
host = torch.empty(n).pin_memory()
dst = torch.empty(n, device=dev)
big = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream()
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return dt
def h2d_main(): dst.copy_(host, non_blocking=True)
def h2d_side():
    with torch.cuda.stream(side): dst.copy_(host, non_blocking=True)
print("idle: h2d main %.1f us, h2d side %.1f us" % (t(h2d_main), t(h2d_side)))
def busy_then(f):
    def g():
        for _ in range(3): torch.mm(big, big)      # ~ms of GPU work queued on main
        t0 = time.perf_counter(); f(); return time.perf_counter() - t0
    return g
for name, f in (("h2d main", h2d_main), ("h2d side", h2d_side)):
    g = busy_then(f); g(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        ts.append(g()); torch.cuda.synchronize()
    print("busy GPU: %s host-blocking %.1f us" % (name, sum(ts) / len(ts) * 1e6))
# device-side read of pinned memory (zero copy)
def zc(): dst.copy_(host.to(dev, non_blocking=True) if False else host, non_blocking=True)
import os
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
