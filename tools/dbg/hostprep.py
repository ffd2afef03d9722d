import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
import socialways_amd as sw
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
leg = bench.Leg("m1", dev, None, 1, 0, "weak", 0, 32)
for i in range(6): leg.one_step(i)
leg.run_steps(0, 64)
torch.cuda.synchronize()
buf = leg._zring[0]
def t(fn, n=200):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("torch.rand(out=buf) us:", t(lambda: torch.rand(buf.shape, out=buf)))
print("np.random.uniform x2 us:", t(lambda: (np.random.uniform(0, 0.1), np.random.uniform(0.9, 1.0))))
print("draw() us:", t(lambda: leg.draw(0)))
# host time of one step() call (GPU async): run when the GPU queue is short
def one():
    leg.one_step(0)
torch.cuda.synchronize()
ts = []
for i in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); one(); ts.append((time.perf_counter() - t0) * 1e6)
print("one_step host us (idle GPU): median %.1f min %.1f" % (np.median(ts), min(ts)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200): one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
