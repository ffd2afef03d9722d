"""Scratch: where the first ~150 us of a timed window go (host side of the first step behind a fence)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gc
import bench
dev = torch.device("cuda:0")
leg = bench.Leg("m1", dev, None, 1, 0, "weak", 256, 4)
gc.collect(); gc.disable()
leg.prime(); leg.run_steps(0, 20)
acc = np.zeros(4)
N = 50
for rep in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = leg.draw(rep)
    t1 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    o, p, zv, ov, noise = d
    leg.tr.step(o, p, leg.sb, zv, ov, noise, leg.data.ss, global_B=leg.Bg, out=False)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print("draw z %.1f us, step() host call %.1f us, wait for the GPU %.1f us, total %.1f us (GPU step ~370 us)" % tuple(1e6 * acc / N))
