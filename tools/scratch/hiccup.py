"""Scratch: find the sporadic slow timed region - per-launch host timestamps + a GPU-side event per launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gc
import bench
dev = torch.device("cuda:0")
T0 = time.perf_counter()
torch.zeros(1, device=dev)
T1 = time.perf_counter()
leg = bench.Leg("m1", dev, None, 1, 0, "weak", 256, 4)
gc.collect(); gc.disable()
leg.prime()
leg.run_steps(0, 20)
def region(n):
    torch.cuda.synchronize()
    ts, evs = [], []
    t0 = time.perf_counter()
    i = 0
    ramp = [1, 1, 2]
    while i < n:
        k = ramp.pop(0) if ramp else 4
        if k > 1:
            leg.tr.step_many([leg.draw(i + j) for j in range(k)], leg.sb, leg.data.ss, global_B=leg.Bg, out=False)
        else:
            leg.one_step(i)
        i += k
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    g = [0.0] + [evs[0].elapsed_time(e) for e in evs[1:]]
    return total, np.array(ts) * 1e3, np.array(g)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    tr0 = time.perf_counter() - T1
    total, ts, g = region(200)
    dh, dg = np.diff(ts), np.diff(g)
    print("[region starts %.2f s after CUDA init] region %2d: %.3f ms/step; host launch gaps max %.2f ms at launch %d (median %.3f); GPU event gaps max %.2f ms at launch %d (median %.3f)"
          % (tr0, rep, 1e3 * total / 200, dh.max(), dh.argmax(), np.median(dh), dg.max(), dg.argmax(), np.median(dg)))
