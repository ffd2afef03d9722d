import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import socialways_amd as sw
dev = torch.device("cuda:0")
S, A, To, Tp = 256, 8, 8, 12
B = S * A
torch.manual_seed(0); np.random.seed(0)
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev)
tracks = sw.synth_tracks(S * 8, A, To, Tp, seed=1234)
data = sw.SceneDataset(tracks["obsvs"], tracks["preds"], tracks["batches"], device=dev)
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
out = torch.zeros(4, 3, device=dev)
noise = torch.rand(B, 32)
for i in range(8):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, noise, data.ss, out=out)
torch.cuda.synchronize()
st = list(tr._graphs.values())[0]
for label, sleep in (("back-to-back", 0.0), ("idle 2ms between", 0.002)):
    evs = []
    t0 = time.perf_counter()
    for i in range(40):
        g = st["graph"][i & 1][0][0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        evs.append((e0, e1))
        if sleep:
            torch.cuda.synchronize(); time.sleep(sleep)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 40 * 1e6
    d = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    print("%-18s graph duration mean %.1f min %.1f max %.1f us | wall/iter %.1f us" % (label, d.mean(), d.min(), d.max(), wall))
