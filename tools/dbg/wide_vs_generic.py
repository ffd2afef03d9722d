"""Debug aid: the wide engine (wide.py) against the layer-by-layer generic path (generic.py, torch's tape) on the same
weights and inputs - losses, rollout, every gradient, weights after the step.  python tools/dbg/wide_vs_generic.py [H] [nl]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import socialways_amd as sw
from socialways_amd.generic import GenericTrainer
from socialways_amd.wide import WideTrainer
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(7)
a = WideTrainer(12, hidden_size=H, n_latent_codes=nl, device="cuda:0", use_graph=False)
torch.manual_seed(7)
b = GenericTrainer(12, hidden_size=H, n_latent_codes=nl, device="cuda:0")
for p, q in zip(list(a.G.parameters()) + list(a.D.parameters()), list(b.G.parameters()) + list(b.D.parameters())):
    assert torch.equal(p, q)
t = sw.synth_tracks(8, [5, 1, 9, 16, 3, 2, 2, 2], 8, 12, seed=5)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
B, sb = 36, data.the_batches[:6]
gen = torch.Generator().manual_seed(2)
def rel(x, y):
    return float((x - y).abs().max()) / max(float(y.abs().max()), 1e-12)
for it in range(3):
    noise = torch.rand(B, H // 2, generator=gen)
    ra = a.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    rb = b.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    print("step", it, "res rel err", rel(ra, rb), "rollout", rel(a.last_pred_hat, b.last_pred_hat))
    if it == 0:
        print(ra.cpu().numpy()); print(rb.cpu().numpy())
    worst = 0
    for (n, p), (_, q) in zip(a.G.named_parameters(), b.G.named_parameters()):
        e = rel(p.grad, q.grad) if q.grad is not None else -1
        worst = max(worst, e)
        if e > 2e-4: print("  dG", n, e)
    for (n, p), (_, q) in zip(list(a.G.named_parameters()) + list(a.D.named_parameters()), list(b.G.named_parameters()) + list(b.D.named_parameters())):
        e = float((p - q).abs().max())
        if e > 2e-4: print("  W ", n, e)
    print("  worst G grad rel err", worst)
# graph replay == eager
torch.manual_seed(7)
c = WideTrainer(12, hidden_size=H, n_latent_codes=nl, device="cuda:0", use_graph=True)
torch.manual_seed(7)
d = WideTrainer(12, hidden_size=H, n_latent_codes=nl, device="cuda:0", use_graph=False)
gen = torch.Generator().manual_seed(3)
for it in range(6):
    noise = torch.rand(B, H // 2, generator=gen)
    rc = c.step(data.obsv[:B], data.pred[:B], sb, 0.01 * it, 0.94, noise, data.ss)
    rd = d.step(data.obsv[:B], data.pred[:B], sb, 0.01 * it, 0.94, noise, data.ss)
    print("graph vs eager step", it, torch.equal(rc, rd), torch.equal(c.gp.flat, d.gp.flat), torch.equal(c.dp.flat, d.dp.flat))
# speed at the metric shape
S, A = 256, 8
t = sw.synth_tracks(S, A, 8, 12, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
B, sb = S * A, data.the_batches[:S]
noise = torch.rand(B, H // 2)
for tr, name in ((c, "wide graph"), (d, "wide eager")):
    for _ in range(4):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.94, noise, data.ss)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s: %.3f ms/step = %.1f steps/s at %d x %d agents, H = %d" % (name, 1e3 * dt, 1 / dt, S, A, H))
