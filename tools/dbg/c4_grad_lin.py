"""Debug: generator gradients at dense-crowd sizes - full batch vs sum over scene chunks (HIP) vs fp64 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O

A = 64
t = sw.synth_tracks(512, A, 8, 12, seed=32)
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0")
orc = O.SocialWaysOracle(12, use_social=True)
orc.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
obsv_all = torch.from_numpy(t["obsvs"]).cuda()
sb_all = np.asarray(t["batches"])
torch.manual_seed(2)
z_all = torch.rand(512 * A, 32)
cot_all = torch.randn(512 * A, 12, 4) * 0.1
names = [(n, k) for n in ("attention", "feature_embedder", "encoder", "decoder") for k, _ in getattr(tr.G, n).named_parameters()]

def hip_grads(lo, hi):
    r0, r1 = int(sb_all[lo, 0]), int(sb_all[hi - 1, 1])
    for n in ("attention", "feature_embedder", "encoder", "decoder"):
        for p in getattr(tr.G, n).parameters():
            p.grad = None
    out = tr.G(obsv_all[r0:r1], z_all[r0:r1].cuda(), 12, sb_all[lo:hi] - r0)
    out.backward(cot_all[r0:r1].cuda())
    return {(n, k): p.grad.detach().double().cpu().clone() for n in ("attention", "feature_embedder", "encoder", "decoder")
            for k, p in getattr(tr.G, n).named_parameters()}

for S in (64, 256, 512):
    full = hip_grads(0, S)
    chunks = None
    for c in range(0, S, 32):
        g = hip_grads(c, c + 32)
        chunks = g if chunks is None else {k: chunks[k] + g[k] for k in g}
    worst = max(((full[k] - chunks[k]).abs().max() / chunks[k].abs().max().clamp_min(1e-30)).item() for k in full)
    wk = max(full, key=lambda k: ((full[k] - chunks[k]).abs().max() / chunks[k].abs().max().clamp_min(1e-30)).item())
    print("S=%d: full vs sum of 32-scene chunks: worst max|d|/max = %.3e at %s" % (S, worst, wk))
    k = ("feature_embedder", "fc.2.bias")
    d = (full[k] - chunks[k]) / chunks[k].abs().max()
    print("   fc.2.bias rel diff per unit: max at unit %d: %.3e ; unit 49: %.3e" % (int(d.abs().argmax()), d.abs().max(), d[49]))

# one 32-scene chunk vs the oracle in fp32 and fp64
S = 32
B = S * A
g = hip_grads(0, S)
res = {}
for dt in (torch.float32, torch.float64):
    o2 = O.SocialWaysOracle(12, use_social=True)
    o2.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
    torch.set_default_dtype(dt)
    for m in (o2.attention, o2.feature_embedder, o2.encoder, o2.decoder):
        m.to(dt)
    ref = o2.predict(obsv_all[:B].cpu().to(dt), z_all[:B].to(dt), 12, sb_all[:S])
    ref.backward(cot_all[:B].to(dt))
    torch.set_default_dtype(torch.float32)
    res[dt] = {(n, k): p.grad.double() for n in ("attention", "feature_embedder", "encoder", "decoder")
               for k, p in getattr(o2, n).named_parameters()}
for k in [("feature_embedder", "fc.2.bias"), ("feature_embedder", "fc.2.weight"), ("feature_embedder", "fc.4.bias"), ("feature_embedder", "fc.0.weight"), ("attention", "W.weight")]:
    w = res[torch.float64][k]
    mx = w.abs().max()
    print(k, "max %.4g | hip-f64 %.3e | f32-f64 %.3e (of max)" % (mx, (g[k] - w).abs().max() / mx, (res[torch.float32][k] - w).abs().max() / mx))
k = ("feature_embedder", "fc.2.bias")
d = (g[k] - res[torch.float64][k])
print("fc.2.bias hip-f64 per unit (abs):", np.array2string(d.numpy(), precision=2, max_line_width=200))
print("fc.2.bias f64:", np.array2string(res[torch.float64][k].numpy(), precision=3, max_line_width=200))
