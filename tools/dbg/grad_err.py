"""Debug: per-tensor error of the HIP generator gradients vs the fp64 oracle (and of the fp32 oracle), small dense batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import ops
import sw_oracle as O

A = int(os.environ.get("A", "64")); S = int(os.environ.get("S", "8"))
t = sw.synth_tracks(int(os.environ.get("NS", S)), A, 8, 12, seed=32); t = {k: (v[:S * A] if k != "batches" else v[:S]) for k, v in t.items()}
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(12, use_social=os.environ.get("SOCIAL", "1") == "1", device="cuda:0")
B = S * A
obsv = torch.from_numpy(t["obsvs"]).cuda(); sb = np.asarray(t["batches"])
torch.manual_seed(2)
z = torch.rand(B, 32); cot = torch.randn(B, 12, 4) * 0.1
G = tr.G
scenes = ops.SceneIndex.get(sb, B, obsv.device)
pred4, ctx = ops.gen_forward(G.encoder._flat, G.feature_embedder._flat, G.attention._flat, G.decoder._flat, obsv, z.cuda(), scenes, 12, G.use_social, save=True)
out = tr.G(obsv, z.cuda(), 12, sb)
out.backward(cot.cuda())
g = {(n, k): p.grad.detach().double().cpu() for n in ("attention", "feature_embedder", "encoder", "decoder") for k, p in getattr(tr.G, n).named_parameters()}
res, fw = {}, {}
for dt in (torch.float32, torch.float64):
    o2 = O.SocialWaysOracle(12, use_social=G.use_social)
    o2.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
    torch.set_default_dtype(dt)
    for m in (o2.attention, o2.feature_embedder, o2.encoder, o2.decoder):
        m.to(dt)
    ref = o2.predict(obsv.cpu().to(dt), z.to(dt), 12, sb)
    ref.backward(cot.to(dt))
    torch.set_default_dtype(torch.float32)
    res[dt] = {(n, k): (p.grad.double() if p.grad is not None else torch.zeros_like(p).double()) for n in ("attention", "feature_embedder", "encoder", "decoder") for k, p in getattr(o2, n).named_parameters()}
    fw[dt] = dict(pred=ref.detach().double(), hT=o2.last["hT"].detach().double(), S=o2.last["S"].detach().double())
f64 = fw[torch.float64]
for nm, hip in (("pred", pred4), ("hT", ctx.hT), ("S", ctx.S)):
    w = f64[nm]; mx = w.abs().max().clamp_min(1e-30)
    print("fwd %-5s max %.3g | hip-f64 %.3e | f32-f64 %.3e (of max)" % (nm, mx, (hip.double().cpu() - w).abs().max() / mx, (fw[torch.float32][nm] - w).abs().max() / mx))
for k in g:
    w = res[torch.float64][k]; mx = w.abs().max().clamp_min(1e-30)
    print("%-40s max %.3g | hip-f64 %.3e | f32-f64 %.3e" % (".".join(k), mx, (g[k] - w).abs().max() / mx, (res[torch.float32][k] - w).abs().max() / mx))
