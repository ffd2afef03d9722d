"""Debug aid: per-parameter generator-gradient error of the wide / generic trainers against the oracle for one configuration.
python tools/dbg/wide_case.py H sizes(comma) To Tp social U"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O
H = int(sys.argv[1]); sizes = [int(x) for x in sys.argv[2].split(",")]; To, Tp = int(sys.argv[3]), int(sys.argv[4])
social, U = bool(int(sys.argv[5])), int(sys.argv[6])
t = sw.synth_tracks(len(sizes) + 2, sizes + [2, 2], To, Tp, seed=77)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
torch.manual_seed(5)
kw = dict(use_social=social, n_unrolling_steps=U)
tr = sw.SocialWaysTrainer(Tp, hidden_size=H, device="cuda:0", **kw)
orc = O.SocialWaysOracle(Tp, hidden_size=H, **kw)
orc.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
B, sb = int(np.sum(sizes)), data.the_batches[:len(sizes)]
noise = torch.rand(B, H // 2)
rec = {}
out = tr.step(data.obsv[:B], data.pred[:B], sb, 0.03, 0.95, noise, data.ss)
want, ade, fde = orc.train_step(data.obsv[:B].cpu(), data.pred[:B].cpu(), sb, 0.03, 0.95, noise, data.ss, record=rec)
print(type(tr).__name__, "losses", np.asarray(tr.losses_from(out, [B], Tp, data.ss)[0]), "\n oracle", np.asarray(want))
for name in ("attention", "feature_embedder", "encoder", "decoder"):
    for k, p in getattr(tr.G, name).named_parameters():
        w = rec["g_grads"].get(name + "." + k)
        if w is None or p.grad is None: continue
        print("%-34s |g|max %.3e  err/max %.2e" % (name + "." + k, float(w.abs().max()), float((p.grad.cpu() - w).abs().max()) / max(float(w.abs().max()), 1e-12)))
