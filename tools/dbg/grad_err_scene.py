import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import socialways_amd as sw
import sw_oracle as O
A = 64
t = sw.synth_tracks(512, A, 8, 12, seed=32)
torch.manual_seed(0)
social = os.environ.get("SOCIAL", "1") == "1"
tr = sw.SocialWaysTrainer(12, use_social=social, device="cuda:0")
o2 = O.SocialWaysOracle(12, use_social=social)
o2.load_state({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in tr.checkpoint().items() if k.endswith("_dict")})
torch.set_default_dtype(torch.float64)
for m in (o2.attention, o2.feature_embedder, o2.encoder, o2.decoder):
    m.double()
torch.set_default_dtype(torch.float32)
torch.manual_seed(2)
z_all = torch.rand(512 * A, 32); cot_all = torch.randn(512 * A, 12, 4) * 0.1
obsv_all = torch.from_numpy(t["obsvs"])
key = ("decoder", "fc1.0.weight")
nag = int(os.environ.get("NAG", A))
starts = [sc * A for sc in range(32)] if "SCENE" not in os.environ else [int(os.environ["SCENE"]) * A + a for a in range(0, A, nag)]
for sc, r0 in enumerate(starts):
    r1 = r0 + nag
    sb = np.array([[0, nag]])
    for n in ("attention", "feature_embedder", "encoder", "decoder"):
        for p in list(getattr(tr.G, n).parameters()) + list(getattr(o2, n).parameters()):
            p.grad = None
    out = tr.G(obsv_all[r0:r1].cuda(), z_all[r0:r1].cuda(), 12, sb)
    out.backward(cot_all[r0:r1].cuda())
    torch.set_default_dtype(torch.float64)
    ref = o2.predict(obsv_all[r0:r1].double(), z_all[r0:r1].double(), 12, sb)
    ref.backward(cot_all[r0:r1].double())
    torch.set_default_dtype(torch.float32)
    errs = []
    for n, k in (("decoder", "fc1.0.weight"), ("decoder", "fc1.5.weight"), ("encoder", "lstm.weight_hh_l0")):
        g = dict(getattr(tr.G, n).named_parameters())[k].grad.double().cpu()
        w = dict(getattr(o2, n).named_parameters())[k].grad
        errs.append(((g - w).abs().max() / w.abs().max()).item())
    print("scene %2d: fwd err %.2e  grad err fc1.0.w %.2e fc1.5.w %.2e whh %.2e" % (sc, (out.detach().double().cpu() - ref.detach()).abs().max(), *errs))
