"""Is the generator's weight-gradient launch bound by where its rows live?  One eager m1 step, then sw_gen_wgrad again
and again on the same save / delta buffers (190 MB: they now sit in the 256 MB Infinity Cache), each timed by events;
between two of the runs 400 MB of other memory are swept to push the rows out again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import ops, _lib as L

S, A, To, Tp = 256, 8, 8, 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, use_graph=False)
tr._fuse_g_adam = False
t = sw.synth_tracks(S, A, To, Tp, seed=1)
data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device=dev)
B = S * A
sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
stash = {}
orig = ops.gen_backward
def spy(enc_w, emb_w, att_w, dec_w, ctx, dpred4, d_enc, d_emb, d_att, d_dec, ws=None, tag="g", aux=None, adam=None):
    stash.update(enc_w=enc_w, dec_w=dec_w, ctx=ctx, d_enc=d_enc, d_dec=d_dec, ws=ws, tag=tag)
    return orig(enc_w, emb_w, att_w, dec_w, ctx, dpred4, d_enc, d_emb, d_att, d_dec, ws=ws, tag=tag, aux=aux, adam=adam)
ops.gen_backward = spy
for _ in range(2):
    tr.step(data.obsv[:B], data.pred[:B], sb, 0.05, 0.95, torch.rand(B, 32), data.ss, out=False)
torch.cuda.synchronize()
c, ws = stash["ctx"], stash["ws"]
gdelta = ws.get(stash["tag"] + ".gdelta", L.workspace_floats(L.WS_GDELTA, B, To, Tp))
wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
tmp = ws.get(stash["tag"] + ".dwx", 2048)
def run():
    L.call("sw_gen_wgrad", L.ptr(stash["enc_w"]), L.ptr(stash["dec_w"]), L.ptr(c.gsave), L.ptr(gdelta), L.ptr(c.noise), L.ptr(c.S),
           B, To, Tp, L.ptr(stash["d_enc"]), L.ptr(stash["d_dec"]), 0, L.ptr(wgrad), L.ptr(tmp), None, L.stream())
junk = torch.empty(100 * 1024 * 1024, device=dev)     # 400 MB
def timed(n, flush):
    ts = []
    for _ in range(n):
        if flush:
            junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return ts
print("back to back (rows in the Infinity Cache):   ", " ".join("%.1f" % x for x in timed(6, False)), "us (partial + reduce + compose, no social problems)")
print("400 MB swept in between (rows from HBM):     ", " ".join("%.1f" % x for x in timed(6, True)), "us")
