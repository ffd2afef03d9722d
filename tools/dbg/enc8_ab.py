"""Round-5 pilot A/B: enc_lstm_fwd on 4 waves (one per SIMD) vs the 8-wave kernel (SW_ENC8=1, two per SIMD, weights split
8 ways) - stand-alone launches with the step's weight images registered: results compared bit for bit (through files: the
switch is read once per process), time per launch at T = 8 and T = 40 (-> per-step and fixed cost), at m1 and c4 sizes.
  python tools/dbg/enc8_ab.py            (runs both variants as subprocesses)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) < 2:
    outs = {}
    for v in ("0", "1"):
        env = dict(os.environ, SW_ENC8=v)
        outs[v] = subprocess.run([sys.executable, __file__, "/tmp/enc8_%s.pt" % v], env=env, capture_output=True, text=True)
        print("SW_ENC8=%s\n%s%s" % (v, outs[v].stdout, outs[v].stderr[-2000:] if outs[v].returncode else ""))
    import torch
    a, b = torch.load("/tmp/enc8_0.pt"), torch.load("/tmp/enc8_1.pt")
    for k in a:
        print("%-8s identical: %s   max |diff| %.3g" % (k, torch.equal(a[k], b[k]), float((a[k] - b[k]).abs().max())))
    sys.exit(0)
import torch
import socialways_amd as sw
from socialways_amd import _lib as L
torch.manual_seed(0)
G = sw.Generator(use_social=True, device="cuda:0")
G.unify()
enc, dec, emb, att = G.encoder._flat, G.decoder._flat, G.feature_embedder._flat, G.attention._flat
lib = L.load()
img = torch.empty(lib.sw_gen_image_floats(), device="cuda")
L.call("sw_gen_images", L.ptr(enc), L.ptr(dec), L.ptr(emb), L.ptr(att), L.ptr(img), L.stream())
save = {}
for B in (2048, 32768):
    res = {}
    for T in (8, 40):
        gen = torch.Generator(device="cuda").manual_seed(T)
        x = torch.rand(B, T, 2, device="cuda", generator=gen).cumsum(1) * 0.1
        hT, cT = torch.empty(B, 64, device="cuda"), torch.empty(B, 64, device="cuda")
        act = torch.empty(T * B * 384, device="cuda"); x4s = torch.empty(T * B * 4, device="cuda")
        def fwd():
            L.call("sw_enc_lstm_fwd", L.ptr(x), 0, L.ptr(enc), None, None, B, T, L.ptr(hT), L.ptr(cT), None, L.ptr(act), L.ptr(x4s), 0, L.stream())
        for _ in range(5): fwd()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n): fwd()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        res[T] = best
        if B == 2048 and T == 8:
            save.update(hT=hT.cpu(), cT=cT.cpu(), act=act.cpu(), x4s=x4s.cpu())
    per = (res[40] - res[8]) / 32
    print("B %5d: T=8 %.1f us, T=40 %.1f us -> %.3f us/step, fixed %.1f us" % (B, res[8], res[40], per, res[8] - 8 * per))
torch.save(save, sys.argv[1])
