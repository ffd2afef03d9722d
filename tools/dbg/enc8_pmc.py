"""Driver for tools/pmc_one.sh: 40 stand-alone launches of the encoder forward at the metric shape (B = 2048, T = 8, weight
images registered) in the variant SW_ENC8 selects - SQ counters of the 4-wave against the 8-wave kernel (profiles/r05_enc8_ab.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
B, T = 2048, 8
x = torch.rand(B, T, 2, device="cuda").cumsum(1) * 0.1
hT, cT = torch.empty(B, 64, device="cuda"), torch.empty(B, 64, device="cuda")
act = torch.empty(T * B * 384, device="cuda"); x4s = torch.empty(T * B * 4, device="cuda")
for _ in range(40):
    L.call("sw_enc_lstm_fwd", L.ptr(x), 0, L.ptr(enc), None, None, B, T, L.ptr(hT), L.ptr(cT), None, L.ptr(act), L.ptr(x4s), 0, L.stream())
torch.cuda.synchronize()
