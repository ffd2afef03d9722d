"""Scratch: what the staging launch spends its time on (eager launches bracketed by HIP events)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import socialways_amd as sw
from socialways_amd import _lib as L
import bench
dev = torch.device("cuda:0")
leg = bench.Leg("m1", dev, None, 1, 0, "weak", 256, 4)
leg.prime()
tr = leg.tr
st = next(iter(tr._stage.values())) if hasattr(tr, "_stage") else None
lib = L.load()
B, To, Tp = leg.B, leg.To, leg.Tp
slot = torch.zeros(8 + B * 64, dtype=torch.float32).pin_memory()
o, p, zv, ov, noise = leg.draw(0)
hn = slot.numpy(); hn[:4].view(np.uint64)[:] = (o.data_ptr(), p.data_ptr())
obsv = torch.empty(B, To, 2, device=dev); pred = torch.empty(B, Tp, 2, device=dev); pred4 = torch.empty(B, Tp, 4, device=dev)
tg = torch.empty(4, device=dev); steps = torch.zeros(4, device=dev)
G = tr.G
def run(img, dimg, n=50):
    lib.sw_debug_spin(600.0, L.stream())
    lib.sw_kernel_timing(1)
    for _ in range(n):
        L.call("sw_stage_step_img", slot.data_ptr(), B, To, Tp, L.ptr(obsv), L.ptr(pred), L.ptr(pred4), L.ptr(tg), None,
               L.ptr(steps), 2, L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), L.ptr(G.feature_embedder._flat),
               L.ptr(G.attention._flat), L.ptr(tr._gimg) if img else None,
               L.ptr(tr.D._flat) if dimg else None, L.ptr(tr._dimg) if dimg else None, L.ptr(tr._dtab) if dimg else None, L.stream())
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.sw_kernel_timing_read(buf, len(buf)); lib.sw_kernel_timing(0)
    for ln in buf.value.decode().splitlines():
        k, c, t = ln.split()
        if "stage" in k: return float(t) / int(c)
for img in (1, 0):
    for dimg in (1, 0):
        run(img, dimg, 10)
        print("images G=%d D=%d: %.2f us" % (img, dimg, run(img, dimg)))
