"""Scratch: fixed cost of a weight-gradient job (one problem, 64 workgroups, rows per wave swept); -DSW_WG_STAMP variant."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from socialways_amd import _lib as L
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.environ["SW_LIB_PATH"])
ws = torch.empty(1 << 24, device=dev)
for N, K in ((64, 64), (160, 32), (256, 64)):
    for rpw in (256, 512):
        R = 256 * rpw
        delta = torch.randn(R, N, device=dev); act = torch.randn(R, K, device=dev)
        dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        for _ in range(3):
            L.call("sw_linear_wgrad", L.ptr(delta), N, L.ptr(act), K, R, N, K, L.ptr(dW), K, L.ptr(db), L.ptr(ws), 0, L.stream())
        torch.cuda.synchronize()
        n = 4096
        buf = (ctypes.c_ulonglong * (4 * n))()
        lib.sw_debug_wg_stamps(buf, 4 * n)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
        tj = int(a[0, 3]); a = a[:tj]
        d = (a[:, 1] - a[:, 0]) / 100.0
        span = (a[:, 1].max() - a[:, 0].min()) / 100.0
        ref = delta.double().t() @ act.double()
        err = (dW.double() - ref).abs().max().item() / ref.abs().max().item()
        ix = np.arange(tj)
        print("   per XCD: " + " ".join("%5.1f" % d[ix % 8 == x].mean() for x in range(8)))
        print("N %3d K %2d R %6d: jobs %4d rows/wave %6.1f  dur min %5.1f avg %5.1f max %5.1f  span %5.1f  err %.1e" % (N, K, R, tj, R / (4.0 * tj / ((N + 63) // 64)), d.min(), d.mean(), d.max(), span, err))
