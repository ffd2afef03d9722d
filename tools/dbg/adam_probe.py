"""Which arithmetic does torch._fused_adam_ use on this build?  Compare with candidate formulas, bit for bit."""
import torch, numpy as np
torch.manual_seed(0)
n = 200000
dev = "cuda"
w = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01; m = torch.randn(n, device=dev) * 0.01; v = torch.rand(n, device=dev) * 1e-4
lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
step = torch.tensor(3.0, device=dev)
w1, m1, v1 = w.clone(), m.clone(), v.clone()
torch._fused_adam_([w1], [g.clone()], [m1], [v1], [], [step], amsgrad=False, lr=lr, beta1=b1, beta2=b2, weight_decay=0.0, eps=eps, maximize=False, grad_scale=None, found_inf=None)
def cmp(name, a, b):
    print("   %-34s mismatches %6d of %d" % (name, (a != b).sum().item(), a.numel()))
md, gd, vd, wd = m.double(), g.double(), v.double(), w.double()
f32 = torch.float32
print("exp_avg:")
cmp("double: b1*m + (1-b1)*g", (b1 * md + (1 - b1) * gd).float(), m1)
cmp("float: b1f*m + (1-b1f)*g", torch.tensor(b1, dtype=f32, device=dev) * m + (1 - torch.tensor(b1, dtype=f32, device=dev)) * g, m1)
cmp("float lerp: m + (g-m)*(1-b1)", torch.lerp(m, g, 1 - b1), m1)
w_ = 1 - b1
cmp("double lerp m + w*(g-m)", (md + w_ * (gd - md)).float(), m1)
cmp("float fma(w,(g-m),m)", torch.addcmul(m, (g - m), torch.tensor(w_, dtype=f32, device=dev)), m1)
print("exp_avg_sq:")
cmp("double: b2*v + (1-b2)*g*g", (b2 * vd + (1 - b2) * gd * gd).float(), v1)
cmp("float: b2f*v + (1-b2f)*g*g", torch.tensor(b2, dtype=f32, device=dev) * v + (1 - torch.tensor(b2, dtype=f32, device=dev)) * g * g, v1)
cmp("float addcmul(b2f*v, g, g, 1-b2)", torch.addcmul(torch.tensor(b2, dtype=f32, device=dev) * v, g, g, value=1 - b2), v1)
print("param (from torch's own m1, v1):")
bc1 = 1 - b1 ** 3.0; bc2s = (1 - b2 ** 3.0) ** 0.5
ss_f = np.float32(lr / np.float32(bc1))
den = ((v1.sqrt() / np.float32(bc2s)).double() + eps).float()
cmp("float: w - ss*m/den", w - (ss_f * m1) / den, w1)
den2 = (v1.sqrt() / np.float32(bc2s)) + np.float32(eps)
cmp("float: den all float", w - (ss_f * m1) / den2, w1)
cmp("addcdiv(w, m, den, -ss)", torch.addcdiv(w, m1, den, value=-float(ss_f)), w1)
ss_d = lr / bc1
cmp("double: w - ss_d*m/den_d", (wd - ss_d * m1.double() / ((v1.double().sqrt() / bc2s) + eps)).float(), w1)
print("more exp_avg candidates:")
diff_f = (g - m)
cmp("m_d + w_d * float(g-m)", (md + w_ * diff_f.double()).float(), m1)
cmp("fma-like: (1-b1)*g + b1*m (double, other order)", ((1 - b1) * gd + b1 * md).float(), m1)
w32 = np.float32(1 - b1)
cmp("m + float(w)*(g-m) in double", (md + float(w32) * (gd - md)).float(), m1)
cmp("b1_f32 as double: b1f*m + (1-b1f)*g", (float(np.float32(b1)) * md + (1 - float(np.float32(b1))) * gd).float(), m1)
