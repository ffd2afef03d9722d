"""The reference's model surface (crowdbotp/socialways train.py) on MI355X.

Same class / function names, constructor arguments, `forward` signatures and state_dict keys as
the reference, so code written against train.py reads the same here:

    get_traj_4d            train.py:130-138        SocialFeatures        train.py:229-241
    AttentionPooling       train.py:153-175        EmbedSocialFeatures   train.py:178-189
    EncoderLstm            train.py:245-269        DecoderFC             train.py:320-335
    Discriminator          train.py:272-316        predict()             train.py:392-432

The reference has no Generator class: `Generator` here only groups the four
sub-modules `predict()` reads as module globals; its checkpoint still splits into the four
reference dicts.  All compute is the HIP library (socialways_amd/_lib.py); parameters of each
module live in ONE packed buffer (the layout the C ABI reads) and the nn.Parameters are views of
it, so state_dict()/load_state_dict()/torch.optim work unchanged.

Differentiable entry points: the two functions train() differentiates - `predict()` / `Generator.forward` and
`Discriminator.forward` (fused kernels) - and the stand-alone sub-modules `EncoderLstm`, `DecoderFC`,
`EmbedSocialFeatures`, `AttentionPooling` (any scene size), which record autograd graphs of their own (`_EncFn`,
`_DecFn`, `_EmbFn`, `_AttFn`: the same forward kernels plus their backward passes, csrc/sw_modules.hip) so that a model
composed differently from predict() still trains.  `SocialFeatures` / `get_traj_4d` act on track data and carry no
gradient.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops


# ------------------------------------------------------------------------------------------------
def _ar(n, base=0):
    return torch.arange(n, dtype=torch.long) + base


def _gate_rows(H):
    """LSTM gate rows (i, f, g, o blocks of H) inside the kernels' 4 x 64 layout."""
    return torch.cat([_ar(H, 64 * g) for g in range(4)])


def _check_hidden(H, what):
    """The kernels are built for 64 hidden units (train.py's default `--hidden-size`).  Smaller networks run on them
    EXACTLY, zero-padded: a padded unit has zero weights and biases, so it stays at 0 through every layer (lrelu(0) = 0,
    an LSTM unit with zero pre-activations keeps c = h = 0), feeds nothing forward, and receives zero gradients."""
    if H > 64 or H < 8 or H % 8:
        raise L.SocialWaysHipError("%s: hidden size %d - the kernels hold 64 hidden units per layer in registers; sizes "
                                   "8, 16, .. 64 run on them (smaller ones zero-padded); larger widths train on the "
                                   "generic-width path (socialways_amd.generic / SocialWaysTrainer(hidden_size=...))" % (what, H))


class _Packed(nn.Module):
    """Parameters as views of one packed fp32 buffer (C-ABI layout) plus a packed grad buffer.

    With a hidden size below 64 the parameters have the KERNEL shapes (padded with zeros); `_true` lists, per parameter,
    (true shape, index of the true entries inside the flattened padded tensor).  state_dict() / load_state_dict() speak
    the reference's true shapes (hooks below), so checkpoints interchange with a reference run at that `--hidden-size`."""
    _GRP = None
    _true = None

    def _tp(self):
        return 1

    @staticmethod
    def _idx(rows, cols=None, ncols=None):
        return rows.clone() if cols is None else (rows[:, None] * ncols + cols[None, :]).reshape(-1)

    def _adopt(self, true_mods, maps):
        """Padded construction: `true_mods` were built first with the reference's shapes (they consumed the RNG exactly
        like train.py:370-384 does), the kernel-shaped modules afterwards with the generator state restored; copy the
        true entries into their padded places, zero everything else."""
        true_params = [q for m in true_mods for q in m.parameters()]
        mine = list(self.parameters())
        assert len(true_params) == len(mine) == len(maps)
        self._true = []
        with torch.no_grad():
            for p, q, idx in zip(mine, true_params, maps):
                assert idx.numel() == q.numel() and int(idx.max()) < p.numel(), (tuple(p.shape), tuple(q.shape))
                p.zero_()
                p.view(-1)[idx] = q.reshape(-1)
                self._true.append((tuple(q.shape), idx))
        self._register_state_dict_hook(_Packed._shrink_hook)
        self._register_load_state_dict_pre_hook(self._expand_hook)

    @staticmethod
    def _shrink_hook(module, state_dict, prefix, local_metadata):
        for (name, p), (shape, idx) in zip(module.named_parameters(), module._true):
            key = prefix + name
            if key in state_dict:
                state_dict[key] = state_dict[key].reshape(-1)[idx.to(state_dict[key].device)].view(shape).clone()

    def _expand_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for (name, p), (shape, idx) in zip(self.named_parameters(), self._true):
            key = prefix + name
            t = state_dict.get(key)
            if t is not None and tuple(t.shape) == shape and tuple(p.shape) != shape:
                full = torch.zeros(p.shape, dtype=t.dtype, device=t.device)
                full.view(-1)[idx.to(t.device)] = t.reshape(-1)
                state_dict[key] = full

    def load_state_dict(self, state_dict, *a, **k):
        return super().load_state_dict(dict(state_dict) if self._true is not None else state_dict, *a, **k)

    def true_view(self, i, t):
        """Parameter-shaped tensor `t` (a gradient, an optimizer moment) of parameter i in the reference's shape."""
        if self._true is None:
            return t
        shape, idx = self._true[i]
        return t.reshape(-1)[idx.to(t.device)].view(shape)

    def pad_mask(self):
        """1.0 on the live entries of the packed buffer, 0.0 on zero padding (all ones at 64 hidden units)."""
        m = torch.ones_like(self._flat) if self._true is None else torch.zeros_like(self._flat)
        if self._true is not None:
            for (off, k), (shape, idx) in zip(self._slices, self._true):
                m[off:off + k][idx.to(m.device)] = 1.0
        return m

    def _move(self, device):
        """Parameters are initialised on the CPU generator exactly like the reference (which builds
        on the CPU and then calls .cuda(), train.py:370-384) and moved afterwards."""
        if device is not None and torch.device(device).type != "cpu":
            self.to(device)

    def _pack(self, into=None):
        params = list(self.named_parameters())
        lib = L.load()
        tp = self._tp()
        n = lib.sw_param_count(self._GRP, tp)
        assert lib.sw_param_tensors(self._GRP) == len(params), (type(self).__name__, len(params))
        dev = params[0][1].device
        if into is not None:                   # slices of a buffer shared with sibling modules
            flat, gflat = into
            assert flat.numel() == n and gflat.numel() == n
        else:
            flat = torch.zeros(n, dtype=torch.float32, device=dev)
            gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._slices = []
        for idx, (name, p) in enumerate(params):
            off = lib.sw_param_offset(self._GRP, idx, tp)
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat[off:off + k].view(p.shape)
            self._slices.append((off, k))
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_gflat", gflat)
        return self

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)      # .to()/.cuda() re-homes parameters: re-pack
        if getattr(self, "_flat", None) is not None:
            self._pack()
        return out

    def packed(self):
        """The packed weight buffer the kernels read (re-packs if a parameter was re-pointed)."""
        flat = self._flat
        base = flat.data_ptr()
        for (off, k), p in zip(self._slices, self.parameters()):
            if p.data_ptr() != base + 4 * off:
                self._pack()
                return self._flat
        return flat

    def grad_views(self):
        """Point every p.grad at its slice of the packed gradient buffer and return the buffer."""
        g = self._gflat
        for (off, k), p in zip(self._slices, self.parameters()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr() + 4 * off:
                p.grad = g[off:off + k].view(p.shape)
        return g

    def split_grad(self, gflat):
        return [gflat[off:off + k].view(p.shape) for (off, k), p in zip(self._slices, self.parameters())]


def _scene_index(sub_batches, B, device):
    return ops.SceneIndex.get(sub_batches if len(sub_batches) else np.zeros((0, 2), np.int64), B, device)


# ------------------------------------------------------------------------------------------------
def get_traj_4d(obsv_p, pred_p):
    """(x,y) -> (x,y,vx,vy) (train.py:130-138); `pred_p=[]` returns the observation part only."""
    L.require_gpu(obsv_p)
    obsv_p = obsv_p.contiguous()
    B, To = obsv_p.shape[0], obsv_p.shape[1]
    o4 = torch.empty(B, To, 4, device=obsv_p.device)
    if len(pred_p) == 0:
        L.call("sw_traj_4d", L.ptr(obsv_p), None, B, To, 0, L.ptr(o4), None, L.stream())
        return o4
    pred_p = pred_p.contiguous()
    Tp = pred_p.shape[1]
    p4 = torch.empty(B, Tp, 4, device=obsv_p.device)
    L.call("sw_traj_4d", L.ptr(obsv_p), L.ptr(pred_p), B, To, Tp, L.ptr(o4), L.ptr(p4), L.stream())
    return o4, p4


def SocialFeatures(x, sub_batches):
    """Dense (B,B,3) [dist, bearing, dca] of the last observed state (train.py:229-241);
    `sub_batches` is ignored, as in the reference."""
    L.require_gpu(x)
    last = x[:, -1].contiguous()
    B = last.shape[0]
    feat = torch.empty(B, B, 3, device=x.device)
    L.call("sw_social_features", L.ptr(last), B, L.ptr(feat), L.stream())
    return feat


def _wants_grad(module, *tensors):
    return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if t is not None)
                                        or any(p.requires_grad for p in module.parameters()))


def _wgrad_ws(dev):
    return ops.default_ws(dev).get("wgrad", L.workspace_floats(L.WS_WGRAD, 1, 2, 1))


class _AttFn(torch.autograd.Function):
    """AttentionPooling.forward on a dense (B,B,F) tensor (train.py:160-174), any scene size."""

    @staticmethod
    def forward(ctx, att, f, h, sc, *params):
        B, dev = h.shape[0], h.device
        w = att.packed()
        f, h = f.contiguous(), h.contiguous()
        wh = torch.empty(B, 64, device=dev)           # W h + b
        L.call("sw_rows_gemm", L.ptr(h), 64, L.ptr(w), 1, 64, w.data_ptr() + 4 * 4096, B, 64, 64, L.ptr(wh), 64, 0, L.stream())
        attn = torch.zeros(B, B, device=dev)
        S = torch.empty(B, 64, device=dev)
        L.call("sw_attention_dense_fwd", L.ptr(f), L.ptr(h), L.ptr(wh), L.ptr(sc.scene_off), sc.S, B, L.ptr(attn), L.ptr(S),
               L.stream())
        ctx.att, ctx.sc, ctx.need_f = att, sc, ctx.needs_input_grad[1]
        ctx.save_for_backward(f, h, wh, attn)
        return S

    @staticmethod
    def backward(ctx, dS):
        att, sc = ctx.att, ctx.sc
        f, h, wh, attn = ctx.saved_tensors
        B, dev = h.shape[0], h.device
        w = att.packed()
        dsig = torch.empty(B, B, device=dev)
        df = torch.zeros(B, B, 64, device=dev) if ctx.need_f else None
        dwh, dh = torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev)
        L.call("sw_attention_dense_bwd", L.ptr(f), L.ptr(h), L.ptr(wh), L.ptr(attn), L.ptr(dS.contiguous()), L.ptr(sc.scene_off),
               sc.S, B, L.ptr(dsig), L.ptr(df), L.ptr(dwh), L.ptr(dh), L.stream())
        # dh += W^T dWh ; dW = dWh^T h ; db = sum dWh
        L.call("sw_rows_gemm", L.ptr(dwh), 64, L.ptr(w), 64, 1, None, B, 64, 64, L.ptr(dh), 64, 1, L.stream())
        gflat = torch.zeros_like(w)
        L.call("sw_linear_wgrad", L.ptr(dwh), 64, L.ptr(h), 64, B, 64, 64, L.ptr(gflat), 64, gflat.data_ptr() + 4 * 4096,
               L.ptr(_wgrad_ws(dev)), 0, L.stream())
        return (None, df, dh, None) + tuple(att.split_grad(gflat))


class _EmbFn(torch.autograd.Function):
    """EmbedSocialFeatures.forward (train.py:185-189) on any (..., 3) feature tensor."""

    @staticmethod
    def forward(ctx, emb, x, *params):
        x = x.contiguous()
        R = x.numel() // 3
        out = torch.empty(*x.shape[:-1], 64, device=x.device)
        L.call("sw_embed_features", L.ptr(x), R, L.ptr(emb.packed()), L.ptr(out), L.stream())
        ctx.emb, ctx.need_x = emb, ctx.needs_input_grad[1]
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        emb = ctx.emb
        (x,) = ctx.saved_tensors
        dev = x.device
        R = x.numel() // 3
        if R >= 2 ** 31:
            raise L.SocialWaysHipError("EmbedSocialFeatures backward: %d rows" % R)
        w = emb.packed()
        dout = dout.contiguous()
        rows = torch.empty(196 * R, device=dev)
        dx = torch.empty_like(x) if ctx.need_x else None
        L.call("sw_embed_features_bwd", L.ptr(x), R, L.ptr(w), L.ptr(dout), L.ptr(rows), L.ptr(dx), L.stream())
        h2, dh2, h1, dh1, f4 = (rows[a * R:b * R] for a, b in ((0, 64), (64, 128), (128, 160), (160, 192), (192, 196)))
        gflat = torch.zeros_like(w)
        off = [o for o, _ in emb._slices]      # fc.0.weight, fc.0.bias, fc.2.weight, fc.2.bias, fc.4.weight, fc.4.bias
        ws, g0 = _wgrad_ws(dev), gflat.data_ptr()
        for delta, ldd, act, lda, N, K, iw, ib in ((dout, 64, h2, 64, 64, 64, 4, 5), (dh2, 64, h1, 32, 64, 32, 2, 3),
                                                   (dh1, 32, f4, 4, 32, 3, 0, 1)):
            L.call("sw_linear_wgrad", L.ptr(delta), ldd, L.ptr(act), lda, R, N, K, g0 + 4 * off[iw], K, g0 + 4 * off[ib],
                   L.ptr(ws), 0, L.stream())
        return (None, dx) + tuple(emb.split_grad(gflat))


class _EncFn(torch.autograd.Function):
    """EncoderLstm.forward (train.py:262-269): embed + LSTM over T steps from the state (h0, c0)."""

    @staticmethod
    def forward(ctx, enc, x, h0, c0, *params):
        B, T, dev = x.shape[0], x.shape[1], x.device
        x, h0, c0 = x.contiguous(), h0.contiguous(), c0.contiguous()
        hT, cT = torch.empty_like(h0), torch.empty_like(c0)
        y = torch.empty(B, T, 64, device=dev)
        act, x4s = torch.empty(T, B, 384, device=dev), torch.empty(T, B, 4, device=dev)
        L.call("sw_enc_lstm_fwd", L.ptr(x), 1, L.ptr(enc.packed()), L.ptr(h0), L.ptr(c0), B, T, L.ptr(hT), L.ptr(cT), L.ptr(y),
               L.ptr(act), L.ptr(x4s), 0, L.stream())
        ctx.enc, ctx.need_x, ctx.need_state = enc, ctx.needs_input_grad[1], (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        ctx.save_for_backward(act, x4s, h0, c0)
        return y, hT, cT

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        enc = ctx.enc
        act, x4s, h0, c0 = ctx.saved_tensors
        T, B, dev = act.shape[0], act.shape[1], act.device
        w = enc.packed()
        zeros = lambda: torch.zeros(B, 64, device=dev)
        dhT = zeros() if dhT is None else dhT.contiguous()
        dcT = zeros() if dcT is None else dcT.contiguous()
        dy = None if dy is None else dy.contiguous()
        dgates = torch.empty(T, B, 256, device=dev)
        dh0, dc0 = torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev)
        L.call("sw_enc_lstm_bwd", L.ptr(w), L.ptr(act), L.ptr(c0), L.ptr(dhT), L.ptr(dcT), L.ptr(dy), B, T, 0, L.ptr(dgates),
               L.ptr(dh0), L.ptr(dc0), L.stream())
        gflat = torch.zeros_like(w)
        tmp = torch.empty(2048, device=dev)
        L.call("sw_enc_lstm_wgrad", L.ptr(w), L.ptr(act), L.ptr(x4s), L.ptr(h0), L.ptr(dgates), B, T, L.ptr(gflat),
               L.ptr(_wgrad_ws(dev)), L.ptr(tmp), L.stream())
        dx = None
        if ctx.need_x:     # dx_t = (W_ih W_embed)^T dgates_t: the composed 256 x 4 input matrix, then one small product
            off = dict(zip(("embed.weight", "embed.bias", "lstm.weight_ih_l0"), [o for o, _ in enc._slices][:3]))
            wx = torch.empty(256, 4, device=dev)
            L.call("sw_rows_gemm", w.data_ptr() + 4 * off["lstm.weight_ih_l0"], 64, w.data_ptr() + 4 * off["embed.weight"], 4, 1,
                   None, 256, 64, 4, L.ptr(wx), 4, 0, L.stream())
            dxt = torch.empty(T, B, 4, device=dev)
            L.call("sw_rows_gemm", L.ptr(dgates), 256, L.ptr(wx), 4, 1, None, T * B, 256, 4, L.ptr(dxt), 4, 0, L.stream())
            dx = dxt.permute(1, 0, 2).contiguous()
        return (None, dx, dh0 if ctx.need_state else None, dc0 if ctx.need_state else None) + tuple(enc.split_grad(gflat))


class _DecFn(torch.autograd.Function):
    """DecoderFC.forward (train.py:330-335) = one decode step of the rollout kernels (Tp = 1)."""
    TO = 2

    @staticmethod
    def forward(ctx, dec, h, s, z, enc_w, *params):
        B, dev = h.shape[0], h.device
        h, s, z = h.contiguous(), s.contiguous(), z.contiguous()
        zero_obs = torch.zeros(B, _DecFn.TO, 2, device=dev)
        c = torch.zeros(B, 64, device=dev)
        pred4 = torch.empty(B, 1, 4, device=dev)
        gsave = torch.empty(L.workspace_floats(L.WS_GSAVE, B, _DecFn.TO, 1), device=dev)
        L.call("sw_dec_rollout_fwd", L.ptr(zero_obs), _DecFn.TO, L.ptr(z), L.ptr(s), L.ptr(h), L.ptr(c), L.ptr(enc_w),
               L.ptr(dec.packed()), B, 1, L.ptr(pred4), None, None, L.ptr(gsave), None, 0.0, None, L.stream())
        ctx.dec, ctx.enc_w = dec, enc_w
        ctx.need = tuple(ctx.needs_input_grad[1:4])
        ctx.save_for_backward(h, s, z, gsave)
        return pred4[:, 0, 2:4].contiguous()

    @staticmethod
    def backward(ctx, dv):
        dec = ctx.dec
        h, s, z, gsave = ctx.saved_tensors
        B, dev = h.shape[0], h.device
        w = dec.packed()
        dpred4 = torch.zeros(B, 1, 4, device=dev)
        dpred4[:, 0, 2:4] = dv
        gdelta = torch.empty(L.workspace_floats(L.WS_GDELTA, B, _DecFn.TO, 1), device=dev)
        dh, dc, ds = (torch.empty(B, 64, device=dev) for _ in range(3))
        L.call("sw_dec_rollout_bwd", L.ptr(dpred4), L.ptr(ctx.enc_w), L.ptr(w), L.ptr(gsave), B, _DecFn.TO, 1, L.ptr(gdelta),
               L.ptr(dh), L.ptr(dc), L.ptr(ds), L.stream())
        dz = None
        if ctx.need[2]:
            dz = torch.empty(B, 32, device=dev)
            L.call("sw_dec_fc_dz", L.ptr(w), L.ptr(gdelta), B, _DecFn.TO, L.ptr(dz), L.stream())
        gflat = torch.zeros_like(w)
        tmp = torch.empty(2048, device=dev)
        L.call("sw_dec_fc_wgrad", L.ptr(w), L.ptr(gsave), L.ptr(gdelta), L.ptr(h), L.ptr(s), L.ptr(z), B, _DecFn.TO, L.ptr(gflat),
               L.ptr(_wgrad_ws(dev)), L.ptr(tmp), L.stream())
        return (None, dh if ctx.need[0] else None, ds if ctx.need[1] else None, dz, None) + tuple(dec.split_grad(gflat))


class AttentionPooling(_Packed):
    _GRP = L.GRP_ATT

    def __init__(self, h_dim, f_dim, device=None):
        super().__init__()
        if h_dim != f_dim:
            raise L.SocialWaysHipError("AttentionPooling kernels are built for h_dim = f_dim")
        _check_hidden(h_dim, "AttentionPooling")
        self.f_dim, self.h_dim = f_dim, h_dim
        H = h_dim
        if H == 64:
            self.W = nn.Linear(64, 64, bias=True)
            self._pack()
        else:
            true = [nn.Linear(H, H, bias=True)]
            st = torch.get_rng_state()
            self.W = nn.Linear(64, 64, bias=True)
            torch.set_rng_state(st)
            self._pack()
            self._adopt(true, [self._idx(_ar(H), _ar(H), 64), _ar(H)])
        self._move(device)

    def forward(self, f, h, sub_batches):
        """f: dense (B,B,F) pair embeddings (only in-scene blocks are read), h: (B,H).
        sigma_ij=<f_ij, W h_j>, sigma_ii:=-1000, softmax over the scene, S_i = sum_j a_ij h_j."""
        L.require_gpu(h)
        B = h.shape[0]
        if self.h_dim < 64:       # zero-padded to the kernels' 64 units (exact)
            pad = 64 - self.h_dim
            return self.forward_padded(nn.functional.pad(f, (0, pad)), nn.functional.pad(h, (0, pad)), sub_batches)[:, :self.h_dim]
        return self.forward_padded(f, h, sub_batches)

    def forward_padded(self, f, h, sub_batches):
        B = h.shape[0]
        sc = _scene_index(sub_batches, B, h.device)
        if _wants_grad(self, f, h) or sc.NB > 0:    # one workgroup per agent: any scene size, records an autograd graph
            return _AttFn.apply(self, f, h, sc, *self.parameters())
        S = torch.empty(B, 64, device=h.device)     # one workgroup per scene (<= 64 agents staged in LDS)
        L.call("sw_attention_pool_dense", L.ptr(f.contiguous()), L.ptr(h.contiguous()), L.ptr(sc.scene_off), sc.S, B,
               L.ptr(self.packed()), L.ptr(S), L.stream())
        return S


class EmbedSocialFeatures(_Packed):
    _GRP = L.GRP_EMB

    def __init__(self, input_size, hidden_size, device=None):
        super().__init__()
        if input_size != 3:
            raise L.SocialWaysHipError("EmbedSocialFeatures kernels are built for 3 input features")
        _check_hidden(hidden_size, "EmbedSocialFeatures")
        self.input_size, self.hidden_size = input_size, hidden_size
        H = hidden_size
        make = lambda n: nn.Sequential(nn.Linear(input_size, 32), nn.ReLU(), nn.Linear(32, 64), nn.ReLU(), nn.Linear(64, n))
        if H == 64:
            self.fc = make(64)
            self._pack()
        else:
            true = [make(H)]
            st = torch.get_rng_state()
            self.fc = make(64)
            torch.set_rng_state(st)
            self._pack()
            self._adopt(true, [_ar(96), _ar(32), _ar(2048), _ar(64), self._idx(_ar(H), _ar(64), 64), _ar(H)])
        self._move(device)

    def forward(self, ftr_list, sub_batches):
        out = self.forward_padded(ftr_list, sub_batches)
        return out if self.hidden_size == 64 else out[..., :self.hidden_size]

    def forward_padded(self, ftr_list, sub_batches):
        L.require_gpu(ftr_list)
        if _wants_grad(self, ftr_list):
            return _EmbFn.apply(self, ftr_list, *self.parameters())
        x = ftr_list.contiguous()
        R = x.numel() // 3
        out = torch.empty(*x.shape[:-1], 64, device=x.device)
        L.call("sw_embed_features", L.ptr(x), R, L.ptr(self.packed()), L.ptr(out), L.stream())
        return out


class EncoderLstm(_Packed):
    _GRP = L.GRP_ENC

    def __new__(cls, hidden_size=None, n_layers=2, device=None):
        # The fused kernels are built for ONE layer of <= 64 units (what train.py:82 constructs).  The class signature's
        # default of 2 stacked layers (train.py:246) and wider encoders are served by the generic-width module: same
        # parameters / state_dict keys / initialisation, its steps run layer by layer through the C-ABI pieces.
        # copy.deepcopy / pickle / torch.load rebuild an object with cls.__new__(cls) and NO arguments: that is not a
        # constructor call and must give a bare instance of THIS class (hidden_size is None only on that path).
        if cls is EncoderLstm and hidden_size is not None and (n_layers != 1 or hidden_size > 64):
            from . import generic
            return generic.EncoderLstm(hidden_size, n_layers, device)
        return super().__new__(cls)

    def __init__(self, hidden_size, n_layers=1, device=None):
        self.hidden_size = hidden_size
        super().__init__()
        _check_hidden(hidden_size, "EncoderLstm")
        H = hidden_size
        if H == 64:
            self.embed = nn.Linear(4, 64)
            self.lstm = nn.LSTM(64, 64, num_layers=1, batch_first=True)
            self._pack()
        else:
            true = [nn.Linear(4, H), nn.LSTM(H, H, num_layers=1, batch_first=True)]
            st = torch.get_rng_state()
            self.embed = nn.Linear(4, 64)
            self.lstm = nn.LSTM(64, 64, num_layers=1, batch_first=True)
            torch.set_rng_state(st)
            self._pack()
            g = _gate_rows(H)
            self._adopt(true, [self._idx(_ar(H), _ar(4), 4), _ar(H), self._idx(g, _ar(H), 64), self._idx(g, _ar(H), 64), g, g])
        self.lstm_h = []
        self._move(device)

    def init_lstm(self, h, c):
        self.lstm_h = (h, c)

    def forward(self, obsv):
        """obsv (B,T,4) or (B,4): embed + LSTM from the stored state; returns y (B,T,H) and keeps
        the new state in `self.lstm_h` (shape (1,B,H) each), like train.py:262-269."""
        L.require_gpu(obsv)
        bs = obsv.shape[0]
        x = obsv.reshape(bs, -1, 4).contiguous()
        T = x.shape[1]
        H = self.hidden_size
        h0 = self.lstm_h[0].reshape(bs, H)
        c0 = self.lstm_h[1].reshape(bs, H)
        if H < 64:          # zero-padded to the kernels' 64 units (exact: padded units stay at 0)
            h0, c0 = nn.functional.pad(h0, (0, 64 - H)), nn.functional.pad(c0, (0, 64 - H))
            if _wants_grad(self, x, h0, c0):
                y, hT, cT = _EncFn.apply(self, x, h0, c0, *self.parameters())
            else:
                y, hT, cT = self._forward_nograd(x, h0, c0)
            self.lstm_h = (hT[:, :H].reshape(1, bs, H), cT[:, :H].reshape(1, bs, H))
            return y[..., :H]
        if _wants_grad(self, x, h0, c0):
            y, hT, cT = _EncFn.apply(self, x, h0, c0, *self.parameters())
            self.lstm_h = (hT.view(1, bs, 64), cT.view(1, bs, 64))
            return y
        h0, c0 = h0.contiguous(), c0.contiguous()
        hT, cT = torch.empty_like(h0), torch.empty_like(c0)
        y = torch.empty(bs, T, 64, device=x.device)
        L.call("sw_enc_lstm_fwd", L.ptr(x), 1, L.ptr(self.packed()), L.ptr(h0), L.ptr(c0), bs, T, L.ptr(hT), L.ptr(cT),
               L.ptr(y), None, None, 0, L.stream())
        self.lstm_h = (hT.view(1, bs, 64), cT.view(1, bs, 64))
        return y

    def _forward_nograd(self, x, h0, c0):
        bs, T = x.shape[0], x.shape[1]
        h0, c0 = h0.contiguous(), c0.contiguous()
        hT, cT = torch.empty_like(h0), torch.empty_like(c0)
        y = torch.empty(bs, T, 64, device=x.device)
        L.call("sw_enc_lstm_fwd", L.ptr(x), 1, L.ptr(self.packed()), L.ptr(h0), L.ptr(c0), bs, T, L.ptr(hT), L.ptr(cT),
               L.ptr(y), None, None, 0, L.stream())
        return y, hT, cT


class DecoderFC(_Packed):
    _GRP = L.GRP_DEC

    def __init__(self, hidden_dim, device=None):
        super().__init__()
        if hidden_dim % 5 or (2 * hidden_dim) % 5:
            raise L.SocialWaysHipError("DecoderFC: hidden_dim = 2.5 x hidden size (train.py:375), got %d" % hidden_dim)
        H = 2 * hidden_dim // 5
        _check_hidden(H, "DecoderFC")
        self.hidden_size = H
        make = lambda d: nn.Sequential(nn.Linear(d, d), nn.LeakyReLU(0.2), nn.Linear(d, d // 2), nn.LeakyReLU(0.2),
                                       nn.Linear(d // 2, d // 4), nn.Linear(d // 4, 2))
        if H == 64:
            self.fc1 = make(160)
            self._pack()
        else:
            D = hidden_dim
            true = [make(D)]
            st = torch.get_rng_state()
            self.fc1 = make(160)
            torch.set_rng_state(st)
            self._pack()
            cin = torch.cat([_ar(H), _ar(H, 64), _ar(H // 2, 128)])      # cat[h, s, z] inside the kernels' 64 | 64 | 32 columns
            self._adopt(true, [self._idx(_ar(D), cin, 160), _ar(D), self._idx(_ar(D // 2), _ar(D), 160), _ar(D // 2),
                               self._idx(_ar(D // 4), _ar(D // 2), 80), _ar(D // 4), self._idx(_ar(2), _ar(D // 4), 40), _ar(2)])
        self._move(device)

    def forward(self, h, s, z, _encoder=None):
        """cat[h,s,z] -> velocity (B,2) (train.py:330-335): one decode step of the rollout kernel."""
        L.require_gpu(h)
        H = self.hidden_size
        if H < 64:          # zero-padded inputs (exact)
            pad = nn.functional.pad
            h, s, z = pad(h, (0, 64 - H)), pad(s, (0, 64 - H)), pad(z, (0, 32 - H // 2))
        B = h.shape[0]
        dev = h.device
        zero_obs = torch.zeros(B, 2, 2, device=dev)
        c = torch.zeros(B, 64, device=dev)
        pred4 = torch.empty(B, 1, 4, device=dev)
        enc_w = _encoder.packed() if _encoder is not None else torch.zeros(L.load().sw_param_count(L.GRP_ENC, 1), device=dev)
        if _wants_grad(self, h, s, z):
            return _DecFn.apply(self, h, s, z, enc_w, *self.parameters())
        L.call("sw_dec_rollout_fwd", L.ptr(zero_obs), 2, L.ptr(z.contiguous()), L.ptr(s.contiguous()),
               L.ptr(h.contiguous()), L.ptr(c), L.ptr(enc_w), L.ptr(self.packed()), B, 1, L.ptr(pred4), None, None, None,
               None, 0.0, None, L.stream())
        return pred4[:, 0, 2:4].contiguous()


# ------------------------------------------------------------------------------------------------
class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, D, obsv, pred, *params):
        labels, codes, dctx = ops.disc_forward(D.packed(), obsv, [pred], save=True)
        ctx.D, ctx.dctx = D, dctx
        ctx.need_pred = pred.requires_grad
        ctx.need_w = any(p.requires_grad for p in params)
        return labels[0], codes[0]

    @staticmethod
    def backward(ctx, dlabel, dcode):
        D = ctx.D
        dev = dlabel.device
        B = ctx.dctx.B
        dlabel = torch.zeros(B, 1, device=dev) if dlabel is None else dlabel
        dcode = torch.zeros(B, 2, device=dev) if dcode is None else dcode
        gflat = torch.empty_like(D._flat) if ctx.need_w else None
        dpreds = ops.disc_backward(D.packed(), ctx.dctx, [dlabel], [dcode], gflat, [ctx.need_pred])
        grads = D.split_grad(gflat) if ctx.need_w else [None] * len(D._slices)
        return (None, None, dpreds[0]) + tuple(grads)


class Discriminator(_Packed):
    _GRP = L.GRP_DISC

    def __init__(self, n_next, hidden_dim, n_latent_code, device=None):
        super().__init__()
        if n_latent_code != 2:
            raise L.SocialWaysHipError("Discriminator kernels are built for n_latent_code=2")
        _check_hidden(hidden_dim, "Discriminator")
        self.lstm_dim = hidden_dim
        self.n_next = n_next
        H = hidden_dim

        def make(d):
            return [nn.LSTM(4, d, batch_first=True),
                    nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, d // 2)),
                    nn.Sequential(nn.Linear(n_next * 4, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, d // 2)),
                    nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, 1)),
                    nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, n_latent_code))]
        true = make(H) if H < 64 else None
        st = torch.get_rng_state()
        (self.obsv_encoder_lstm, self.obsv_encoder_fc, self.pred_encoder, self.classifier, self.latent_decoder) = make(64)
        if H < 64:
            torch.set_rng_state(st)
        self._pack()
        if H < 64:
            Hh, K4, g = H // 2, 4 * n_next, _gate_rows(H)
            both = torch.cat([_ar(Hh), _ar(Hh, 32)])      # cat[obsv_code, pred_code] inside the kernels' 32 | 32 columns
            I = self._idx
            self._adopt(true, [I(g, _ar(4), 4), I(g, _ar(H), 64), g, g,
                               I(_ar(Hh), _ar(H), 64), _ar(Hh), I(_ar(Hh), _ar(Hh), 32), _ar(Hh),
                               I(_ar(Hh), _ar(K4), K4), _ar(Hh), I(_ar(Hh), _ar(Hh), 32), _ar(Hh),
                               I(_ar(Hh), both, 64), _ar(Hh), I(_ar(1), _ar(Hh), 32), _ar(1),
                               I(_ar(Hh), both, 64), _ar(Hh), I(_ar(2), _ar(Hh), 32), _ar(2)])
        self._move(device)

    def _tp(self):
        return self.n_next

    def forward(self, obsv, pred):
        """obsv (B,To,4), pred (B,Tp,4) -> (label (B,1) raw LSGAN score, code_hat (B,2))."""
        if torch.is_grad_enabled() and (pred.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DiscFn.apply(self, obsv, pred, *self.parameters())
        labels, codes, _ = ops.disc_forward(self.packed(), obsv, [pred], save=False)
        return labels[0], codes[0]

    def load(self, backup):
        """Restore nn.Linear weights/biases only; the LSTM keeps its update (train.py:311-316)."""
        for m_from, m_to in zip(backup.modules(), self.modules()):
            if isinstance(m_to, nn.Linear):
                m_to.weight.data.copy_(m_from.weight.data)
                if m_to.bias is not None:
                    m_to.bias.data.copy_(m_from.bias.data)

    def linear_mask(self):
        """1.0 on the packed slots of nn.Linear parameters (what `load()` restores), else 0."""
        m = torch.zeros_like(self._flat)
        for (off, k), (name, _) in zip(self._slices, self.named_parameters()):
            if "lstm" not in name:
                m[off:off + k] = 1.0
        return m


# ------------------------------------------------------------------------------------------------
class _PredictFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, obsv, noise, n_next, scenes, *params):
        pred4, gctx = ops.gen_forward(G.encoder.packed(), G.feature_embedder.packed(), G.attention.packed(),
                                      G.decoder.packed(), obsv, noise, scenes, n_next, G.use_social, save=True)
        ctx.G, ctx.gctx = G, gctx
        return pred4

    @staticmethod
    def backward(ctx, dpred4):
        G = ctx.G
        mods = (G.attention, G.feature_embedder, G.encoder, G.decoder)
        gf = {m: torch.empty_like(m._flat) for m in mods}
        ops.gen_backward(G.encoder.packed(), G.feature_embedder.packed(), G.attention.packed(), G.decoder.packed(),
                         ctx.gctx, dpred4, gf[G.encoder], gf[G.feature_embedder], gf[G.attention], gf[G.decoder])
        grads = []
        for m in mods:
            grads += m.split_grad(gf[m])
        return (None, None, None, None, None) + tuple(grads)


class Generator(nn.Module):
    """encoder + feature_embedder + attention + decoder in the reference's construction order
    (train.py:370-375: fixes the RNG -> init mapping) and `predict()` as forward."""

    def __init__(self, hidden_size=64, n_lstm_layers=1, use_social=False, device=None):
        super().__init__()
        self.encoder = EncoderLstm(hidden_size, n_lstm_layers)
        self.feature_embedder = EmbedSocialFeatures(3, hidden_size)
        self.attention = AttentionPooling(hidden_size, hidden_size)
        self.decoder = DecoderFC(hidden_size + hidden_size + hidden_size // 2)
        self.use_social = use_social            # train.py:83 hard-codes False; the flag is explicit here
        self.hidden_size = hidden_size
        self.noise_len = hidden_size // 2
        if device is not None and torch.device(device).type != "cpu":
            self.to(device)                     # built on the CPU generator first, like train.py:370-375

    def predictor_params(self):
        """Parameter order of the reference's generator optimizer (train.py:379-380)."""
        from itertools import chain
        return chain(self.attention.parameters(), self.feature_embedder.parameters(),
                     self.encoder.parameters(), self.decoder.parameters())

    def unify(self):
        """Re-home the four packed weight / gradient buffers into ONE buffer each (optimizer order,
        16-byte aligned), so a data-parallel step all-reduces a single flat gradient."""
        mods = (self.attention, self.feature_embedder, self.encoder, self.decoder)
        sizes = [m._flat.numel() for m in mods]
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += (n + 3) // 4 * 4
        dev = mods[0]._flat.device
        flat_all = torch.zeros(o, dtype=torch.float32, device=dev)
        gflat_all = torch.zeros(o, dtype=torch.float32, device=dev)
        for m, off, n in zip(mods, offs, sizes):
            m._pack(into=(flat_all[off:off + n], gflat_all[off:off + n]))
        object.__setattr__(self, "_flat_all", flat_all)
        object.__setattr__(self, "_gflat_all", gflat_all)
        return self

    def grad_views(self):
        for m in (self.attention, self.feature_embedder, self.encoder, self.decoder):
            m.grad_views()

    def packed_slices(self):
        """[(offset, numel, shape)] of every parameter inside the unified buffer, optimizer order."""
        out = []
        base = self._flat_all.data_ptr()
        for m in (self.attention, self.feature_embedder, self.encoder, self.decoder):
            moff = (m._flat.data_ptr() - base) // 4
            for i, ((off, k), p) in enumerate(zip(m._slices, m.parameters())):
                out.append((moff + off, k, tuple(p.shape)) + (m._true[i] if m._true is not None else ()))
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if getattr(self, "_flat_all", None) is not None:
            self.unify()
        return out

    def forward(self, obsv_p, noise, n_next, sub_batches=[]):
        """predict(obsv_p (B,To,2), noise (B, hidden_size / 2), n_next, sub_batches) -> pred_hat_4d (B,n_next,4)."""
        L.require_gpu(obsv_p)
        if noise.shape[-1] < 32:     # smaller hidden size: zero-padded to the kernels' 32 noise columns (their weights are zero)
            noise = nn.functional.pad(noise, (0, 32 - noise.shape[-1]))
        scenes = _scene_index(sub_batches, obsv_p.shape[0], obsv_p.device)
        params = list(self.predictor_params())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _PredictFn.apply(self, obsv_p, noise, n_next, scenes, *params)
        pred4, _ = ops.gen_forward(self.encoder.packed(), self.feature_embedder.packed(), self.attention.packed(),
                                   self.decoder.packed(), obsv_p, noise, scenes, n_next, self.use_social, save=False)
        return pred4


_default_generator = None


def set_default_generator(g):
    global _default_generator
    _default_generator = g


def predict(obsv_p, noise, n_next, sub_batches=[], generator=None):
    """Module-level predict() with the reference's signature (train.py:392); the four sub-modules
    it reads as globals in the reference are those of `generator` (or the default generator)."""
    g = generator or _default_generator
    if g is None:
        raise RuntimeError("no generator: pass generator= or call set_default_generator()")
    return g(obsv_p, noise, n_next, sub_batches)


def predict_cv(obsv, n_next):
    """Constant-velocity baseline of test() (utils/linear_models.py:9-20)."""
    n_past = obsv.shape[1]
    my_vel = (obsv[:, -1] - obsv[:, -3]) / 2. if n_past > 2 else (obsv[:, -1] - obsv[:, -2])
    last, out = obsv[:, -1], []
    for _ in range(n_next):                     # repeated addition, like the reference's loop
        last = last + my_vel
        out.append(last)
    return torch.stack(out, dim=1)
