"""Kernel sequencing for the two differentiable functions of the reference's train():
`predict()` (train.py:392-432) and `Discriminator.forward` (train.py:294-309).

These are thin host wrappers: they own the workspaces (torch allocator), put the C-ABI calls of
include/socialways_hip.h in order on the current stream and nothing else.  Both the autograd
Functions of model.py and the fused training step of trainer.py go through them.
"""
import os

import numpy as np
import torch

from . import _lib as L


class SceneIndex:
    """Device-side form of the reference's `sub_batches` ((S,2) [start,end) rows, train.py:446-461):
    int32 prefix offsets; scenes of up to AMAX agents: int64 pair offsets (n^2 per scene with n > 1) and
    their largest size; larger scenes: the row-block records of include/socialways_hip.h (`big_blocks`)."""

    def __init__(self, sub_batches, B, device):
        sb = np.asarray(sub_batches, dtype=np.int64).reshape(-1, 2)
        if len(sb) == 0:                                   # predict() default: one scene (train.py:405-406)
            sb = np.array([[0, B]], dtype=np.int64)
        if sb[0, 0] != 0 or sb[-1, 1] != B or (sb[1:, 0] != sb[:-1, 1]).any():
            raise ValueError("sub_batches must tile [0, B) contiguously")
        n = sb[:, 1] - sb[:, 0]
        if (n <= 0).any():
            raise ValueError("empty scene in sub_batches")
        self.S = len(sb)
        self.B = int(B)
        small = n <= L.AMAX
        self.amax = int(n[small].max()) if small.any() else 0      # largest one-workgroup scene
        pairs = np.where((n > 1) & small, n * n, 0)
        recs, row = [], 0
        for sc in np.nonzero(~small)[0]:                          # scenes above AMAX agents: blocks of 16 query agents
            ns = int(n[sc])
            nblk = (ns + 15) // 16
            for k in range(nblk):
                recs.append((sc, 16 * k, row + k * ns, row, nblk, k, 0, 0))
            row += nblk * ns
        self.NB, self.big_rows = len(recs), row
        self.big_blocks = torch.tensor(recs, dtype=torch.int32).reshape(-1, 8).to(device) if recs else None
        poff = np.concatenate([[0], np.cumsum(pairs)]).astype(np.int64)
        self.P = int(poff[-1])
        off = np.concatenate([sb[:, 0], [B]]).astype(np.int32)
        self.scene_off = torch.from_numpy(off).to(device)
        self.pair_off = torch.from_numpy(poff).to(device)
        self.sizes = n

    _cache = {}

    @classmethod
    def get(cls, sub_batches, B, device):
        sb = np.ascontiguousarray(np.asarray(sub_batches, dtype=np.int64))
        key = (sb.tobytes(), int(B), str(device))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 4096:
                cls._cache.clear()
            hit = cls._cache[key] = cls(sb, B, device)
            hit.key = key
        return hit


class Workspaces:
    """Grow-only fp32 scratch buffers keyed by name (saves, deltas, split-K partials) and the host-side handle
    that lets the weight-gradient problems of one backward pass share a launch."""

    def __init__(self, device):
        self.device = device
        self.buf = {}
        self._wgb = None
        # Growth protocol: a captured step graph has the ADDRESSES of these buffers baked in.  When a larger batch
        # layout needs more space the old tensor is not released - it moves to `retired`, so graphs captured for
        # smaller layouts keep replaying on live memory - and `version` is bumped; the owner of the graphs
        # (SocialWaysTrainer) drops its graphs and then calls `release_retired()` at its next safe point.
        self.retired = []
        self.version = 0

    @property
    def wgrad_batch(self):
        if self._wgb is None:
            lib = L.load()
            self._wgb, self._free = lib.sw_wgrad_batch_new(), lib.sw_wgrad_batch_free
        return self._wgb

    def __del__(self):      # may run at interpreter shutdown, when module globals are already gone
        if getattr(self, "_wgb", None):
            self._free(self._wgb)
            self._wgb = None

    def get(self, name, nfloats):
        t = self.buf.get(name)
        if t is None or t.numel() < nfloats:
            if t is not None:
                self.retired.append(t)
                self.version += 1
            # geometric growth: ragged datasets with slowly increasing batch sizes would otherwise retire a buffer per step
            n = max(int(nfloats), 1) if t is None else max(int(nfloats), int(1.25 * t.numel()))
            t = torch.empty(n, dtype=torch.float32, device=self.device)
            self.buf[name] = t
        return t

    def release_retired(self):
        """Only when nothing captured can still reference the outgrown buffers."""
        self.retired.clear()


_default_ws = {}


def default_ws(device):
    key = str(device)
    if key not in _default_ws:
        _default_ws[key] = Workspaces(device)
    return _default_ws[key]


class GenCtx:
    __slots__ = ("obsv", "noise", "scenes", "hT", "cT", "S", "attn", "gsave", "B", "To", "Tp", "use_social", "wh", "ml")


def gen_forward(enc_w, emb_w, att_w, dec_w, obsv, noise, scenes, n_next, use_social, save, ws=None, tag="g", ade=None,
                noise_src=None, d_obs=None):
    """predict(): encode obs (train.py:397-404), social pooling (408-413), decode loop (415-432).
    Returns pred_hat_4d (B, n_next, 4) and, if `save`, the context backward needs.
    ade = (gt (B,n_next,2), 1/ss, out (ceil(B/16),3)): the decode kernel also leaves the per-tile ADE/FDE
    partial sums of train.py:546-551 in `out`.
    noise_src: address `noise` is filled from (pinned host memory) by idle workgroups of the encoder launch.
    d_obs = (packed D weights, dsave buffer): idle workgroups of the decode launch run the discriminator's
    observation LSTM for the next disc_forward(..., save_lstm=2) on the same obsv (see d_obs_buffer)."""
    L.require_gpu(obsv)
    obsv = obsv.contiguous()
    noise = noise.contiguous()
    B, To = obsv.shape[0], obsv.shape[1]
    if noise.shape != (B, 32):
        raise ValueError("noise must be (B, 32)")
    dev = obsv.device
    st = L.stream()
    hT = torch.empty(B, 64, device=dev)
    cT = torch.empty(B, 64, device=dev)
    pred4 = torch.empty(B, n_next, 4, device=dev)
    gsave = None
    if save:   # ws=None (autograd path): a private save buffer per call, several may be alive
        nfl = L.workspace_floats(L.WS_GSAVE, B, To, n_next)
        gsave = ws.get(tag + ".gsave", nfl) if ws is not None else torch.empty(nfl, device=dev)
    # act rows live at offset 0 of gsave, x4s right behind (sw_common.h:gsave_layout)
    x4s_off = (To + n_next - 1) * B * 384
    # z (256 KB over PCIe: ~16 us, request-bound) is first read by the decode launch.  The encoder launch leaves half of
    # the CUs idle for ~20 us at the metric shape: its spare workgroups pull z for free.
    # (Dense crowds: 4 MB of z at c4 are ~170 us of request-bound PCIe reads against 113 us of encoder work.  Splitting the
    # pull over this launch and the social block's moved the cost, 168 + 148 -> 135 + 185 us: the CUs that host the pulling
    # workgroups serve their tiles / scenes late.  A caller that wants it gone hands z over in device memory.)
    z_in_enc, z_in_soc, n_enc = noise_src, None, noise.numel() if noise_src else 0
    L.call("sw_enc_lstm_fwd_aux", L.ptr(obsv), 0, L.ptr(enc_w), None, None, B, To, L.ptr(hT), L.ptr(cT), None,
           L.ptr(gsave), (gsave.data_ptr() + 4 * x4s_off) if save else None, 0,
           z_in_enc, L.ptr(noise) if z_in_enc else None, n_enc, st)
    attn = wh = ml = None
    if use_social:
        S = torch.empty(B, 64, device=dev)
        attn = torch.empty(B, L.AMAX, device=dev) if save else None
        if scenes.NB:          # scenes above AMAX agents: W h + b per agent and the softmax statistics of every row
            wh = torch.empty(B * 132, device=dev)       # Wh | v | c rows of the row-block kernels (sw_social_pool_fwd)
            ml = torch.empty(B, 2, device=dev) if save else None
        L.call("sw_social_pool_fwd_aux", L.ptr(obsv), To, L.ptr(hT), L.ptr(scenes.scene_off), scenes.S, B, scenes.amax,
               L.ptr(emb_w), L.ptr(att_w), L.ptr(S), L.ptr(attn), L.ptr(scenes.big_blocks), scenes.NB, L.ptr(wh), L.ptr(ml),
               z_in_soc, (noise.data_ptr() + 4 * n_enc) if z_in_soc else None, (noise.numel() - n_enc) if z_in_soc else 0, st)
    else:
        S = torch.zeros(B, 64, device=dev)                                   # train.py:413
    L.call("sw_dec_rollout_fwd_aux", L.ptr(obsv), To, L.ptr(noise), L.ptr(S), L.ptr(hT), L.ptr(cT), L.ptr(enc_w),
           L.ptr(dec_w), B, n_next, L.ptr(pred4), None, None, L.ptr(gsave),
           L.ptr(ade[0]) if ade else None, float(ade[1]) if ade else 0.0, L.ptr(ade[2]) if ade else None,
           L.ptr(d_obs[0]) if d_obs else None, L.ptr(d_obs[1]) if d_obs else None, st)
    if not save:
        return pred4, None
    ctx = GenCtx()
    ctx.obsv, ctx.noise, ctx.scenes, ctx.hT, ctx.cT, ctx.S, ctx.attn = obsv, noise, scenes, hT, cT, S, attn
    ctx.gsave, ctx.B, ctx.To, ctx.Tp, ctx.use_social = gsave, B, To, n_next, use_social
    ctx.wh, ctx.ml = wh, ml
    return pred4, ctx


DFUSE = True     # generator-phase D pass inside the decode BPTT launch (a test compares it with the two launches)


def gen_backward(enc_w, emb_w, att_w, dec_w, ctx, dpred4, d_enc, d_emb, d_att, d_dec, ws=None, tag="g", aux=None, adam=None,
                 dfuse=None):
    """Backward of predict(): decode BPTT -> social block -> obs BPTT -> ONE grouped weight-gradient GEMM launch
    (the social block's problems ride in it).  d_* are the packed gradient buffers (overwritten).
    aux = (src, dst, mask): masked copy dst = mask > 0 ? src : dst done by idle workgroups of the decode BPTT
    launch.  (Running part of the weight GEMMs on a side stream under the BPTT was measured slower: it takes
    CUs from the latency-bound chain and every cross-stream edge of a captured graph costs 5-10 us.)
    adam = (w_all, g_all, m, v, step scalar, lr, beta1, beta2, eps): the kernels that finish the gradients also apply
    the generator's Adam update (sw_gen_wgrad_adam; the four weight / gradient buffers are views of w_all / g_all)."""
    dev = ctx.obsv.device
    ws = ws or default_ws(dev)
    B, To, Tp = ctx.B, ctx.To, ctx.Tp
    gdelta = ws.get(tag + ".gdelta", L.workspace_floats(L.WS_GDELTA, B, To, Tp))
    wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
    tmp = ws.get(tag + ".dwx", 2048)
    dhT = torch.empty(B, 64, device=dev)
    dcT = torch.empty(B, 64, device=dev)
    dS = torch.empty(B, 64, device=dev)
    aux_late = None
    if dfuse is not None:
        # dfuse = (d_w, pred_hat, targets, t_idx, z, g_label, g_code, loss_part): the generator-phase D pass (disc_dpred) runs
        # inside the decode BPTT launch, tile by tile; that launch READS D's weights, so the masked copy that restores them
        # (aux) moves to the observation BPTT launch
        d_w, pred_hat, targets, t_idx, z, g_label, g_code, loss_part = dfuse
        dscr = ws.get(tag + ".dpred", B * Tp * 4)
        L.call("sw_dec_rollout_bwd_dfuse", L.ptr(ctx.obsv), L.ptr(pred_hat), L.ptr(d_w), L.ptr(targets), int(t_idx), L.ptr(z),
               g_label, g_code, L.ptr(loss_part), L.ptr(dscr), L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave), B, To, Tp,
               L.ptr(gdelta), L.ptr(dhT), L.ptr(dcT), L.ptr(dS), L.stream())
        aux_late = aux
    else:
        dpred4 = dpred4.contiguous()
        L.call("sw_dec_rollout_bwd_aux", L.ptr(dpred4), L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave), B, To, Tp,
               L.ptr(gdelta), L.ptr(dhT), L.ptr(dcT), L.ptr(dS), L.ptr(aux[0]) if aux else None, L.ptr(aux[1]) if aux else None,
               L.ptr(aux[2]) if aux else None, aux[1].numel() if aux else 0, L.stream())
    pending = None
    if ctx.use_social and (ctx.scenes.P > 0 or ctx.scenes.NB > 0):
        sc = ctx.scenes
        pending = ws.wgrad_batch
        pws = ws.get("pairs", L.workspace_floats(L.WS_PAIRS, B, To, Tp, 1, sc.P))
        bigp = ws.get("bigpart", sc.big_rows * 132) if sc.NB else None      # scenes above AMAX agents: per-block partial rows
        L.call("sw_social_pool_bwd", L.ptr(ctx.obsv), To, L.ptr(ctx.hT), L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S,
               B, sc.amax, sc.P, L.ptr(emb_w), L.ptr(att_w), L.ptr(ctx.attn), L.ptr(dS), L.ptr(dhT), L.ptr(d_emb),
               L.ptr(d_att), L.ptr(pws), L.ptr(wgrad), L.ptr(sc.big_blocks), sc.NB, L.ptr(ctx.wh), L.ptr(ctx.ml),
               L.ptr(ctx.S), L.ptr(bigp), pending, L.stream())
    else:
        d_emb.zero_()
        d_att.zero_()
    L.call("sw_enc_lstm_bwd_aux", L.ptr(enc_w), L.ptr(ctx.gsave), None, L.ptr(dhT), L.ptr(dcT), None, B, To, 0,
           L.ptr(gdelta), None, None, L.ptr(aux_late[0]) if aux_late else None, L.ptr(aux_late[1]) if aux_late else None,
           L.ptr(aux_late[2]) if aux_late else None, aux_late[1].numel() if aux_late else 0, L.stream())
    if adam is not None:
        w_all, g_all, m, v, step, lr, b1, b2, eps = adam
        L.call("sw_gen_wgrad_adam", L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave), L.ptr(gdelta), L.ptr(ctx.noise), L.ptr(ctx.S),
               B, To, Tp, L.ptr(d_enc), L.ptr(d_dec), L.ptr(wgrad), L.ptr(tmp), pending, L.ptr(w_all), L.ptr(m), L.ptr(v),
               L.ptr(g_all), w_all.numel(), L.ptr(step), lr, b1, b2, eps, L.stream())
        return
    L.call("sw_gen_wgrad", L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave), L.ptr(gdelta), L.ptr(ctx.noise), L.ptr(ctx.S), B, To, Tp,
           L.ptr(d_enc), L.ptr(d_dec), 0, L.ptr(wgrad), L.ptr(tmp), pending, L.stream())


class GenCtxK:
    """Context of K rollouts that share the observation encoding (gen_forward_k)."""
    __slots__ = ("one", "K", "obsv_k", "noise_k", "S_k", "gsave_k")


def gen_forward_k(enc_w, emb_w, att_w, dec_w, obsv, noise_k, scenes, n_next, use_social, K, ws, tag="gv"):
    """K rollouts of predict() on the SAME observations with K noise draws (the best-of-K variety term, train.py:527-536
    with its intended semantics): EncoderLstm over the observed steps and the social pooling do not depend on z, so they run
    ONCE on the B agents; only the decode loop runs on the K*B copies (copy k = rows [k*B, (k+1)*B)).
    Returns pred_hat_4d (K*B, n_next, 4) and the context for gen_backward_k."""
    L.require_gpu(obsv)
    obsv = obsv.contiguous()
    noise_k = noise_k.contiguous()
    B, To = obsv.shape[0], obsv.shape[1]
    KB = K * B
    if noise_k.shape != (KB, 32):
        raise ValueError("noise must be (K * B, 32)")
    dev = obsv.device
    st = L.stream()
    hT, cT = torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev)
    gsave_1 = ws.get(tag + ".gsave1", L.workspace_floats(L.WS_GSAVE, B, To, n_next))
    gsave_k = ws.get(tag + ".gsave", L.workspace_floats(L.WS_GSAVE, KB, To, n_next))
    Ta = To + n_next - 1
    L.call("sw_enc_lstm_fwd_aux", L.ptr(obsv), 0, L.ptr(enc_w), None, None, B, To, L.ptr(hT), L.ptr(cT), None,
           L.ptr(gsave_1), gsave_1.data_ptr() + 4 * Ta * B * 384, 0, None, None, 0, st)
    attn = wh = ml = None
    if use_social:
        S = torch.empty(B, 64, device=dev)
        attn = torch.empty(B, L.AMAX, device=dev)
        if scenes.NB:
            wh, ml = torch.empty(B * 132, device=dev), torch.empty(B, 2, device=dev)
        L.call("sw_social_pool_fwd_aux", L.ptr(obsv), To, L.ptr(hT), L.ptr(scenes.scene_off), scenes.S, B, scenes.amax,
               L.ptr(emb_w), L.ptr(att_w), L.ptr(S), L.ptr(attn), L.ptr(scenes.big_blocks), scenes.NB, L.ptr(wh), L.ptr(ml),
               None, None, 0, st)
    else:
        S = torch.zeros(B, 64, device=dev)
    # the K copies: state / pooled context / last observed point of every agent, and the saved LSTM row of the last observed
    # step in the K*B layout (the decode BPTT and the weight-gradient rows of decode step 0 read c / h of step To - 1 there)
    obsv_k, hT_k, cT_k, S_k = obsv.repeat(K, 1, 1), hT.repeat(K, 1), cT.repeat(K, 1), S.repeat(K, 1)
    row = gsave_1[(To - 1) * B * 384: To * B * 384].view(1, B, 384)
    gsave_k[(To - 1) * KB * 384: To * KB * 384].view(K, B, 384).copy_(row.expand(K, B, 384))
    pred4 = torch.empty(KB, n_next, 4, device=dev)
    L.call("sw_dec_rollout_fwd_aux", L.ptr(obsv_k), To, L.ptr(noise_k), L.ptr(S_k), L.ptr(hT_k), L.ptr(cT_k), L.ptr(enc_w),
           L.ptr(dec_w), KB, n_next, L.ptr(pred4), None, None, L.ptr(gsave_k), None, 0.0, None, None, None, st)
    one = GenCtx()
    one.obsv, one.noise, one.scenes, one.hT, one.cT, one.S, one.attn = obsv, noise_k[:B], scenes, hT, cT, S, attn
    one.gsave, one.B, one.To, one.Tp, one.use_social, one.wh, one.ml = gsave_1, B, To, n_next, use_social, wh, ml
    ctx = GenCtxK()
    ctx.one, ctx.K, ctx.obsv_k, ctx.noise_k, ctx.S_k, ctx.gsave_k = one, K, obsv_k, noise_k, S_k, gsave_k
    return pred4, ctx


def gen_backward_k(enc_w, emb_w, att_w, dec_w, ctx, dpred4_k, d_enc, d_emb, d_att, d_dec, ws, tag="gv", aux=None):
    """Backward of gen_forward_k: decode BPTT on the K*B copies, their gradients w.r.t. the shared state / pooled context
    summed over the copies (back-propagation is linear in the upstream gradient), then the social block and the observation
    BPTT ONCE on the B agents.  Weight gradients: the decode phase's problems over K*B rows (sw_gen_wgrad part 1), the
    observation phase's over B rows on top (part 2)."""
    one, K = ctx.one, ctx.K
    dev = dpred4_k.device
    B, To, Tp = one.B, one.To, one.Tp
    KB = K * B
    dpred4_k = dpred4_k.contiguous()
    gdelta_k = ws.get(tag + ".gdelta", L.workspace_floats(L.WS_GDELTA, KB, To, Tp))
    gdelta_1 = ws.get(tag + ".gdelta1", L.workspace_floats(L.WS_GDELTA, B, To, Tp))
    wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
    tmp = ws.get(tag + ".dwx", 2048)
    dh_k, dc_k, dS_k = (torch.empty(KB, 64, device=dev) for _ in range(3))
    L.call("sw_dec_rollout_bwd_aux", L.ptr(dpred4_k), L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave_k), KB, To, Tp,
           L.ptr(gdelta_k), L.ptr(dh_k), L.ptr(dc_k), L.ptr(dS_k), L.ptr(aux[0]) if aux else None,
           L.ptr(aux[1]) if aux else None, L.ptr(aux[2]) if aux else None, aux[1].numel() if aux else 0, L.stream())
    dhT, dcT, dS = (t.view(K, B, 64).sum(0) for t in (dh_k, dc_k, dS_k))
    L.call("sw_gen_wgrad", L.ptr(enc_w), L.ptr(dec_w), L.ptr(ctx.gsave_k), L.ptr(gdelta_k), L.ptr(ctx.noise_k), L.ptr(ctx.S_k),
           KB, To, Tp, L.ptr(d_enc), L.ptr(d_dec), 1, L.ptr(wgrad), L.ptr(tmp), None, L.stream())
    pending = None
    if one.use_social and (one.scenes.P > 0 or one.scenes.NB > 0):
        sc = one.scenes
        pending = ws.wgrad_batch
        pws = ws.get("pairs", L.workspace_floats(L.WS_PAIRS, B, To, Tp, 1, sc.P))
        bigp = ws.get("bigpart", sc.big_rows * 132) if sc.NB else None
        L.call("sw_social_pool_bwd", L.ptr(one.obsv), To, L.ptr(one.hT), L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S,
               B, sc.amax, sc.P, L.ptr(emb_w), L.ptr(att_w), L.ptr(one.attn), L.ptr(dS), L.ptr(dhT), L.ptr(d_emb),
               L.ptr(d_att), L.ptr(pws), L.ptr(wgrad), L.ptr(sc.big_blocks), sc.NB, L.ptr(one.wh), L.ptr(one.ml),
               L.ptr(one.S), L.ptr(bigp), pending, L.stream())
    else:
        d_emb.zero_()
        d_att.zero_()
    L.call("sw_enc_lstm_bwd", L.ptr(enc_w), L.ptr(one.gsave), None, L.ptr(dhT), L.ptr(dcT), None, B, To, 0,
           L.ptr(gdelta_1), None, None, L.stream())
    L.call("sw_gen_wgrad", L.ptr(enc_w), L.ptr(dec_w), L.ptr(one.gsave), L.ptr(gdelta_1), L.ptr(one.noise), L.ptr(one.S), B, To, Tp,
           L.ptr(d_enc), L.ptr(d_dec), 2, L.ptr(wgrad), L.ptr(tmp), pending, L.stream())


class DiscCtx:
    __slots__ = ("dsave", "B", "To", "Tp", "nb")


D_OBS_MAX_TILES = 128      # the decode launch has idle CUs for the D observation LSTM up to this many 16-agent tiles


def d_obs_buffer(ws, B, To, Tp, nb=2, tag="d"):
    """The save buffer disc_forward(tag, nb branches) will use, or None when the decode launch has no idle CUs to
    precompute the observation LSTM in (gen_forward(d_obs=...) / disc_forward(save_lstm=2))."""
    if (B + 15) // 16 > D_OBS_MAX_TILES:
        return None
    return ws.get(tag + ".dsave", L.workspace_floats(L.WS_DSAVE, B, To, Tp, nb))


def disc_forward(d_w, obsv, preds, save, ws=None, tag="d", save_lstm=True, w_snapshot=None):
    """Discriminator.forward for 1 or 2 future branches sharing the observation encoding.
    Returns ([label_k (B,1)], [code_k (B,2)], ctx).  save_lstm=2: the LSTM rows are already in the save buffer
    (gen_forward(d_obs=...))."""
    L.require_gpu(obsv)
    obsv = obsv.contiguous()
    preds = [p.contiguous() for p in preds]
    B, To = obsv.shape[0], obsv.shape[1]
    x_mode = {2: 0, 4: 1}[obsv.shape[2]]        # positions (B,To,2) or obsv_4d (B,To,4)
    Tp = preds[0].shape[1]
    nb = len(preds)
    dev = obsv.device
    labels = [torch.empty(B, 1, device=dev) for _ in preds]
    codes = [torch.empty(B, 2, device=dev) for _ in preds]
    dsave = None
    if save:
        nfl = L.workspace_floats(L.WS_DSAVE, B, To, Tp, nb)
        dsave = ws.get(tag + ".dsave", nfl) if ws is not None else torch.empty(nfl, device=dev)
    pp, _k1 = L.ptr_array(preds)
    lp, _k2 = L.ptr_array(labels)
    cp, _k3 = L.ptr_array(codes)
    L.call("sw_disc_fwd", L.ptr(obsv), To, x_mode, pp, nb, L.ptr(d_w), B, Tp, lp, cp, L.ptr(dsave), int(save_lstm),
           L.ptr(w_snapshot), L.stream())
    if not save:
        return labels, codes, None
    ctx = DiscCtx()
    ctx.dsave, ctx.B, ctx.To, ctx.Tp, ctx.nb = dsave, B, To, Tp, nb
    return labels, codes, ctx


def disc_backward(d_w, ctx, dlabels, dcodes, d_d_w=None, want_dpred=(), ws=None, tag="d"):
    """Backward of Discriminator.forward.  d_d_w (packed, overwritten) None = no weight gradients;
    want_dpred[k] True = return d loss / d pred4 of branch k."""
    dev = dlabels[0].device
    ws = ws or default_ws(dev)
    B, To, Tp, nb = ctx.B, ctx.To, ctx.Tp, ctx.nb
    dlabels = [t.contiguous() for t in dlabels]
    dcodes = [t.contiguous() for t in dcodes]
    want = list(want_dpred) + [False] * (nb - len(want_dpred))
    dpreds = [torch.empty(B, Tp, 4, device=dev) if w else None for w in want]
    ddelta = wgrad = None
    if d_d_w is not None:
        ddelta = ws.get(tag + ".ddelta", L.workspace_floats(L.WS_DDELTA, B, To, Tp, nb))
        wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
    lp, _k1 = L.ptr_array(dlabels)
    cp, _k2 = L.ptr_array(dcodes)
    dp, _k3 = L.ptr_array(dpreds)
    L.call("sw_disc_bwd", L.ptr(d_w), L.ptr(ctx.dsave), lp, cp, nb, B, To, Tp, L.ptr(ddelta), L.ptr(d_d_w), dp,
           L.ptr(wgrad), L.stream())
    return dpreds


def disc_dpred(d_w, obsv, pred_hat, targets, t_idx, z, g_label, g_code, loss_part=None):
    """Generator phase: D(obsv, pred_hat) forward + the backward of its prediction heads in one launch;
    returns d(g_loss)/d(pred_hat) (B,Tp,4)."""
    L.require_gpu(obsv)
    obsv, pred_hat = obsv.contiguous(), pred_hat.contiguous()
    B, To, Tp = obsv.shape[0], obsv.shape[1], pred_hat.shape[1]
    x_mode = {2: 0, 4: 1}[obsv.shape[2]]
    dpred = torch.empty(B, Tp, 4, device=obsv.device)
    L.call("sw_disc_dpred", L.ptr(obsv), To, x_mode, L.ptr(pred_hat), L.ptr(d_w), B, Tp, L.ptr(targets), int(t_idx), L.ptr(z),
           g_label, g_code, L.ptr(dpred), None, None, L.ptr(loss_part), L.stream())
    return dpred


def disc_backward_gan(d_w, ctx, labels, codes, targets, t_idx, z, g_label, g_code, d_d_w=None, want_dpred=(), ws=None,
                      tag="d", loss_part=None, adam=None):
    """disc_backward with the LSGAN / InfoGAN loss gradients formed inside the kernel from the forward
    outputs `labels` / `codes` (targets = device [2] label-noise scalars, t_idx = target index per branch).
    loss_part (ceil(B/16),3): receives the per-tile sums of the squared errors (the reported MSE terms)."""
    dev = labels[0].device
    ws = ws or default_ws(dev)
    B, To, Tp, nb = ctx.B, ctx.To, ctx.Tp, ctx.nb
    want = list(want_dpred) + [False] * (nb - len(want_dpred))
    dpreds = [torch.empty(B, Tp, 4, device=dev) if w else None for w in want]
    ddelta = wgrad = None
    if d_d_w is not None:
        ddelta = ws.get(tag + ".ddelta", L.workspace_floats(L.WS_DDELTA, B, To, Tp, nb))
        wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
    lp, _k1 = L.ptr_array(labels)
    cp, _k2 = L.ptr_array(codes)
    dp, _k3 = L.ptr_array(dpreds)
    t0, t1 = (list(t_idx) + [0])[:2]
    if adam is not None:      # (m, v, step scalar, lr, beta1, beta2, eps): the update rides in the gradient reduction
        m, v, step, lr, b1, b2, eps = adam
        L.call("sw_disc_bwd_gan_adam", L.ptr(d_w), L.ptr(ctx.dsave), lp, cp, L.ptr(targets), t0, t1, L.ptr(z), g_label, g_code,
               nb, B, To, Tp, L.ptr(ddelta), L.ptr(d_d_w), dp, L.ptr(wgrad), L.ptr(loss_part), L.ptr(d_w), L.ptr(m), L.ptr(v),
               L.ptr(step), float(lr), float(b1), float(b2), float(eps), L.stream())
        return dpreds
    L.call("sw_disc_bwd_gan", L.ptr(d_w), L.ptr(ctx.dsave), lp, cp, L.ptr(targets), t0, t1, L.ptr(z), g_label, g_code, nb,
           B, To, Tp, L.ptr(ddelta), L.ptr(d_d_w), dp, L.ptr(wgrad), L.ptr(loss_part), L.stream())
    return dpreds


def disc_update_supported(d_w, B, To, Tp):
    """Can one discriminator update pass run as ONE launch (sw_disc_update: shapes that leave CUs idle, registered images)?"""
    return bool(L.load().sw_disc_update_supported(L.ptr(d_w), int(B), int(To), int(Tp)))


def disc_update(d_w, obsv, preds, targets, t_idx, z, g_label, g_code, d_d_w, ws, tag="d", obs_pre=False, w_snapshot=None,
                loss_part=None, adam=None):
    """disc_forward(obsv, [fake, real]) + disc_backward_gan(...) of one D update (train.py:476-495) as one launch
    (sw_disc_update) + the weight-gradient GEMM (+ the Adam update).  Returns (labels, codes)."""
    L.require_gpu(obsv)
    obsv = obsv.contiguous()
    preds = [p.contiguous() for p in preds]
    B, To, Tp = obsv.shape[0], obsv.shape[1], preds[0].shape[1]
    dev = obsv.device
    labels = [torch.empty(B, 1, device=dev) for _ in preds]
    codes = [torch.empty(B, 2, device=dev) for _ in preds]
    dsave = ws.get(tag + ".dsave", L.workspace_floats(L.WS_DSAVE, B, To, Tp, 2))
    ddelta = ws.get(tag + ".ddelta", L.workspace_floats(L.WS_DDELTA, B, To, Tp, 2))
    wgrad = ws.get("wgrad", L.workspace_floats(L.WS_WGRAD, B, To, Tp))
    pp, _k1 = L.ptr_array(preds)
    lp, _k2 = L.ptr_array(labels)
    cp, _k3 = L.ptr_array(codes)
    m = v = step = None
    lr = b1 = b2 = eps = 0.0
    if adam is not None:
        m, v, step, lr, b1, b2, eps = adam
    L.call("sw_disc_update", L.ptr(obsv), To, pp, L.ptr(d_w), B, Tp, lp, cp, L.ptr(dsave), int(bool(obs_pre)), L.ptr(w_snapshot),
           L.ptr(targets), int(t_idx[0]), int(t_idx[1]), L.ptr(z), g_label, g_code, L.ptr(ddelta), L.ptr(d_d_w), L.ptr(wgrad),
           L.ptr(loss_part), L.ptr(d_w) if adam is not None else None, L.ptr(m), L.ptr(v), L.ptr(step), float(lr), float(b1),
           float(b2), float(eps), L.stream())
    return labels, codes
