"""The GENERIC-WIDTH path: the reference's model for any `--hidden-size` and latent-code count (train.py:42-44, 65, 76-81).

The fused kernels behind `model.py` keep one 64-unit layer per workgroup in registers; every BASELINE config uses 64
units, smaller sizes run on them zero-padded.  Sizes ABOVE 64 (W_hh alone is 256 KB at 128 units) and latent-code
counts other than 2 run here instead: the same modules (same names, constructor order - hence the same initial weights
for a seed -, state_dict keys and shapes as train.py:153-335), evaluated LAYER BY LAYER through the C ABI -

    matrix products         sw_rows_gemm (y = x W^T + b, dx = dy W), sw_linear_wgrad (dW = dy^T x, db)
    LSTM cell               sw_lstm_point_fwd / _bwd            (nn.LSTM gate order, train.py:254, 278)
    ReLU / LeakyReLU(0.2)   sw_act_fwd / _bwd
    SocialFeatures          sw_pair_features                    (in-scene pairs only, train.py:208-241)
    AttentionPooling        sw_attn_pairs_fwd / _bwd            (train.py:153-175)
    nn.MSELoss              sw_sqdiff                           (train.py:484-494, 512-523)
    get_traj_4d, ADE / FDE  sw_traj_4d, sw_ade_fde

- each wrapped in a torch.autograd.Function, so torch's tape does the bookkeeping of the backward pass (as it does for the
stand-alone sub-modules of model.py) and torch.optim.Adam the update (north_star: "Host code stays Python on
PyTorch-ROCm for glue and the Adam step").  The matrix products, LSTM cells, activations, pair features, attention and loss
terms run in the library's kernels; torch supplies memory, concatenation / slicing, the tape - and a few element-wise
operations of its own: the sum of the two gate pre-activation products, the position integration p += v, the scaling of
the loss gradients, the loss sum and both Adam steps.  This path is launch-bound (over a thousand small launches per step
under the tape): 42 steps/s at 128 units on the metric shape.  Widths that are multiples of 32 train on the WIDE path instead
(wide.py: the same modules, explicit backward over time-step-level kernels, one hipGraph per step - 10 x faster); this file
remains for every other width the reference accepts (data parallel and with the L2 / variety terms like the other engines),
as the module classes of both, and as the cross-check of the wide engine (tests).  No CPU fallback: CPU tensors raise.
"""
import copy

import torch
import torch.nn as nn

from . import _lib as L
from .model import _scene_index, _wgrad_ws, get_traj_4d
from .trainer import SocialWaysTrainer


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Lin(torch.autograd.Function):
    """y = x[:, :K] W^T + b for W (N, K) (nn.Linear, train.py:158, 183-185, 251, 278-292, 324-328); x may carry padding
    columns beyond K (the pair features are stored 4 wide)."""

    @staticmethod
    def forward(ctx, x, W, b):
        L.require_gpu(x)
        x, W = _c(x), _c(W)
        R, ldx = x.shape[0], x.shape[1]
        N, K = W.shape
        y = torch.empty(R, N, device=x.device)
        if R:
            L.call("sw_rows_gemm", L.ptr(x), ldx, L.ptr(W), 1, K, L.ptr(b), R, K, N, L.ptr(y), N, 0, L.stream())
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _c(dy)
        R, ldx = x.shape[0], x.shape[1]
        N, K = W.shape
        dev = x.device
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros(R, ldx, device=dev) if ldx != K else torch.empty(R, K, device=dev)
            if R:
                L.call("sw_rows_gemm", L.ptr(dy), N, L.ptr(W), K, 1, None, R, N, K, L.ptr(dx), ldx, 0, L.stream())
        if ctx.needs_input_grad[1]:
            # the grouped GEMM reads delta / activation rows as float4s and cuts them into whole lane vectors: row strides and
            # the contracted / output widths are padded to multiples of 4 with zeros (widths like 50 = 0.625 x 80 units)
            N4, K4 = (N + 3) // 4 * 4, (K + 3) // 4 * 4
            dWp = torch.zeros(N4, K4, device=dev)
            dbp = torch.zeros(N4, device=dev)
            if R:
                d4 = dy
                if N4 != N:
                    d4 = torch.zeros(R, N4, device=dev)
                    d4[:, :N] = dy
                x4 = x
                if ldx % 4 or K4 > ldx:
                    x4 = torch.zeros(R, max(K4, (ldx + 3) // 4 * 4), device=dev)
                    x4[:, :K] = x[:, :K]
                ws = _wgrad_ws(dev)
                for n0 in range(0, N4, 256):       # sw_linear_wgrad takes up to 256 output rows per call
                    n1 = min(N4, n0 + 256)
                    L.call("sw_linear_wgrad", d4.data_ptr() + 4 * n0, d4.shape[1], L.ptr(x4), x4.shape[1], R, n1 - n0, K4,
                           dWp.data_ptr() + 4 * n0 * K4, K4, dbp.data_ptr() + 4 * n0, L.ptr(ws), 0, L.stream())
            dW = dWp[:N, :K].contiguous() if (N4 != N or K4 != K) else dWp
            db = dbp[:N].contiguous() if N4 != N else dbp
            if not ctx.has_b:
                db = None
        return dx, dW, db


class _Act(torch.autograd.Function):
    """kind 0: ReLU (train.py:183-185), 1: LeakyReLU(0.2) (train.py:280-292, 324-328)."""

    @staticmethod
    def forward(ctx, x, kind):
        x = _c(x)
        y = torch.empty_like(x)
        L.call("sw_act_fwd", L.ptr(x), x.numel(), kind, L.ptr(y), L.stream())
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        L.call("sw_act_bwd", L.ptr(y), L.ptr(dy), y.numel(), ctx.kind, L.ptr(dx), L.stream())
        return dx, None


class _LstmCell(torch.autograd.Function):
    """The element-wise part of an nn.LSTM step: pre (B, 4H) gate pre-activations (i | f | g | o), c_prev (B, H) ->
    (h, c).  c' = f c + i g, h' = o tanh(c')."""

    @staticmethod
    def forward(ctx, pre, c_prev):
        pre, c_prev = _c(pre), _c(c_prev)
        B, H = c_prev.shape
        gates, c, h = torch.empty_like(pre), torch.empty_like(c_prev), torch.empty_like(c_prev)
        L.call("sw_lstm_point_fwd", L.ptr(pre), L.ptr(c_prev), B, H, L.ptr(gates), L.ptr(c), L.ptr(h), L.stream())
        ctx.save_for_backward(gates, c, c_prev)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c, c_prev = ctx.saved_tensors
        B, H = c.shape
        dpre, dcp = torch.empty_like(gates), torch.empty_like(c)
        L.call("sw_lstm_point_bwd", L.ptr(gates), L.ptr(c), L.ptr(c_prev), L.ptr(None if dh is None else _c(dh)),
               L.ptr(None if dc is None else _c(dc)), B, H, L.ptr(dpre), L.ptr(dcp), L.stream())
        return dpre, dcp


class _AttnPairs(torch.autograd.Function):
    """AttentionPooling (train.py:160-174) on the embedded pair rows f (P, F): S_i = sum_j softmax_j(<f_ij, wh_j>) h_j."""

    @staticmethod
    def forward(ctx, f, wh, h, sc):
        f, wh, h = _c(f), _c(wh), _c(h)
        B, H = h.shape
        F = wh.shape[1]
        attn = torch.empty(max(sc.P, 1), device=h.device)
        S = torch.empty(B, H, device=h.device)
        L.call("sw_attn_pairs_fwd", L.ptr(f), L.ptr(wh), L.ptr(h), L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S, B, F, H,
               L.ptr(attn), L.ptr(S), L.stream())
        ctx.save_for_backward(f, wh, h, attn)
        ctx.sc = sc
        return S

    @staticmethod
    def backward(ctx, dS):
        f, wh, h, attn = ctx.saved_tensors
        sc = ctx.sc
        B, H = h.shape
        F = wh.shape[1]
        dev = h.device
        dsig = torch.empty_like(attn)
        df = torch.zeros_like(f)
        dwh, dh = torch.empty(B, F, device=dev), torch.empty(B, H, device=dev)
        L.call("sw_attn_pairs_bwd", L.ptr(f), L.ptr(wh), L.ptr(h), L.ptr(attn), L.ptr(_c(dS)), L.ptr(sc.scene_off),
               L.ptr(sc.pair_off), sc.S, B, F, H, L.ptr(dsig), L.ptr(df), L.ptr(dwh), L.ptr(dh), L.stream())
        return df, dwh, dh, None


class _Mse(torch.autograd.Function):
    """nn.MSELoss()(a, b) for 2-d blocks (row strides free, unit column stride); b = a tensor of a's shape or a
    (targets, index) pair for a scalar target kept on the device.  Returns (mean, sum of squares)."""

    @staticmethod
    def forward(ctx, a, b, targets, t_idx):
        assert a.dim() == 2 and a.stride(1) == 1 and (b is None or (b.shape == a.shape and b.stride(1) == 1))
        R, C = a.shape
        out = torch.empty(1, device=a.device)
        L.call("sw_sqdiff", L.ptr_strided(a), a.stride(0), L.ptr_strided(b), 0 if b is None else b.stride(0),
               L.ptr(targets), int(t_idx), R, C, 0.0, L.ptr(out), None, 0, L.stream())
        ctx.save_for_backward(a, b if b is not None else a, targets if targets is not None else a)
        ctx.meta = (b is None, int(t_idx))
        ssum = out[0].clone()
        ctx.mark_non_differentiable(ssum)
        return out[0] / float(R * C), ssum

    @staticmethod
    def backward(ctx, g, _gsum):
        a, b, targets = ctx.saved_tensors
        scalar, t_idx = ctx.meta
        R, C = a.shape
        da = torch.empty(R, C, device=a.device)
        L.call("sw_sqdiff", L.ptr_strided(a), a.stride(0), None if scalar else L.ptr_strided(b), 0 if scalar else b.stride(0),
               L.ptr(targets) if scalar else None, t_idx, R, C, 2.0 / float(R * C), None, L.ptr(da), C, L.stream())
        return da * g, None, None, None


def _lin(layer, x):
    return _Lin.apply(x, layer.weight, layer.bias)


def _mlp(seq, x):
    """nn.Sequential of Linear / ReLU / LeakyReLU(0.2) layers through the kernels."""
    for m in seq:
        if isinstance(m, nn.Linear):
            x = _lin(m, x)
        elif isinstance(m, nn.ReLU):
            x = _Act.apply(x, 0)
        elif isinstance(m, nn.LeakyReLU):
            assert abs(m.negative_slope - 0.2) < 1e-12
            x = _Act.apply(x, 1)
        else:
            raise TypeError(type(m))
    return x


def _lstm_step(lstm, x, h, c, layer=0):
    """One step of layer `layer` of an nn.LSTM on x (B, in): the two gate products + the cell."""
    w_ih, w_hh = getattr(lstm, "weight_ih_l%d" % layer), getattr(lstm, "weight_hh_l%d" % layer)
    b_ih, b_hh = getattr(lstm, "bias_ih_l%d" % layer), getattr(lstm, "bias_hh_l%d" % layer)
    pre = _Lin.apply(x, w_ih, b_ih) + _Lin.apply(h, w_hh, b_hh)
    return _LstmCell.apply(pre, c)


# ---- the reference's modules as parameter containers (same construction order = same initial weights) ------------------
class EncoderLstm(nn.Module):                               # train.py:245-269
    """Any width, any number of stacked layers (the class default is 2, train.py:246; the script builds 1, train.py:82)."""

    def __init__(self, hidden_size, n_layers=2, device=None):
        super().__init__()
        if n_layers < 1:
            raise L.SocialWaysHipError("EncoderLstm: n_layers >= 1")
        self.hidden_size, self.n_layers = hidden_size, n_layers
        self.embed = nn.Linear(4, hidden_size)
        self.lstm = nn.LSTM(hidden_size, hidden_size, num_layers=n_layers, batch_first=True)
        self.lstm_h = []
        if device is not None:
            self.to(device)

    def init_lstm(self, h, c):
        self.lstm_h = (h, c)

    def step(self, x4, h, c):
        """One step of a one-layer encoder on states (B, H) (the generator's rollout)."""
        return _lstm_step(self.lstm, _lin(self.embed, x4), h, c)

    def forward(self, obsv):
        """obsv (B,T,4) or (B,4): embed + the stacked LSTM from the stored state (n_layers, B, H); returns the top layer's
        y (B,T,H) and keeps the new state in `self.lstm_h`, like train.py:262-269."""
        L.require_gpu(obsv)
        bs, H = obsv.shape[0], self.hidden_size
        x = obsv.reshape(bs, -1, 4)
        T = x.shape[1]
        e = _lin(self.embed, x.reshape(bs * T, 4)).view(bs, T, H)
        h = [self.lstm_h[0][k].reshape(bs, H) for k in range(self.n_layers)]
        c = [self.lstm_h[1][k].reshape(bs, H) for k in range(self.n_layers)]
        ys = []
        for t in range(T):
            inp = e[:, t]
            for k in range(self.n_layers):
                h[k], c[k] = _lstm_step(self.lstm, inp, h[k], c[k], k)
                inp = h[k]
            ys.append(inp)
        self.lstm_h = (torch.stack(h, 0), torch.stack(c, 0))
        return torch.stack(ys, 1)


class EmbedSocialFeatures(nn.Module):                       # train.py:178-189
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(input_size, 32), nn.ReLU(), nn.Linear(32, 64), nn.ReLU(), nn.Linear(64, hidden_size))


class AttentionPooling(nn.Module):                          # train.py:153-175
    def __init__(self, h_dim, f_dim):
        super().__init__()
        self.W = nn.Linear(h_dim, f_dim, bias=True)


class DecoderFC(nn.Module):                                 # train.py:320-335
    def __init__(self, hidden_dim):
        super().__init__()
        d = hidden_dim
        self.fc1 = nn.Sequential(nn.Linear(d, d), nn.LeakyReLU(0.2), nn.Linear(d, d // 2), nn.LeakyReLU(0.2),
                                 nn.Linear(d // 2, d // 4), nn.Linear(d // 4, 2))


class Discriminator(nn.Module):                             # train.py:272-316
    def __init__(self, n_next, hidden_dim, n_latent_code, device=None):
        super().__init__()
        d = hidden_dim
        self.lstm_dim, self.n_next, self.n_latent_code = d, n_next, n_latent_code
        self.obsv_encoder_lstm = nn.LSTM(4, d, batch_first=True)
        self.obsv_encoder_fc = nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, d // 2))
        self.pred_encoder = nn.Sequential(nn.Linear(n_next * 4, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, d // 2))
        self.classifier = nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, 1))
        self.latent_decoder = nn.Sequential(nn.Linear(d, d // 2), nn.LeakyReLU(0.2), nn.Linear(d // 2, n_latent_code))
        if device is not None:
            self.to(device)

    def forward(self, obsv, pred):
        """obsv (B,To,4), pred (B,Tp,4) -> (label (B,1) raw LSGAN score, code_hat (B,n_latent_code)) (train.py:294-309)."""
        L.require_gpu(obsv)
        B, dev = obsv.shape[0], obsv.device
        h = torch.zeros(B, self.lstm_dim, device=dev)
        c = torch.zeros(B, self.lstm_dim, device=dev)
        for t in range(obsv.shape[1]):
            h, c = _lstm_step(self.obsv_encoder_lstm, obsv[:, t], h, c)
        obsv_code = _mlp(self.obsv_encoder_fc, h)
        pred_code = _mlp(self.pred_encoder, pred.reshape(B, self.n_next * 4))
        both = torch.cat([obsv_code, pred_code], dim=1)
        return _mlp(self.classifier, both), _mlp(self.latent_decoder, both)

    def load(self, backup):
        """Restore nn.Linear weights/biases only; the LSTM keeps its update (train.py:311-316)."""
        for m_from, m_to in zip(backup.modules(), self.modules()):
            if isinstance(m_to, nn.Linear):
                m_to.weight.data.copy_(m_from.weight.data)
                if m_to.bias is not None:
                    m_to.bias.data.copy_(m_from.bias.data)


class Generator(nn.Module):
    """encoder, feature_embedder, attention, decoder in the reference's construction order (train.py:370-375) and
    predict() (train.py:392-432) as forward."""

    def __init__(self, hidden_size, n_lstm_layers=1, use_social=False, device=None):
        super().__init__()
        self.encoder = EncoderLstm(hidden_size, n_lstm_layers)
        self.feature_embedder = EmbedSocialFeatures(3, hidden_size)
        self.attention = AttentionPooling(hidden_size, hidden_size)
        self.decoder = DecoderFC(hidden_size + hidden_size + hidden_size // 2)
        self.use_social, self.hidden_size, self.noise_len = use_social, hidden_size, hidden_size // 2
        if device is not None:
            self.to(device)

    def predictor_params(self):
        from itertools import chain
        return chain(self.attention.parameters(), self.feature_embedder.parameters(), self.encoder.parameters(),
                     self.decoder.parameters())

    def social(self, last4, hT, sc):
        """SocialFeatures -> EmbedSocialFeatures -> AttentionPooling on the in-scene pairs (train.py:408-411)."""
        if sc.NB:
            raise L.SocialWaysHipError("generic-width path: scenes above %d agents are not supported" % L.AMAX)
        B, dev = hT.shape[0], hT.device
        if sc.P == 0:
            return torch.zeros_like(hT)
        feat = torch.empty(sc.P, 4, device=dev)
        L.call("sw_pair_features", L.ptr(_c(last4)), L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S, L.ptr(feat), L.stream())
        f = _mlp(self.feature_embedder.fc, feat)
        return _AttnPairs.apply(f, _lin(self.attention.W, hT), hT, sc)

    def forward(self, obsv_p, noise, n_next, sub_batches=[]):
        L.require_gpu(obsv_p)
        B, dev, H = obsv_p.shape[0], obsv_p.device, self.hidden_size
        sc = _scene_index(sub_batches, B, dev)
        o4 = get_traj_4d(obsv_p, [])
        h, c = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
        for t in range(o4.shape[1]):
            h, c = self.encoder.step(o4[:, t], h, c)
        S = self.social(o4[:, -1], h, sc) if self.use_social else torch.zeros_like(h)
        noise = noise.to(dev)
        last, out = o4[:, -1], []
        for i in range(n_next):
            v = _mlp(self.decoder.fc1, torch.cat([h, S, noise], dim=1))
            p = v + last[:, :2]                              # position integration (train.py:423): memory-level glue
            last = torch.cat([p, v], dim=1)
            out.append(last)
            if i + 1 < n_next:                               # the step after the last decode is dead compute (train.py:430)
                h, c = self.encoder.step(last, h, c)
        return torch.stack(out, dim=1)


# ------------------------------------------------------------------------------------------------------------------------
class GenericTrainer(SocialWaysTrainer):
    """train() / test() / checkpoint (train.py:439-668) for any hidden size / latent-code count on the generic-width
    path.  Same public surface as SocialWaysTrainer (step, step_many, train_epoch, test, checkpoint, load_checkpoint,
    losses_from), data parallel over a process group like it (scene-aligned shards of every packed batch, losses normalised by
    the global batch, the gradients of each of the step's three updates all-reduced before their Adam step)."""

    def __init__(self, n_next, hidden_size=64, lr_g=1e-4, lr_d=1e-3, n_unrolling_steps=1, use_social=True, use_info_loss=True,
                 loss_info_w=0.5, n_latent_codes=2, device="cuda", process_group=None, use_l2_loss=False,
                 use_variety_loss=False, loss_l2_w=0.5, **perf_only):
        # options of the fused trainer that change speed, not results, are accepted and have no effect here; anything else
        # is a mistake worth hearing about
        unknown = set(perf_only) - {"variety_k", "use_graph", "fused_adam"}
        if unknown:
            raise TypeError("unexpected keyword arguments: %s" % sorted(unknown))
        if use_variety_loss not in (False, True):
            raise L.SocialWaysHipError("generic-width path: use_variety_loss=%r (the folded best-of-K form) is not implemented; "
                                       "True = the reference's term as written" % (use_variety_loss,))
        if int(n_latent_codes) < 2:
            # train.py:486, 516 compare code_hat.squeeze() of shape (B,) with noise[:, :1] of shape (B, 1): nn.MSELoss
            # broadcasts them to (B, B).  That accident is not reproduced here - and not silently replaced by another loss
            raise L.SocialWaysHipError("n_latent_codes = 1: the reference's info loss broadcasts (B,) against (B, 1) into a "
                                       "(B, B) mean (train.py:486, 516); not supported - use >= 2 latent codes")
        self.device = L.indexed_device(device)
        if self.device.type != "cuda":
            raise L.SocialWaysHipError("socialways_amd runs on MI355X only (no CPU fallback)")
        self.n_next, self.noise_len = n_next, hidden_size // 2
        self.n_unrolling_steps, self.use_info_loss, self.loss_info_w = n_unrolling_steps, use_info_loss, loss_info_w
        self.use_l2_loss, self.use_variety_loss, self.loss_l2_w = use_l2_loss, bool(use_variety_loss), loss_l2_w
        self.n_latent_codes = n_latent_codes
        # construction order = train.py:370-385 (RNG -> init mapping, optimizer parameter order); built on the CPU
        # generator like the reference, then moved
        self.G = Generator(hidden_size, 1, use_social=use_social).to(self.device)
        self.predictor_optimizer = torch.optim.Adam(self.G.predictor_params(), lr=lr_g, betas=(0.9, 0.999))
        self.D = Discriminator(n_next, hidden_size, n_latent_codes).to(self.device)
        self.D_optimizer = torch.optim.Adam(self.D.parameters(), lr=lr_d, betas=(0.9, 0.999))
        self.pg, self.world, self.rank, self.epoch = process_group, 1, 0, 0
        if process_group is not None:      # data parallel like the fused trainer: scene shards, three all-reduces per step
            self.world, self.rank = torch.distributed.get_world_size(process_group), torch.distributed.get_rank(process_group)
        self.use_graph = False
        self.last_variety = None
        if self.world > 1:
            self.sync_replicas()

    @property
    def use_social(self):
        return self.G.use_social

    def step(self, obsv, pred, sub_batches, zeros_val, ones_val, noise, ss=1.0, global_B=None, out=None, global_row0=0,
             variety_noise=None):
        """One packed batch (train.py:458-554).  Returns the (U+3, 3) float64 sums of SocialWaysTrainer.step()."""
        dev, U, nl = self.device, self.n_unrolling_steps, self.n_latent_codes
        G, D = self.G, self.D
        B = obsv.shape[0]
        Bg = float(global_B if global_B is not None else B)
        if self.use_variety_loss and Bg < 20:
            raise ValueError("use_variety_loss indexes agent 19 of the packed batch (train.py:531): batch of %d" % Bg)
        k = B / Bg                       # this shard's share of the batch means (1 in a single process)
        obsv, pred = _c(obsv), _c(pred)
        z = _c(noise.to(dev))
        targets = torch.tensor([float(zeros_val), float(ones_val)], dtype=torch.float32).to(dev)
        o4, p4 = get_traj_4d(obsv, pred)
        res = torch.zeros(U + 3, 3, dtype=torch.float64, device=dev)
        wi = self.loss_info_w if self.use_info_loss else 0.0
        # the three predict() calls of a step are identical: one rollout, its tape serves the G phase
        pred_hat = G(obsv, z, self.n_next, sub_batches)
        fake = pred_hat.detach()
        backup = None
        for u in range(U + 1):                                                 # train.py:476-499
            self.D_optimizer.zero_grad(set_to_none=True)
            fl, code = D(o4, fake)
            l_fake, s_fake = _Mse.apply(fl, None, targets, 0)
            l_info, s_info = _Mse.apply(code, z[:, :nl], None, 0)
            rl, _ = D(o4, p4)
            l_real, s_real = _Mse.apply(rl, None, targets, 1)
            (k * (l_fake + l_real + wi * l_info)).backward()
            self._reduce_grads(self.D_optimizer)
            self.D_optimizer.step()
            res[u, 0], res[u, 1], res[u, 2] = s_fake.double(), s_info.double() * (2.0 / nl), s_real.double()
            if u == 0 and U > 0:
                backup = copy.deepcopy(D)
        self.D_optimizer.zero_grad(set_to_none=True)                           # train.py:503-539
        self.predictor_optimizer.zero_grad(set_to_none=True)
        gl, code = D(o4, pred_hat)
        l_fool, s_fool = _Mse.apply(gl, None, targets, 1)
        l_info, s_info = _Mse.apply(code, z[:, :nl], None, 0)
        g_loss = l_fool + wi * l_info
        if self.use_l2_loss:
            l2, _ = _Mse.apply(pred_hat[:, :, :2].reshape(B, -1), pred.reshape(B, -1), None, 0)
            g_loss = g_loss + self.loss_l2_w * l2
        g_loss = k * g_loss
        if self.use_variety_loss:        # train.py:527-536 as written: only the k = 19 term survives - the L2 of AGENT 19
            r = 19 - int(global_row0)    # of the packed batch (its 20 rollouts are identical); it lives on one rank's shard
            if 0 <= r < B:
                lv, _ = _Mse.apply(pred_hat[r:r + 1, :, :2].reshape(1, -1), pred[r:r + 1].reshape(1, -1), None, 0)
                g_loss = g_loss + self.loss_l2_w * lv
        g_loss.backward()
        self._reduce_grads(self.predictor_optimizer)
        self.predictor_optimizer.step()
        self.D_optimizer.zero_grad(set_to_none=True)
        if backup is not None:
            D.load(backup)                                                     # train.py:541-542
        res[U + 1, 0], res[U + 1, 1] = s_fool.double(), s_info.double() * (2.0 / nl)
        ade = torch.zeros(3, device=dev)
        L.call("sw_ade_fde", L.ptr(fake), L.ptr(pred), B, self.n_next, 1.0 / float(ss), L.ptr(ade),
               L.ptr(torch.empty(3 * L.RED_BLOCKS, device=dev)), L.stream())
        res[U + 2] = ade.double()
        self.last_pred_hat = fake
        return res

    def step_many(self, batches, sub_batches, ss=1.0, global_B=None, out=None, global_row0=0):
        return [self.step(o, p, sub_batches, zv, ov, nz, ss, global_B, out, global_row0) for o, p, zv, ov, nz in batches]

    def release_graphs(self):
        pass

    # ---- data parallelism ------------------------------------------------------------------------------------------------
    def _reduce_grads(self, optim):
        """SUM over the ranks of every gradient of `optim`'s parameters (the losses are already weighted B_r / B) as ONE
        flat all-reduce; a parameter without a gradient on this rank (e.g. the social block on a shard of single-agent
        scenes) contributes zeros - every rank then applies the same update."""
        if self.pg is None or self.world == 1:
            return
        ps = [p for g in optim.param_groups for p in g["params"]]
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in ps]).to(self.device)
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps] + [has])
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        has = flat[-len(ps):].tolist()
        o = 0
        for p, h in zip(ps, has):      # a parameter no rank has a gradient for stays without one (Adam skips it, as in one process)
            p.grad = flat[o:o + p.numel()].view_as(p).clone() if h > 0 else None
            o += p.numel()

    def _empty_step(self):
        """A rank without scenes in this packed batch still takes part in the step's three all-reduces and applies the
        same updates."""
        U = self.n_unrolling_steps
        backup = None
        for u in range(U + 1):
            self.D_optimizer.zero_grad(set_to_none=True)
            self._reduce_grads(self.D_optimizer)
            self.D_optimizer.step()
            if u == 0 and U > 0:
                backup = copy.deepcopy(self.D)
        self.predictor_optimizer.zero_grad(set_to_none=True)
        self._reduce_grads(self.predictor_optimizer)
        self.predictor_optimizer.step()
        if backup is not None:
            self.D.load(backup)
        return torch.zeros(U + 3, 3, dtype=torch.float64, device=self.device)

    def sync_replicas(self):
        """Rank 0's weights and optimizer state to every rank (construction, load_checkpoint)."""
        if self.pg is None or self.world == 1:
            return
        dist = torch.distributed
        src = dist.get_global_rank(self.pg, 0)
        for p in list(self.G.parameters()) + list(self.D.parameters()):
            dist.broadcast(p.data, src, group=self.pg)
        for optim in (self.predictor_optimizer, self.D_optimizer):
            box = [optim.state_dict() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src, group=self.pg,
                                       device=self.device if dist.get_backend(self.pg) == "nccl" else None)
            if self.rank != 0:
                optim.load_state_dict(box[0])
        ep = torch.tensor([float(self.epoch)], device=self.device)
        dist.broadcast(ep, src, group=self.pg)
        self.epoch = int(ep.item())

    def _allreduce(self, flat):
        if self.pg is not None and self.world > 1:
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=self.pg)
