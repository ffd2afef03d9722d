"""The WIDE path: `--hidden-size` above the fused kernels' 64 units (train.py:42-44, 76-81), H % 32 == 0, and any latent-code
count >= 2 (train.py:65), as an explicit forward / backward ENGINE over time-step-level kernels (csrc/sw_wide.hip):

    one launch per LSTM step          sw_wide_lstm_fwd / _bwd   gate products + cell (+ the recurrent dh product) fused
    one launch per decoder / head     sw_wide_gemm              product + bias + LeakyReLU / ReLU forward, product x
    layer                                                       activation derivative on the way back
    weight gradients                  sw_wide_wgrad             every matrix of a backward pass in one or two launches of
                                                                the grouped split-K GEMM over the time-major saved rows
    Adam                              sw_adam_packed            one launch per optimizer over a packed parameter buffer

- no autograd tape: the backward pass is written out like the fused trainer's (same saved-row layout idea: time-major
[t][agent][...] rows, deferred weight gradients), and the whole training step (~400 launches) is captured into ONE hipGraph
per batch layout and replayed.  Same modules, construction order (= initial weights), state_dict keys and optimizer
state_dicts as train.py:153-335, 370-385; results match the layer-by-layer generic path (generic.py, kept for widths that
are not multiples of 32 and as the cross-check of this engine) and the oracle to the usual tolerances.

torch is used for memory (buffers, slicing, broadcast copies of S / z into the decoder's input rows, weight transposes, the
Linear-only restore of train.py:541-542); arithmetic runs in the library's kernels.  No CPU fallback.
"""
import ctypes
import struct
import os

import numpy as np
import torch

from . import _lib as L
from .generic import Discriminator, Generator, GenericTrainer
from .model import _scene_index
from .trainer import PackedAdam

EPI_NONE, EPI_RELU, EPI_LRELU, EPI_DRELU, EPI_DLRELU = 0, 1, 2, 3, 4


def _p(t):
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


def gemm(x, x_rs, w, w_rs, bias, R, K, N, y, y_ld, epi=EPI_NONE, cin=None, cin_ld=0, aux=None, aux_ld=0, x_cs=1, w_cs=1):
    """y[r][n] = epi(sum_k x[r][k] w[n][k] + bias[n] + cin[r][n]; aux[r][n]) (sw_wide_gemm); tensors or raw pointers."""
    L.call("sw_wide_gemm", _p(x), x_rs, x_cs, _p(w), w_rs, w_cs, _p(bias), _p(cin), cin_ld, _p(aux), aux_ld, R, K, N, _p(y), y_ld,
           epi, L.stream())


def _off(t, floats):
    """Pointer `floats` floats into tensor t."""
    return t.data_ptr() + 4 * int(floats)


class _Flat:
    """The parameters of a list of modules as views of ONE packed fp32 buffer (each tensor on a 4-float boundary), their
    .grad as views of a second one: one Adam launch, one all-reduce, transposes by offset."""

    def __init__(self, params, device):
        self.params = list(params)
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, device=device)
        self.gflat = torch.zeros(n, device=device)
        self.slices = []
        for p, o in zip(self.params, offs):
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)
            p.grad = self.gflat[o:o + k].view(p.shape)
            self.slices.append((o, k, tuple(p.shape)))
        self.off = {id(p): o for p, o in zip(self.params, offs)}

    def g(self, p):
        """The gradient view of parameter p (re-attached if something set .grad to None)."""
        o = self.off[id(p)]
        v = self.gflat[o:o + p.numel()].view(p.shape)
        p.grad = v
        return v


class WideTrainer(GenericTrainer):
    """train() / test() / checkpoint (train.py:439-668) at hidden sizes above 64 on the wide path.  Same public surface
    as SocialWaysTrainer (step, step_many, train_epoch, test, checkpoint, load_checkpoint, losses_from)."""

    @staticmethod
    def supports(hidden_size, n_latent_codes=2, use_variety_loss=False, process_group=None):
        # use_variety_loss True = train.py:527-536 AS WRITTEN (the L2 of agent 19 of the packed batch); the best-of-K form
        # ("fixed") exists on the fused 64-unit trainer only
        return int(hidden_size) % 32 == 0 and int(hidden_size) >= 32 and int(n_latent_codes) >= 2 and use_variety_loss in (False, True)

    def __init__(self, n_next, hidden_size=128, lr_g=1e-4, lr_d=1e-3, n_unrolling_steps=1, use_social=True, use_info_loss=True,
                 loss_info_w=0.5, n_latent_codes=2, device="cuda", process_group=None, use_l2_loss=False,
                 use_variety_loss=False, loss_l2_w=0.5, use_graph=None, **perf_only):
        unknown = set(perf_only) - {"variety_k", "fused_adam"}
        if unknown:
            raise TypeError("unexpected keyword arguments: %s" % sorted(unknown))
        if not self.supports(hidden_size, n_latent_codes, use_variety_loss, process_group):
            raise L.SocialWaysHipError("wide path: hidden_size % 32 == 0, n_latent_codes >= 2, use_variety_loss False / True")
        self.device = L.indexed_device(device)
        if self.device.type != "cuda":
            raise L.SocialWaysHipError("socialways_amd runs on MI355X only (no CPU fallback)")
        self.n_next, self.noise_len = n_next, hidden_size // 2
        self.n_unrolling_steps, self.use_info_loss, self.loss_info_w = n_unrolling_steps, use_info_loss, loss_info_w
        self.use_l2_loss, self.use_variety_loss, self.loss_l2_w = use_l2_loss, bool(use_variety_loss), loss_l2_w
        self._row0 = 0
        self.n_latent_codes = n_latent_codes
        self.H = H = int(hidden_size)
        # construction order = train.py:370-385 (RNG -> init mapping, optimizer parameter order)
        self.G = Generator(H, 1, use_social=use_social).to(self.device)
        self.D = Discriminator(n_next, H, n_latent_codes).to(self.device)
        self.gp = _Flat(self.G.predictor_params(), self.device)
        self.dp = _Flat(self.D.parameters(), self.device)
        self.predictor_optimizer = PackedAdam(self.gp.flat, self.gp.gflat, self.gp.slices, lr_g)
        self.D_optimizer = PackedAdam(self.dp.flat, self.dp.gflat, self.dp.slices, lr_d)
        # data parallelism as in the fused trainer: scene-aligned shards, losses normalised by the
        # GLOBAL batch size, the two packed gradient buffers all-reduced (SUM) in front of their Adam steps - three
        # all-reduces per training step -, rank 0's replica and RNG streams broadcast (sync_replicas / sync_rng)
        self.pg, self.epoch = process_group, 0
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self._force_dist = False
        self.use_graph = True if use_graph is None else bool(use_graph)
        self.last_variety = None
        self._ws = {}            # workspaces per (B, To, P)
        self._graphs = {}        # captured steps per (B, To, scene layout, switches)
        self._graph_P = {}       # ... and the pair count of each key's layout (the workspace it replays on)
        self._seen = {}
        # Linear-only mask of D.load() (train.py:311-316): 1 for nn.Linear parameters
        m = torch.zeros_like(self.dp.flat)
        for name, p in self.D.named_parameters():
            if "lstm" not in name:
                o = self.dp.off[id(p)]
                m[o:o + p.numel()] = 1.0
        self._lin_mask = m > 0
        self._d_backup = torch.zeros_like(self.dp.flat)
        # transposed copies of the weight matrices (the backward products dx = dy W read W^T rows): ONE launch per module
        # (sw_wide_transpose) through a device table of (source offset, rows, cols, destination offset)
        enc, dec, emb, att = self.G.encoder, self.G.decoder.fc1, self.G.feature_embedder.fc, self.G.attention.W
        self.gT, self._gT_args = self._transpose_table(self.gp, dict(
            whh=enc.lstm.weight_hh_l0, w1=dec[0].weight, w2=dec[2].weight, w3=dec[4].weight, w4=dec[5].weight,
            e1=emb[2].weight, e2=emb[4].weight, att=att.weight))
        Dm = self.D
        self.dT, self._dT_args = self._transpose_table(self.dp, dict(
            whh=Dm.obsv_encoder_lstm.weight_hh_l0, of0=Dm.obsv_encoder_fc[0].weight, of1=Dm.obsv_encoder_fc[2].weight,
            pe0=Dm.pred_encoder[0].weight, pe1=Dm.pred_encoder[2].weight, cl0=Dm.classifier[0].weight,
            cl1=Dm.classifier[2].weight, la0=Dm.latent_decoder[0].weight, la1=Dm.latent_decoder[2].weight))
        # hidden sizes 64 / 128: the observation sequences run as LSTM SEQUENCE kernels (W_hh / W_hh^T register-resident for
        # the whole sequence, loaded from operand-layout images made once per weight update: sw_wide_opimage)
        self.seq = bool(L.load().sw_wide_lstm_seq_supported(H))
        # 128 units: the decode loop of predict() is ONE persistent launch streaming the decoder's operand images
        self.decloop = bool(L.load().sw_wide_dec_loop_supported(H))
        D1 = 2 * H + H // 2
        g_entries = [("whh", enc.lstm.weight_hh_l0, 4 * H, H, 0, 0, 0), ("whhT", enc.lstm.weight_hh_l0, H, 4 * H, 0, 1, 0)]
        if self.decloop:
            g_entries += [("w1h", dec[0].weight, D1, H, 0, 0, D1), ("w2", dec[2].weight, D1 // 2, D1, 0, 0, 0),
                          ("w3", dec[4].weight, D1 // 4, D1 // 2, 0, 0, 0),
                          # ... and of the transposes the backward loop streams (dx = dy W wants W^T rows)
                          ("w3T", dec[4].weight, D1 // 2, D1 // 4, 0, 1, 0), ("w2T", dec[2].weight, D1, D1 // 2, 0, 1, 0),
                          ("w1hT", dec[0].weight, H, D1, 0, 1, D1)]
            # Wx^T (composed per step, not a parameter) zero-padded to one 16-row tile, and its image
            self._wxT16 = torch.zeros(16, 4 * H, device=self.device)
            self._wxT_img = torch.zeros(16 * 4 * H, device=self.device)
            self._wxT_tab = torch.tensor([[0, 16, 4 * H, 0, 0, 0]], dtype=torch.int32).to(self.device)
        self.gI, self._gI_args = self._image_table(self.gp, g_entries)
        dwhh = Dm.obsv_encoder_lstm.weight_hh_l0
        d_entries = [("whh", dwhh, 4 * H, H, 0, 0, 0), ("whhT", dwhh, H, 4 * H, 0, 1, 0)] if self.seq else []
        # D's heads as one launch per direction (sw_wide_disc_heads_*): images of the six 2-D head matrices and their transposes
        K4, H2 = 4 * n_next, H // 2
        self.heads = bool(L.load().sw_wide_disc_heads_supported(H, K4, n_latent_codes))
        if self.heads:
            for nm, m, r, k in (("of0", Dm.obsv_encoder_fc[0], H2, H), ("of1", Dm.obsv_encoder_fc[2], H2, H2),
                                ("pe0", Dm.pred_encoder[0], H2, K4), ("pe1", Dm.pred_encoder[2], H2, H2),
                                ("cl0", Dm.classifier[0], H2, H), ("la0", Dm.latent_decoder[0], H2, H)):
                d_entries += [(nm, m.weight, r, k, 0, 0, 0), (nm + "T", m.weight, k, r, 0, 1, 0)]
        self.dI, self._dI_args = self._image_table(self.dp, d_entries) if d_entries else ({}, None)
        nl = n_latent_codes                     # reported sums: the info term's mean runs over B * nl elements (losses_from)
        k = np.ones((n_unrolling_steps + 3, 3))
        k[:n_unrolling_steps + 2, 1] = 2.0 / nl
        self._kres = torch.from_numpy(k).to(self.device)
        if self.world > 1:
            self.sync_replicas()

    def sync_replicas(self):
        """Rank 0's weights and optimizer state to every rank (construction, load_checkpoint)."""
        dist = torch.distributed
        src = dist.get_global_rank(self.pg, 0)
        for b in (self.gp.flat, self.dp.flat, self.predictor_optimizer.m, self.predictor_optimizer.v, self.D_optimizer.m,
                  self.D_optimizer.v):
            dist.broadcast(b, src, group=self.pg)
        ts = torch.tensor([float(self.predictor_optimizer.t), float(self.D_optimizer.t), float(self.epoch)],
                          device=self.device if dist.get_backend(self.pg) == "nccl" else "cpu")
        dist.broadcast(ts, src, group=self.pg)
        for o, t in zip((self.predictor_optimizer, self.D_optimizer), ts.tolist()):
            o.t = int(t)
            o.step_t.fill_(float(o.t))
        self.epoch = int(ts[2].item())

    def _allreduce(self, flat):
        if self.pg is not None and self.world > 1:
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def _empty_step(self):
        """A rank without scenes in this packed batch still takes part in the step's three all-reduces and applies the
        same updates."""
        U = self.n_unrolling_steps
        for u in range(U + 1):
            self.dp.gflat.zero_()
            self._allreduce(self.dp.gflat)
            self.D_optimizer.step()
            if u == 0 and U > 0:
                self._d_backup.copy_(self.dp.flat)
        self.gp.gflat.zero_()
        self._allreduce(self.gp.gflat)
        self.predictor_optimizer.step()
        if U > 0:
            torch.where(self._lin_mask, self._d_backup, self.dp.flat, out=self.dp.flat)
        return torch.zeros(U + 3, 3, dtype=torch.float64, device=self.device)

    def _transpose_table(self, fl, mats):
        tab, views, off, tiles = [], {}, 0, 0
        for name, p in mats.items():
            r, c = p.shape
            tab.append((fl.off[id(p)], r, c, off))
            views[name] = (off, c, r)
            off += (r * c + 3) // 4 * 4
            tiles += ((r + 31) // 32) * ((c + 31) // 32)
        buf = torch.zeros(off, device=self.device)
        out = {name: buf[o:o + a * b].view(a, b) for name, (o, a, b) in views.items()}
        tab_d = torch.tensor(tab, dtype=torch.int32).to(self.device)
        return out, (fl.flat, tab_d, len(tab), tiles, buf)

    def _image_table(self, fl, entries):
        """MFMA operand images (sw_wide_opimage) of matrices of the packed buffer `fl`: entries = (name, parameter, R, K,
        first column, transposed, source row stride or 0) for the image of Mx [R][K] - a (column block of a) matrix or its
        transpose."""
        tab, views, off = [], {}, 0
        for name, p, R, K, col0, tr, ld in entries:
            tab.append((fl.off[id(p)] + col0, R, K, off, tr, ld))
            views[name] = (off, R * K)
            off += R * K
        buf = torch.zeros(off, device=self.device)
        out = {name: buf[o:o + n] for name, (o, n) in views.items()}
        return out, (fl.flat, torch.tensor(tab, dtype=torch.int32).to(self.device), len(tab), off // 4, buf)

    def _images(self, args):
        src, tab, n, n4, dst = args
        L.call("sw_wide_opimage", L.ptr(src), L.ptr(tab), n, n4, L.ptr(dst), L.stream())

    def _transposes(self, args):
        src, tab, n, tiles, dst = args
        L.call("sw_wide_transpose", L.ptr(src), L.ptr(tab), n, tiles, L.ptr(dst), L.stream())

    # ---- buffers -----------------------------------------------------------------------------------------------------------
    # batch shapes (agents, observed steps, pairs) with live buffers + graphs: ragged datasets produce one per packed batch,
    # recurring every epoch - kept up to a count and a byte budget (a set is ~0.4 GB at 2 048 agents and 128 units; 288 GB of HBM)
    MAX_WORKSPACES = 64
    MAX_GRAPHS = 96         # captured layouts (a layout beyond the cap runs eagerly)
    WORKSPACE_BYTES = 24 << 30

    def _buffers(self, B, To, P):
        key = (B, To, P)
        w = self._ws.get(key)
        if w is not None:
            self._ws[key] = self._ws.pop(key)          # most recently used last
            return w
        while self._ws and (len(self._ws) >= self.MAX_WORKSPACES or
                            sum(v["_bytes"] for v in self._ws.values()) > self.WORKSPACE_BYTES):
            old = next(iter(self._ws))                 # the least recently used shape goes, with the graphs that have its addresses baked in
            for gk in [k for k in self._graphs if k[:2] == old[:2] and self._graph_P.get(k) == old[2]]:
                del self._graphs[gk]
                self._seen.pop(gk, None)
            del self._ws[old]
        H, Tp, dev = self.H, self.n_next, self.device
        Z, D1 = H // 2, 2 * H + H // 2
        D2, D3 = D1 // 2, D1 // 4
        Ta = To + Tp - 1
        nl, nlp = self.n_latent_codes, (self.n_latent_codes + 3) // 4 * 4
        z = lambda *s: torch.zeros(*s, device=dev)
        # the fake and the real future of a D update are the two halves of ONE [2B][4 Tp] buffer: the rollout writes its
        # prediction rows into the first, get_traj_4d the real rows into the second; D's pred_encoder reads them in place
        px = z(2 * B, 4 * Tp)
        w = dict(
            obsv=z(B, To, 2), pred=z(B, Tp, 2), noise=z(B, Z), scal=z(self.n_unrolling_steps + 4), o4=z(B, To, 4), px=px,
            pred4=px[:B].view(B, Tp, 4), p4=px[B:].view(B, Tp, 4),
            x4=z(Ta + 1, B, 4), hs=z(Ta + 1, B, H), cs=z(Ta, B, H), gates=z(Ta, B, 4 * H),
            Wx=z(4 * H, 4), WxT=z(4, 4 * H), bxc=z(4 * H),
            feat=z(max(P, 1), 4), f1=z(max(P, 1), 32), f2=z(max(P, 1), 64), f3=z(max(P, 1), H), wh=z(B, H), attn=z(max(P, 1)),
            S=z(B, H), u=z(B, D1), cat=z(Tp, B, D1), a1=z(Tp, B, D1), a2=z(Tp, B, D2), a3=z(Tp, B, D3), pcur=z(B, 2),
            # generator backward
            dgates=z(Ta, B, 4 * H), dc=z(B, H), dx4=z(B, 4), dprun=z(B, 2), dv=z(Tp, B, 4), dz3=z(Tp, B, D3), dz2=z(Tp, B, D2),
            dz1=z(Tp, B, D1), dhcat=z(B, H), dsz=z(Tp, B, H), dS=z(B, H), dhT=z(B, H), dsig=z(max(P, 1)),
            df3=z(max(P, 1), H), df2=z(max(P, 1), 64), df1=z(max(P, 1), 32), dwh=z(B, H), dh_att=z(B, H),
            dWx=z(4 * H, 4), dbx=z(4 * H),
            # discriminator
            d_hs=z(To + 1, B, H), d_cs=z(To, B, H), d_gates=z(To, B, 4 * H), o1=z(B, H // 2),
            q1=z(2 * B, H // 2), both=z(2 * B, H), c1=z(2 * B, H // 2), l1=z(2 * B, H // 2),
            label=z(2 * B, 1), code=z(2 * B, nl), dlab=z(2 * B, 4), dcod=z(2 * B, nlp), dc1=z(2 * B, H // 2),
            dl1=z(2 * B, H // 2), dboth=z(2 * B, H), dq1=z(2 * B, H // 2), docode=z(B, H // 2), do1=z(B, H // 2),
            d_dhT=z(B, H), d_dgates=z(To, B, 4 * H), d_dc=z(B, H), dpx=z(B, 4 * Tp),
            sums=z(self.n_unrolling_steps + 3, 3), ade_scr=z(3 * L.RED_BLOCKS),
            lpart=z(self.n_unrolling_steps + 2, (B + 15) // 16, 3),      # per-tile loss sums of the U + 1 D passes and the G phase
            wgrad=torch.empty(L.workspace_floats(L.WS_WGRAD, 1, 2, 1), device=dev),
            res=torch.zeros(self.n_unrolling_steps + 3, 3, dtype=torch.float64, device=dev),
        )
        # the step's host scalars travel in ONE small copy: [zeros_val, ones_val | Adam step indices of the U + 1 D updates | of G's]
        w["targets"] = w["scal"][:2]
        w["_bytes"] = sum(v.numel() * v.element_size() for v in w.values() if torch.is_tensor(v) and v._base is None)
        self._ws[key] = w
        return w

    # ---- pieces ------------------------------------------------------------------------------------------------------------
    def _wgrad(self, w, problems):
        """problems: (delta, ldd, act, lda, R, N, K, dW, ldw, db) with tensors or raw pointers."""
        arr = (ctypes.c_longlong * (10 * len(problems)))()
        for i, pr in enumerate(problems):
            for j, v in enumerate(pr):
                arr[10 * i + j] = 0 if v is None else (int(v) if isinstance(v, (int, np.integer)) else v.data_ptr())
        L.call("sw_wide_wgrad", ctypes.cast(arr, ctypes.c_void_p), len(problems), L.ptr(w["wgrad"]), L.stream())

    def _gen_forward(self, w, sc, B, To):
        """predict() (train.py:392-432): observation encoding, social pooling, Tp decode steps with the re-fed encoder."""
        H, Tp, st = self.H, self.n_next, L.stream()
        Z, D1 = H // 2, 2 * H + H // 2
        D2, D3 = D1 // 2, D1 // 4
        G = self.G
        enc, dec = G.encoder, G.decoder.fc1
        wih, whh = enc.lstm.weight_ih_l0, enc.lstm.weight_hh_l0
        # composed input matrix Wx = W_ih W_embed [4H][4], bxc = W_ih b_embed + b_ih (no non-linearity between embed and the
        # LSTM, train.py:266-268); b_hh is added by the step kernel
        gemm(wih, H, enc.embed.weight, 1, None, 4 * H, H, 4, w["Wx"], 4, w_cs=4)
        gemm(wih, H, enc.embed.bias, 1, None, 4 * H, H, 1, w["bxc"], 1, cin=enc.lstm.bias_ih_l0, cin_ld=1)
        # ... and its transpose [4][4H] (the rows of dx4 = dgates Wx in the backward pass)
        gemm(enc.embed.weight, 1, wih, H, None, 4, H, 4 * H, w["WxT"], 4 * H, x_cs=4)
        if self.decloop:
            self._wxT16[:4].copy_(w["WxT"])
            L.call("sw_wide_opimage", L.ptr(self._wxT16), L.ptr(self._wxT_tab), 1, 16 * 4 * H // 4, L.ptr(self._wxT_img), st)
        x4, hs, cs, gates, cat = w["x4"], w["hs"], w["cs"], w["gates"], w["cat"]
        x4[:To].copy_(w["o4"].transpose(0, 1))

        def lstm(t, h2=None, h2_ld=0):
            L.call("sw_wide_lstm_fwd", L.ptr(x4[t]), 4, L.ptr(hs[t]) if t > 0 else None, H, L.ptr(cs[t - 1]) if t > 0 else None,
                   L.ptr(w["Wx"]), L.ptr(w["bxc"]), L.ptr(enc.lstm.bias_hh_l0), L.ptr(whh), B, H, L.ptr(gates[t]),
                   L.ptr(cs[t]), L.ptr(hs[t + 1]), H, _p(h2), h2_ld, st)

        if self.seq:
            self._images(self._gI_args)
            L.call("sw_wide_lstm_seq_fwd", L.ptr(x4), L.ptr(w["Wx"]), L.ptr(w["bxc"]), L.ptr(enc.lstm.bias_hh_l0), L.ptr(self.gI["whh"]),
                   B, H, To, L.ptr(gates), L.ptr(cs), L.ptr(hs), L.ptr(cat[0]), D1, st)
        else:
            for t in range(To):
                lstm(t, cat[0] if t == To - 1 else None, D1)
        hT = hs[To]
        if G.use_social and sc.P > 0:
            if sc.NB:
                raise L.SocialWaysHipError("wide path: scenes above %d agents are not supported" % L.AMAX)
            emb, att = G.feature_embedder.fc, G.attention.W
            P = sc.P
            L.call("sw_pair_features", L.ptr(w["o4"][:, -1].contiguous()), L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S,
                   L.ptr(w["feat"]), st)
            gemm(w["feat"], 4, emb[0].weight, 3, emb[0].bias, P, 3, 32, w["f1"], 32, EPI_RELU)
            gemm(w["f1"], 32, emb[2].weight, 32, emb[2].bias, P, 32, 64, w["f2"], 64, EPI_RELU)
            gemm(w["f2"], 64, emb[4].weight, 64, emb[4].bias, P, 64, H, w["f3"], H)
            gemm(hT, H, att.weight, H, att.bias, B, H, H, w["wh"], H)
            L.call("sw_attn_pairs_fwd", L.ptr(w["f3"]), L.ptr(w["wh"]), L.ptr(hT), L.ptr(sc.scene_off), L.ptr(sc.pair_off),
                   sc.S, B, H, H, L.ptr(w["attn"]), L.ptr(w["S"]), st)
        else:
            w["S"].zero_()
        cat[:, :, H:2 * H] = w["S"]
        cat[:, :, 2 * H:] = w["noise"]
        if self.decloop:
            # u = W1[:, H:] [S; z] + b1: constant over the decode steps (train.py:411, 421), the initial accumulators of layer 1
            gemm(_off(cat, H), D1, _off(dec[0].weight, H), D1, dec[0].bias, B, D1 - H, D1, w["u"], D1)
            L.call("sw_wide_dec_loop_fwd", L.ptr(self.gI["w1h"]), L.ptr(self.gI["w2"]), L.ptr(self.gI["w3"]), L.ptr(self.gI["whh"]),
                   L.ptr(w["u"]), L.ptr(dec[2].bias), L.ptr(dec[4].bias), L.ptr(dec[5].weight), L.ptr(dec[5].bias), L.ptr(w["Wx"]),
                   L.ptr(w["bxc"]), L.ptr(enc.lstm.bias_hh_l0), _off(w["obsv"], 2 * (To - 1)), 2 * To, L.ptr(w["a1"]), L.ptr(w["a2"]),
                   L.ptr(w["a3"]), L.ptr(w["pred4"]), L.ptr(x4), L.ptr(gates), L.ptr(cs), L.ptr(hs), L.ptr(cat), B, H, To, Tp, st)
            return w["pred4"]
        w["pcur"].copy_(w["obsv"][:, -1])
        for i in range(Tp):
            gemm(cat[i], D1, dec[0].weight, D1, dec[0].bias, B, D1, D1, w["a1"][i], D1, EPI_LRELU)
            gemm(w["a1"][i], D1, dec[2].weight, D1, dec[2].bias, B, D1, D2, w["a2"][i], D2, EPI_LRELU)
            gemm(w["a2"][i], D2, dec[4].weight, D2, dec[4].bias, B, D2, D3, w["a3"][i], D3)
            L.call("sw_wide_out_fwd", L.ptr(w["a3"][i]), D3, L.ptr(dec[5].weight), L.ptr(dec[5].bias), L.ptr(w["pcur"]), B,
                   _off(w["pred4"], 4 * i), 4 * Tp, L.ptr(x4[To + i]), st)
            if i + 1 < Tp:                      # the step after the last decode is dead compute (train.py:430)
                lstm(To + i, cat[i + 1], D1)
        return w["pred4"]

    def _gen_backward(self, w, sc, B, To, dpred4):
        """Backward of predict() from d(loss)/d(pred_hat_4d) [B][Tp][4]: data gradients step by step, then every weight
        gradient of the generator as deferred products over the saved rows."""
        H, Tp, st = self.H, self.n_next, L.stream()
        Z, D1 = H // 2, 2 * H + H // 2
        D2, D3 = D1 // 2, D1 // 4
        Ta = To + Tp - 1
        G, gp = self.G, self.gp
        enc, dec = G.encoder, G.decoder.fc1
        whh = enc.lstm.weight_hh_l0
        self._transposes(self._gT_args)
        gT = self.gT
        gates, cs, hs, dg = w["gates"], w["cs"], w["hs"], w["dgates"]
        w["dprun"].zero_()
        have_dc = False

        def lstm_bwd(t, dh_ext, dhe_ld, dh_ext2=None, dhe2_ld=0):
            nonlocal have_dc
            L.call("sw_wide_lstm_bwd", _p(dh_ext), dhe_ld, _p(dh_ext2), dhe2_ld, L.ptr(dg[t + 1]) if t + 1 < Ta else None,
                   L.ptr(gT["whh"]), L.ptr(gates[t]), L.ptr(cs[t]), L.ptr(cs[t - 1]) if t > 0 else None,
                   L.ptr(w["dc"]) if have_dc else None, B, H, L.ptr(dg[t]), L.ptr(w["dc"]), st)
            have_dc = True

        if self.decloop:
            gI = self.gI
            L.call("sw_wide_dec_loop_bwd", L.ptr(gI["whhT"]), L.ptr(gI["w3T"]), L.ptr(gI["w2T"]), L.ptr(gI["w1hT"]), L.ptr(self._wxT_img),
                   L.ptr(dec[5].weight), L.ptr(dpred4), L.ptr(w["a1"]), L.ptr(w["a2"]), L.ptr(gates), L.ptr(cs), L.ptr(dg), L.ptr(w["dv"]),
                   L.ptr(w["dz3"]), L.ptr(w["dz2"]), L.ptr(w["dz1"]), L.ptr(w["dhcat"]), L.ptr(w["dc"]), B, H, To, Tp, st)
            have_dc = Tp > 1
        for i in range(-1 if self.decloop else Tp - 1, -1, -1):
            t_in = To + i                      # the LSTM step that consumed x4 = (p_i, v_i)
            dgt = None
            if i + 1 < Tp:
                # h of step t_in feeds decode step i + 1 (dhcat) and LSTM step t_in + 1 (dgates of t_in + 1)
                lstm_bwd(t_in, w["dhcat"], H)
                dgt = dg[t_in]
            # dx4 = dgates Wx, the position / velocity chain and dz3 = dv W4 (fc4: linear, no activation) in one launch
            L.call("sw_wide_out_bwd", _off(dpred4, 4 * i), 4 * Tp, _p(dgt), L.ptr(w["WxT"]), 4 * H, L.ptr(w["dprun"]), B,
                   L.ptr(w["dv"][i]), L.ptr(dec[5].weight), D3, L.ptr(w["dz3"][i]), st)
            gemm(w["dz3"][i], D3, gT["w3"], D3, None, B, D3, D2, w["dz2"][i], D2, EPI_DLRELU, aux=w["a2"][i], aux_ld=D2)
            gemm(w["dz2"][i], D2, gT["w2"], D2, None, B, D2, D1, w["dz1"][i], D1, EPI_DLRELU, aux=w["a1"][i], aux_ld=D1)
            gemm(w["dz1"][i], D1, gT["w1"], D1, None, B, D1, H, w["dhcat"], H)                      # d cat[:, :H] = d h_{To-1+i}
        # dS = sum over the steps of dz1 W1[:, H:2H] (z is an input: no gradient wanted): one product over all Tp B rows, then
        # the sum over the steps
        gemm(w["dz1"], D1, _off(gT["w1"], H * D1), D1, None, Tp * B, D1, H, w["dsz"], H)
        L.call("sw_wide_sum_steps", L.ptr(w["dsz"]), B * H, H, Tp, B, H, L.ptr(w["dS"]), H, st)
        problems = []
        hT = hs[To]
        dh2, dh2_ld = None, 0
        if G.use_social and sc.P > 0:
            emb, att = G.feature_embedder.fc, G.attention.W
            P = sc.P
            L.call("sw_attn_pairs_bwd", L.ptr(w["f3"]), L.ptr(w["wh"]), L.ptr(hT), L.ptr(w["attn"]), L.ptr(w["dS"]),
                   L.ptr(sc.scene_off), L.ptr(sc.pair_off), sc.S, B, H, H, L.ptr(w["dsig"]), L.ptr(w["df3"]), L.ptr(w["dwh"]),
                   L.ptr(w["dh_att"]), st)
            gemm(w["df3"], H, gT["e2"], H, None, P, H, 64, w["df2"], 64, EPI_DRELU, aux=w["f2"], aux_ld=64)
            gemm(w["df2"], 64, gT["e1"], 64, None, P, 64, 32, w["df1"], 32, EPI_DRELU, aux=w["f1"], aux_ld=32)
            gemm(w["dwh"], H, gT["att"], H, None, B, H, H, w["dh_att"], H, cin=w["dh_att"], cin_ld=H)   # += dwh W_att
            dh2, dh2_ld = w["dh_att"], H
            problems += [(w["df1"], 32, w["feat"], 4, P, 32, 3, gp.g(emb[0].weight), 3, gp.g(emb[0].bias)),
                         (w["df2"], 64, w["f1"], 32, P, 64, 32, gp.g(emb[2].weight), 32, gp.g(emb[2].bias)),
                         (w["df3"], H, w["f2"], 64, P, H, 64, gp.g(emb[4].weight), 64, gp.g(emb[4].bias)),
                         (w["dwh"], H, hT, H, B, H, H, gp.g(att.weight), H, gp.g(att.bias))]
        else:
            for p in list(G.feature_embedder.parameters()) + list(G.attention.parameters()):
                gp.g(p).zero_()
        # observation steps: h_{To-1} feeds decode step 0 (dhcat), LSTM step To (dgates of To) and the social block
        if self.seq:
            L.call("sw_wide_lstm_seq_bwd", L.ptr(w["dhcat"]), H, _p(dh2), dh2_ld, L.ptr(dg[To]) if To < Ta else None,
                   L.ptr(w["dc"]) if have_dc else None, L.ptr(self.gI["whhT"]), L.ptr(gates), L.ptr(cs), B, H, To, L.ptr(dg), st)
        else:
            for t in range(To - 1, -1, -1):
                if t == To - 1:
                    lstm_bwd(t, w["dhcat"], H, dh2, dh2_ld)
                else:
                    lstm_bwd(t, None, 0)
        # weight gradients: LSTM (W_hh against h_{t-1}: hs[t], zero slab first; the composed input matrix against x4)
        problems += [(dg, 4 * H, hs, H, Ta * B, 4 * H, H, gp.g(whh), H, gp.g(enc.lstm.bias_hh_l0)),
                     (dg, 4 * H, w["x4"], 4, Ta * B, 4 * H, 4, w["dWx"], 4, w["dbx"]),
                     (w["dz1"], D1, w["cat"], D1, Tp * B, D1, D1, gp.g(dec[0].weight), D1, gp.g(dec[0].bias)),
                     (w["dz2"], D2, w["a1"], D1, Tp * B, D2, D1, gp.g(dec[2].weight), D1, gp.g(dec[2].bias)),
                     (w["dz3"], D3, w["a2"], D2, Tp * B, D3, D2, gp.g(dec[4].weight), D2, gp.g(dec[4].bias)),
                     (w["dv"], 4, w["a3"], D3, Tp * B, 2, D3, gp.g(dec[5].weight), D3, gp.g(dec[5].bias))]
        self._wgrad(w, problems)
        # back through the composition Wx = W_ih W_e, bxc = W_ih b_e + b_ih:
        #   dW_ih = dWx W_e^T + dbx b_e^T, dW_e = W_ih^T dWx, db_e = W_ih^T dbx, db_ih = dbx (= db_hh, written above)
        wih = enc.lstm.weight_ih_l0
        gemm(w["dWx"], 4, enc.embed.weight, 4, None, 4 * H, 4, H, gp.g(wih), H)
        gemm(w["dbx"], 1, enc.embed.bias, 1, None, 4 * H, 1, H, gp.g(wih), H, cin=gp.g(wih), cin_ld=H)
        gemm(wih, 1, w["dWx"], 1, None, H, 4 * H, 4, gp.g(enc.embed.weight), 4, x_cs=H, w_cs=4)
        gemm(wih, 1, w["dbx"], 1, None, H, 4 * H, 1, gp.g(enc.embed.bias), 1, x_cs=H)
        gp.g(enc.lstm.bias_ih_l0).copy_(w["dbx"])

    def _disc_forward(self, w, B, To, nb, loss=None):
        """Discriminator.forward (train.py:294-309) on nb future branches sharing the observation encoding; rows of branch k
        at [k B, (k + 1) B) of w["px"] (0: the rollout's prediction, 1: the real future)."""
        H, Tp, st, D = self.H, self.n_next, L.stream(), self.D
        H2, nl = H // 2, self.n_latent_codes
        lstm = D.obsv_encoder_lstm
        x4, hs, cs, gates = w["x4"], w["d_hs"], w["d_cs"], w["d_gates"]
        if self._dI_args is not None:
            self._images(self._dI_args)
        if self.seq:
            L.call("sw_wide_lstm_seq_fwd", L.ptr(x4), L.ptr(lstm.weight_ih_l0), L.ptr(lstm.bias_ih_l0), L.ptr(lstm.bias_hh_l0),
                   L.ptr(self.dI["whh"]), B, H, To, L.ptr(gates), L.ptr(cs), L.ptr(hs), None, 0, st)
        for t in range(0 if self.seq else To):
            L.call("sw_wide_lstm_fwd", L.ptr(x4[t]), 4, L.ptr(hs[t]) if t > 0 else None, H, L.ptr(cs[t - 1]) if t > 0 else None,
                   L.ptr(lstm.weight_ih_l0), L.ptr(lstm.bias_ih_l0), L.ptr(lstm.bias_hh_l0), L.ptr(lstm.weight_hh_l0), B, H,
                   L.ptr(gates[t]), L.ptr(cs[t]), L.ptr(hs[t + 1]), H, None, 0, st)
        of, pe, cl, la = D.obsv_encoder_fc, D.pred_encoder, D.classifier, D.latent_decoder
        R = nb * B
        if self.heads:
            L.call("sw_wide_disc_heads_fwd", self._heads_args(w, B, To, nb, False, False, False, loss), st)
            return True
        gemm(hs[To], H, of[0].weight, H, of[0].bias, B, H, H2, w["o1"], H2, EPI_LRELU)
        for k in range(nb):      # obsv_code into the first half of `both`, once per branch
            gemm(w["o1"], H2, of[2].weight, H2, of[2].bias, B, H2, H2, _off(w["both"], k * B * H), H)
        gemm(w["px"], 4 * Tp, pe[0].weight, 4 * Tp, pe[0].bias, R, 4 * Tp, H2, w["q1"], H2, EPI_LRELU)
        gemm(w["q1"], H2, pe[2].weight, H2, pe[2].bias, R, H2, H2, _off(w["both"], H2), H)
        gemm(w["both"], H, cl[0].weight, H, cl[0].bias, R, H, H2, w["c1"], H2, EPI_LRELU)
        gemm(w["c1"], H2, cl[2].weight, H2, cl[2].bias, R, H2, 1, w["label"], 1)
        gemm(w["both"], H, la[0].weight, H, la[0].bias, R, H, H2, w["l1"], H2, EPI_LRELU)
        gemm(w["l1"], H2, la[2].weight, H2, la[2].bias, R, H2, nl, w["code"], nl)

    def _heads_args(self, w, B, To, nb, backward, need_obs, want_dpred, loss=None):
        """struct WideHeads of sw_wide.hip as 52 host values.  loss = (target index of branch 0, of branch 1, label scale, code
        scale, partial-sum rows [tiles][3]): the forward launch also forms the loss gradients and the tile's sums of squares."""
        D, I = self.D, self.dI
        of, pe, cl, la = D.obsv_encoder_fc, D.pred_encoder, D.classifier, D.latent_decoder
        sfx = "T" if backward else ""
        vals = [I[n + sfx] for n in ("of0", "of1", "pe0", "pe1", "cl0", "la0")]
        vals += [of[0].bias, of[2].bias, pe[0].bias, pe[2].bias, cl[0].bias, la[0].bias]
        vals += [cl[2].weight, cl[2].bias, la[2].weight, la[2].bias, w["d_hs"][To], w["px"]]
        vals += [w[k] for k in ("o1", "q1", "both", "c1", "l1", "label", "code", "dlab", "dcod", "dc1", "dl1", "dboth", "dq1",
                                "docode", "do1", "d_dhT", "dpx")]
        nl = self.n_latent_codes
        ints = [B, self.H, 4 * self.n_next, nb, nl, (nl + 3) // 4 * 4, int(need_obs), int(want_dpred)]
        if loss is None:
            tail = [0] * 9
        else:
            t0, t1, gl, gc, part = loss
            as_bits = lambda x: struct.unpack("q", struct.pack("d", float(x)))[0]
            tail = [1, int(t0), int(t1), self.H // 2, as_bits(gl), as_bits(gc), w["targets"].data_ptr(), w["noise"].data_ptr(),
                    part.data_ptr()]
        arr = (ctypes.c_longlong * 52)(*([v.data_ptr() for v in vals] + ints + tail))
        return ctypes.cast(arr, ctypes.c_void_p)

    def _disc_heads_backward(self, w, B, nb, want_dpred, need_obs=False):
        """Backward of the heads from dlab / dcod: every delta the weight gradients need, optionally d/d(pred) of branch 0."""
        H, Tp, D = self.H, self.n_next, self.D
        H2, nl, nlp = H // 2, self.n_latent_codes, (self.n_latent_codes + 3) // 4 * 4
        of, pe, cl, la = D.obsv_encoder_fc, D.pred_encoder, D.classifier, D.latent_decoder
        R = nb * B
        if self.heads:
            if not self.seq and need_obs:
                self._transposes(self._dT_args)        # W_hh^T for the per-step LSTM backward
            L.call("sw_wide_disc_heads_bwd", self._heads_args(w, B, 0, nb, True, need_obs, want_dpred), L.stream())
            return
        self._transposes(self._dT_args)
        dT = self.dT
        gemm(w["dlab"], 4, dT["cl1"], 1, None, R, 1, H2, w["dc1"], H2, EPI_DLRELU, aux=w["c1"], aux_ld=H2)
        gemm(w["dcod"], nlp, dT["la1"], nl, None, R, nl, H2, w["dl1"], H2, EPI_DLRELU, aux=w["l1"], aux_ld=H2)
        gemm(w["dc1"], H2, dT["cl0"], H2, None, R, H2, H, w["dboth"], H)
        gemm(w["dl1"], H2, dT["la0"], H2, None, R, H2, H, w["dboth"], H, cin=w["dboth"], cin_ld=H)
        gemm(_off(w["dboth"], H2), H, dT["pe1"], H2, None, R, H2, H2, w["dq1"], H2, EPI_DLRELU, aux=w["q1"], aux_ld=H2)
        if want_dpred:
            gemm(w["dq1"], H2, dT["pe0"], H2, None, B, H2, 4 * Tp, w["dpx"], 4 * Tp)

    def _disc_backward(self, w, B, To):
        """Backward of a D update (both branches): heads, observation path, LSTM through time, weight gradients."""
        H, Tp, st, D, dp = self.H, self.n_next, L.stream(), self.D, self.dp
        H2, nl, nlp = H // 2, self.n_latent_codes, (self.n_latent_codes + 3) // 4 * 4
        of, pe, cl, la = D.obsv_encoder_fc, D.pred_encoder, D.classifier, D.latent_decoder
        lstm = D.obsv_encoder_lstm
        dT = self.dT
        self._disc_heads_backward(w, B, 2, False, need_obs=True)
        if not self.heads:
            # obsv_code feeds both branches: its gradient is the sum over the branches
            L.call("sw_wide_sum_steps", L.ptr(w["dboth"]), B * H, H, 2, B, H2, L.ptr(w["docode"]), H2, st)
            gemm(w["docode"], H2, dT["of1"], H2, None, B, H2, H2, w["do1"], H2, EPI_DLRELU, aux=w["o1"], aux_ld=H2)
            gemm(w["do1"], H2, dT["of0"], H2, None, B, H2, H, w["d_dhT"], H)
        hs, cs, gates, dg = w["d_hs"], w["d_cs"], w["d_gates"], w["d_dgates"]
        if self.seq:
            L.call("sw_wide_lstm_seq_bwd", L.ptr(w["d_dhT"]), H, None, 0, None, None, L.ptr(self.dI["whhT"]), L.ptr(gates), L.ptr(cs),
                   B, H, To, L.ptr(dg), st)
        for t in range(-1 if self.seq else To - 1, -1, -1):
            L.call("sw_wide_lstm_bwd", L.ptr(w["d_dhT"]) if t == To - 1 else None, H, None, 0,
                   L.ptr(dg[t + 1]) if t + 1 < To else None, L.ptr(dT["whh"]), L.ptr(gates[t]), L.ptr(cs[t]),
                   L.ptr(cs[t - 1]) if t > 0 else None, L.ptr(w["d_dc"]) if t + 1 < To else None, B, H, L.ptr(dg[t]),
                   L.ptr(w["d_dc"]), st)
        g = dp.g
        R = 2 * B
        self._wgrad(w, [
            (dg, 4 * H, hs, H, To * B, 4 * H, H, g(lstm.weight_hh_l0), H, g(lstm.bias_hh_l0)),
            (dg, 4 * H, w["x4"], 4, To * B, 4 * H, 4, g(lstm.weight_ih_l0), 4, g(lstm.bias_ih_l0)),
            (w["do1"], H2, hs[To], H, B, H2, H, g(of[0].weight), H, g(of[0].bias)),
            (w["docode"], H2, w["o1"], H2, B, H2, H2, g(of[2].weight), H2, g(of[2].bias)),
            (w["dq1"], H2, w["px"], 4 * Tp, R, H2, 4 * Tp, g(pe[0].weight), 4 * Tp, g(pe[0].bias)),
            (_off(w["dboth"], H2), H, w["q1"], H2, R, H2, H2, g(pe[2].weight), H2, g(pe[2].bias)),
            (w["dc1"], H2, w["both"], H, R, H2, H, g(cl[0].weight), H, g(cl[0].bias)),
            (w["dlab"], 4, w["c1"], H2, R, 1, H2, g(cl[2].weight), H2, g(cl[2].bias)),
            (w["dl1"], H2, w["both"], H, R, H2, H, g(la[0].weight), H, g(la[0].bias)),
            (w["dcod"], nlp, w["l1"], H2, R, nl, H2, g(la[2].weight), H2, g(la[2].bias)),
        ])

    def _sq(self, a, lda, b, ldb, targets, t_idx, R, C, gscale, out, da, ldda):
        L.call("sw_sqdiff", _p(a), lda, _p(b), ldb, _p(targets), int(t_idx), R, C, float(gscale), _p(out), _p(da), ldda, L.stream())

    # ---- the step ------------------------------------------------------------------------------------------------------------
    def _step_device(self, w, sc, B, To, ss, Bg):
        """Device-only body of train.py:458-554 on the staged inputs of `w` (capturable), as a GENERATOR: it yields the
        packed gradient buffer at each of the three points where data-parallel ranks all-reduce before the optimizer step
        (D, D, G).  Bg = agents of the whole packed batch over all ranks (the loss means run over it)."""
        U, Tp, nl, H = self.n_unrolling_steps, self.n_next, self.n_latent_codes, self.H
        nlp = (nl + 3) // 4 * 4
        Bg = float(Bg)
        wi = self.loss_info_w if self.use_info_loss else 0.0
        sums, z, tg = w["sums"], w["noise"], w["targets"]
        L.call("sw_traj_4d", L.ptr(w["obsv"]), L.ptr(w["pred"]), B, To, Tp, L.ptr(w["o4"]), L.ptr(w["p4"]), L.stream())
        fake = self._gen_forward(w, sc, B, To)        # the three predict() calls of a step are identical
        tiles = (B + 15) // 16
        gl, gc = 2.0 / Bg, wi * 2.0 / (nl * Bg)
        for u in range(U + 1):                           # train.py:476-499
            # (with the fused heads the loss gradients and per-tile sums of squares come out of the forward launch)
            if not self._disc_forward(w, B, To, 2, loss=(0, 1, gl, gc, w["lpart"][u])):
                self._sq(w["label"], 1, None, 0, tg, 0, B, 1, gl, _off(sums, 3 * u), w["dlab"], 4)
                self._sq(_off(w["label"], B), 1, None, 0, tg, 1, B, 1, gl, _off(sums, 3 * u + 2), _off(w["dlab"], 4 * B), 4)
                self._sq(w["code"], nl, z, H // 2, None, 0, B, nl, gc, _off(sums, 3 * u + 1), w["dcod"], nlp)
            self._disc_backward(w, B, To)
            yield self.dp.gflat
            self.D_optimizer.step()
            if u == 0 and U > 0:
                self._d_backup.copy_(self.dp.flat)       # deepcopy(D) after the first update (train.py:498-499)
        # ---- generator update (train.py:503-539) ----
        if not self._disc_forward(w, B, To, 1, loss=(1, 1, gl, gc, w["lpart"][U + 1])):
            self._sq(w["label"], 1, None, 0, tg, 1, B, 1, gl, _off(sums, 3 * (U + 1)), w["dlab"], 4)
            self._sq(w["code"], nl, z, H // 2, None, 0, B, nl, gc, _off(sums, 3 * (U + 1) + 1), w["dcod"], nlp)
        self._disc_heads_backward(w, B, 1, True)
        dpred4 = w["dpx"]
        if self.use_l2_loss:                             # train.py:525-526
            L.call("sw_l2_grad", L.ptr(fake), L.ptr(w["pred"]), B, Tp, 0, B, self.loss_l2_w / (Bg * Tp), L.ptr(dpred4), L.stream())
        if self.use_variety_loss:                        # train.py:527-536 as written: only the k = 19 term survives - the
            r = 19 - self._row0                          # L2 of AGENT 19 of the packed batch (its 20 rollouts are identical)
            if 0 <= r < B:
                L.call("sw_l2_grad", L.ptr(fake), L.ptr(w["pred"]), B, Tp, r, r + 1, self.loss_l2_w / Tp, L.ptr(dpred4), L.stream())
        self._gen_backward(w, sc, B, To, dpred4)
        yield self.gp.gflat
        self.predictor_optimizer.step()
        if U > 0:                                        # D.load(backup): Linear layers only (train.py:311-316, 541-542)
            torch.where(self._lin_mask, self._d_backup, self.dp.flat, out=self.dp.flat)
        if self.heads:               # the loss rows of all passes: one sum over the tiles
            torch.sum(w["lpart"], dim=1, out=sums[:U + 2])
        sums[U + 2].zero_()
        L.call("sw_ade_fde", L.ptr(fake), L.ptr(w["pred"]), B, Tp, 1.0 / float(ss), L.ptr(sums[U + 2]), L.ptr(w["ade_scr"]), L.stream())
        # the reported sums (SocialWaysTrainer.step()'s layout): the info term's mean runs over B * nl elements
        torch.mul(sums, self._kres, out=w["res"])          # (float32 x float64 -> float64 in one launch)

    def step(self, obsv, pred, sub_batches, zeros_val, ones_val, noise, ss=1.0, global_B=None, out=None, global_row0=0,
             variety_noise=None):
        """One packed batch (train.py:458-554).  Returns the (U+3, 3) float64 sums of SocialWaysTrainer.step()."""
        dev = self.device
        B, To = obsv.shape[0], obsv.shape[1]
        Bg = float(global_B if global_B is not None else B)
        self._row0 = int(global_row0)       # first row of this rank's shard in the packed batch (variety term only)
        if self.use_variety_loss and Bg < 20:
            raise ValueError("use_variety_loss indexes agent 19 of the packed batch (train.py:531): batch of %d" % Bg)
        sc = _scene_index(sub_batches, B, dev)
        if self.G.use_social and sc.NB:       # (checked HERE: a batch whose multi-agent scenes are ALL above the limit has P = 0)
            raise L.SocialWaysHipError("wide path: scenes above %d agents are not supported" % L.AMAX)
        w = self._buffers(B, To, sc.P)
        w["obsv"].copy_(obsv)
        w["pred"].copy_(pred)
        w["noise"].copy_(noise, non_blocking=True)
        U, dopt, gopt = self.n_unrolling_steps, self.D_optimizer, self.predictor_optimizer
        # label-noise scalars + the 1-based indices of this step's Adam updates (read by a REPLAYED step; an eager step counts itself)
        w["scal"].copy_(torch.tensor([float(zeros_val), float(ones_val)] + [float(dopt.t + k + 1) for k in range(U + 1)]
                                     + [float(gopt.t + 1)], dtype=torch.float32))
        # the captured step bakes the Adam step indices' ADDRESSES in (PackedAdam.step_t) - their values advance on the host
        # ... and, as host scalars of the recorded launches, both optimizers' hyper-parameters and the loss weights: they are part
        # of the key (as in SocialWaysTrainer._graph_key), so a checkpoint loaded with another lr never replays the old one
        og, od = gopt.param_groups[0], dopt.param_groups[0]
        key = (B, To, sc.key, float(ss), Bg, self.use_l2_loss, self.use_info_loss, self.n_unrolling_steps, self.use_variety_loss,
               self._row0, self.loss_info_w, self.loss_l2_w,
               og["lr"], tuple(og["betas"]), og["eps"], og.get("weight_decay", 0),
               od["lr"], tuple(od["betas"]), od["eps"], od.get("weight_decay", 0))
        n_seen = self._seen.get(key, 0)
        self._seen[key] = n_seen + 1
        self._graph_P[key] = sc.P
        if len(self._seen) > 4096:                     # ragged datasets: bookkeeping of layouts seen once does not grow forever
            self._seen = {k: v for k, v in self._seen.items() if k in self._graphs}
            self._graph_P = {k: v for k, v in self._graph_P.items() if k in self._graphs}
        if not self.use_graph or n_seen < 2 or (key not in self._graphs and len(self._graphs) >= self.MAX_GRAPHS):
            # two eager steps of a layout first (allocations, caches); beyond the cap on captured layouts: eager
            for buf in self._step_device(w, sc, B, To, ss, Bg):
                self._allreduce(buf)
        else:
            g = self._graphs.get(key)
            if g is None:
                g = self._capture(key, w, sc, B, To, ss, Bg)
            self._replay(g)
        self.last_pred_hat = w["pred4"]
        return w["res"].clone()

    def step_many(self, batches, sub_batches, ss=1.0, global_B=None, out=None, global_row0=0):
        return [self.step(o, p, sub_batches, zv, ov, nz, ss, global_B, out, global_row0) for o, p, zv, ov, nz in batches]

    # The optimizers count their updates on the host (PackedAdam.t -> step_t.fill_) - inside a captured graph that fill is
    # replayed with the value of capture time.  A captured step therefore reads the update indices from device scalars that
    # the host sets before every replay (w["scal"], together with the label-noise scalars: one small copy per step).
    def _capture(self, key, w, sc, B, To, ss, Bg):
        """Single process: the whole step as ONE graph.  Data parallel: one graph SEGMENT per stretch between two
        all-reduce points (4 segments), the collectives run between the segment replays."""
        torch.cuda.synchronize()
        dopt, gopt = self.D_optimizer, self.predictor_optimizer
        U = self.n_unrolling_steps
        steps_d = [w["scal"][2 + k] for k in range(U + 1)]       # set by step() with the label-noise scalars (one copy)
        step_g = w["scal"][U + 3]
        real_d, real_g = dopt.step, gopt.step
        it = iter(steps_d)
        dopt.step = lambda st=None: real_d(next(it))
        gopt.step = lambda st=None: real_g(step_g)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        segments, pool = [], None
        gen = self._step_device(w, sc, B, To, ss, Bg)
        try:
            with torch.cuda.stream(side):
                done = False
                while not done:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side, pool=pool):
                        buf = None
                        while True:           # a single process runs through the all-reduce points inside ONE capture
                            try:
                                buf = next(gen)
                            except StopIteration:
                                buf, done = None, True
                                break
                            if self.world > 1:
                                break
                    pool = graph.pool() if pool is None else pool
                    segments.append((graph, buf))
        finally:
            dopt.step, gopt.step = real_d, real_g
        torch.cuda.current_stream().wait_stream(side)
        g = dict(segments=segments, n_d=U + 1)
        self._graphs[key] = g
        return g

    def _replay(self, g):
        dopt, gopt = self.D_optimizer, self.predictor_optimizer
        for graph, buf in g["segments"]:
            graph.replay()
            if buf is not None:
                self._allreduce(buf)
        dopt.t += g["n_d"]           # (their device counters step_t are refreshed by the next eager update)
        gopt.t += 1

    def release_graphs(self):
        self._graphs.clear()

    def load_checkpoint(self, ck):
        r = super().load_checkpoint(ck)             # (broadcasts rank 0's replica when there is a process group)
        self.release_graphs()                       # captured steps bake the optimizers' host scalars in: re-capture
        self._seen.clear()
        self._graph_P.clear()
        for fl in (self.gp, self.dp):          # load_state_dict copies in place: the views still alias the packed buffers
            for p in fl.params:
                assert p.data_ptr() == fl.flat.data_ptr() + 4 * fl.off[id(p)], "parameter left its packed buffer"
        return r
