"""CPU oracle for the Social Ways GAN training inner loop  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 *restatement* of the algorithm of the reference
`crowdbotp/socialways` `train.py` (the path named by BASELINE.json:north_star).  It is the
checker the HIP path is compared against; it is never the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `socialways_amd/` imports it (tests/test_abi.py::test_product_never_imports_the_oracle enforces that).

Parity status: PINNED.  The reference holds no tests or golden vectors of its own,
so the pin is the reference itself: `oracle/make_golden.py` imports the unmodified
`/root/reference/train.py` in the build container (CPU shims only), runs it on the
same seeds and commits its outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks
this restatement against those vectors (per-op tensors, per-step losses, gradients, weights after
training, ADE/FDE, `test()` outputs).

Every function cites the reference lines it follows (paths relative to /root/reference).
Two variants of the social block are provided:
  * ``faithful``  - the reference's op sequence: dense BxB pair tensors + the per-agent Python loop
                    (train.py:153-175, 229-241).  O(B^2) memory, O(B^3) backward.
  * ``blockdiag`` - same math restricted to in-scene pairs, vectorised per scene size.  This is
                    what scales to the dense-crowd config and what `bench.py` times as cpu_baseline.
"""
import copy
from itertools import chain

import numpy as np
import torch
import torch.nn as nn
import torch.optim as opt


# ----------------------------------------------------------------------------- train.py:130-138
def get_traj_4d(obsv_p, pred_p):
    """(x,y) -> (x,y,vx,vy).  First obs velocity duplicated (train.py:132); predicted velocities
    chained from the last observation (train.py:135-136).  `pred_p=[]` -> obs only (train.py:134)."""
    obsv_v = obsv_p[:, 1:] - obsv_p[:, :-1]
    obsv_v = torch.cat([obsv_v[:, 0].unsqueeze(1), obsv_v], dim=1)
    obsv_4d = torch.cat([obsv_p, obsv_v], dim=2)
    if len(pred_p) == 0:
        return obsv_4d
    pred_p_1 = torch.cat([obsv_p[:, -1].unsqueeze(1), pred_p[:, :-1]], dim=1)
    pred_v = pred_p - pred_p_1
    pred_4d = torch.cat([pred_p, pred_v], dim=2)
    return obsv_4d, pred_4d


# ----------------------------------------------------------------------------- train.py:153-175
class AttentionPooling(nn.Module):
    def __init__(self, h_dim, f_dim):
        super().__init__()
        self.f_dim = f_dim
        self.h_dim = h_dim
        self.W = nn.Linear(h_dim, f_dim, bias=True)

    def forward(self, f, h, sub_batches):
        """Faithful: per scene, per agent: sigma_ij=<f[i,j],Wh[j]>, sigma_ii:=-1000, softmax over
        the scene, S_i = sum_j a_ij h_j (pools raw h).  N==1 scenes keep S=0 (train.py:165)."""
        Wh = self.W(h)
        S = torch.zeros_like(h)
        for sb in sub_batches:
            s0, s1 = int(sb[0]), int(sb[1])
            N = s1 - s0
            if N == 1:
                continue
            for ii in range(s0, s1):
                fi = f[ii, s0:s1]
                sigma_i = torch.bmm(fi.unsqueeze(1), Wh[s0:s1].unsqueeze(2))
                sigma_i[ii - s0] = -1000
                attentions = torch.softmax(sigma_i.squeeze(), dim=0)
                S[ii] = torch.mm(attentions.view(1, N), h[s0:s1])
        return S


# ----------------------------------------------------------------------------- train.py:178-189
class EmbedSocialFeatures(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.fc = nn.Sequential(nn.Linear(input_size, 32), nn.ReLU(),
                                nn.Linear(32, 64), nn.ReLU(),
                                nn.Linear(64, hidden_size))

    def forward(self, ftr_list, sub_batches):
        return self.fc(ftr_list)


# ----------------------------------------------------------------------------- train.py:192-205
def DCA(xA_4d, xB_4d):
    """Scalar spec of distance-to-closest-approach (train.py:192-198); tau not clamped."""
    dp = xA_4d[:2] - xB_4d[:2]
    dv = xA_4d[2:] - xB_4d[2:]
    ttca = torch.dot(-dp, dv) / (torch.norm(dv) ** 2 + 1E-6)
    return torch.norm(dp + ttca * dv)


def Bearing(xA_4d, xB_4d):
    """Scalar spec of the bearing cosine (train.py:201-205)."""
    dp = xA_4d[:2] - xB_4d[:2]
    v = xA_4d[2:]
    return torch.dot(dp, v) / (torch.norm(dp) * torch.norm(v) + 1E-6)


# ----------------------------------------------------------------------------- train.py:208-241
def SocialFeatures(x, sub_batches):
    """Dense (B,B,3) [dist, bearing, dca] from the last observed 4-d state (train.py:229-241).
    Row i, column j: dp = p_i - p_j, dv = v_i - v_j (x_hor - x_ver, train.py:232-234)."""
    N = x.shape[0]
    last = x[:, -1]
    x_ver = last.unsqueeze(0).repeat(N, 1, 1)
    x_hor = last.unsqueeze(1).repeat(1, N, 1)
    D = x_hor - x_ver
    Dp, Dv = D[:, :, :2], D[:, :, 2:]
    l2 = Dp.norm(dim=2)
    # BearingMTX, train.py:221-226
    v = last[:, 2:].unsqueeze(1).repeat(1, N, 1)
    dot_dp_v = Dp[:, :, 0] * v[:, :, 0] + Dp[:, :, 1] * v[:, :, 1]
    bearing = torch.div(dot_dp_v, torch.norm(Dp, dim=2) * torch.norm(v, dim=2) + 1E-6)
    # DCA_MTX, train.py:208-218
    dot_dp_dv = Dp[:, :, 0] * Dv[:, :, 0] + Dp[:, :, 1] * Dv[:, :, 1]
    dv_sq = Dv[:, :, 0] * Dv[:, :, 0] + Dv[:, :, 1] * Dv[:, :, 1] + 1E-6
    ttca = -torch.div(dot_dp_dv, dv_sq)
    dca = torch.stack([Dp[:, :, 0] + ttca * Dv[:, :, 0], Dp[:, :, 1] + ttca * Dv[:, :, 1]], dim=2)
    dca = torch.norm(dca, dim=2)
    return torch.stack([l2, bearing, dca], dim=2)


def pair_features(si, sj):
    """[dist, bearing, dca] for explicit pairs; si/sj (...,4) are the 4-d states of agent i / j.
    Same arithmetic as SocialFeatures (train.py:208-241) on arbitrary leading dims."""
    dpx, dpy = si[..., 0] - sj[..., 0], si[..., 1] - sj[..., 1]
    dvx, dvy = si[..., 2] - sj[..., 2], si[..., 3] - sj[..., 3]
    vx, vy = si[..., 2], si[..., 3]
    l2 = torch.sqrt(dpx * dpx + dpy * dpy)
    bearing = (dpx * vx + dpy * vy) / (l2 * torch.sqrt(vx * vx + vy * vy) + 1E-6)
    ttca = -(dpx * dvx + dpy * dvy) / (dvx * dvx + dvy * dvy + 1E-6)
    cx, cy = dpx + ttca * dvx, dpy + ttca * dvy
    dca = torch.sqrt(cx * cx + cy * cy)
    return torch.stack([l2, bearing, dca], dim=-1)


def social_pool_blockdiag(last4, h, sub_batches, feature_embedder, attention):
    """Block-diagonal restatement of SocialFeatures -> feature_embedder -> attention
    (train.py:409-411): identical math, only in-scene pairs are formed.  Scenes are grouped by
    size so each group is one batched tensor op."""
    S = torch.zeros_like(h)
    Wh = attention.W(h)
    sb = np.asarray(sub_batches, dtype=np.int64).reshape(-1, 2)
    sizes = sb[:, 1] - sb[:, 0]
    for n in np.unique(sizes):
        n = int(n)
        if n == 1:
            continue                                                         # train.py:165
        starts = torch.as_tensor(sb[sizes == n, 0])
        idx = starts[:, None] + torch.arange(n)[None, :]                     # (G,n) agent rows
        st = last4[idx]                                                      # (G,n,4)
        feat = pair_features(st[:, :, None, :], st[:, None, :, :])           # (G,n,n,3) [i,j]
        emb = feature_embedder(feat, None)                                   # (G,n,n,F)
        sigma = (emb * Wh[idx][:, None, :, :]).sum(-1)                       # <f_ij, Wh_j>
        eye = torch.eye(n, dtype=torch.bool)
        sigma = sigma.masked_fill(eye[None], -1000.0)                        # train.py:170
        a = torch.softmax(sigma, dim=2)
        S = S.index_put((idx.reshape(-1),), torch.bmm(a, h[idx]).reshape(-1, h.shape[1]))
    return S


# ----------------------------------------------------------------------------- train.py:245-269
class EncoderLstm(nn.Module):
    def __init__(self, hidden_size, n_layers=2):
        self.hidden_size = hidden_size
        super().__init__()
        self.embed = nn.Linear(4, self.hidden_size)
        self.lstm = nn.LSTM(self.hidden_size, self.hidden_size, num_layers=n_layers, batch_first=True)
        self.lstm_h = []

    def init_lstm(self, h, c):
        self.lstm_h = (h, c)

    def forward(self, obsv):
        bs = obsv.shape[0]
        obsv = self.embed(obsv)
        y, self.lstm_h = self.lstm(obsv.view(bs, -1, self.hidden_size), self.lstm_h)
        return y


# ----------------------------------------------------------------------------- train.py:272-316
class Discriminator(nn.Module):
    def __init__(self, n_next, hidden_dim, n_latent_code):
        super().__init__()
        self.lstm_dim = hidden_dim
        self.n_next = n_next
        self.obsv_encoder_lstm = nn.LSTM(4, hidden_dim, batch_first=True)
        self.obsv_encoder_fc = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.LeakyReLU(0.2),
                                             nn.Linear(hidden_dim // 2, hidden_dim // 2))
        self.pred_encoder = nn.Sequential(nn.Linear(n_next * 4, hidden_dim // 2), nn.LeakyReLU(0.2),
                                          nn.Linear(hidden_dim // 2, hidden_dim // 2))
        self.classifier = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.LeakyReLU(0.2),
                                        nn.Linear(hidden_dim // 2, 1))
        self.latent_decoder = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.LeakyReLU(0.2),
                                            nn.Linear(self.lstm_dim // 2, n_latent_code))

    def forward(self, obsv, pred):
        bs = obsv.size(0)
        lstm_h_c = (torch.zeros(1, bs, self.lstm_dim), torch.zeros(1, bs, self.lstm_dim))
        obsv_code, lstm_h_c = self.obsv_encoder_lstm(obsv, lstm_h_c)
        obsv_code = self.obsv_encoder_fc(obsv_code[:, -1])
        pred_code = self.pred_encoder(pred.reshape(-1, self.n_next * 4))
        both_codes = torch.cat([obsv_code, pred_code], dim=1)
        label = self.classifier(both_codes)            # raw score, no sigmoid (LSGAN)
        code_hat = self.latent_decoder(both_codes)
        return label, code_hat

    def load(self, backup):
        """Restores nn.Linear weights/biases only; the LSTM keeps the unrolled update
        (train.py:311-316)."""
        for m_from, m_to in zip(backup.modules(), self.modules()):
            if isinstance(m_to, nn.Linear):
                m_to.weight.data = m_from.weight.data.clone()
                if m_to.bias is not None:
                    m_to.bias.data = m_from.bias.data.clone()


# ----------------------------------------------------------------------------- train.py:320-335
class DecoderFC(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.LeakyReLU(0.2),
                                 nn.Linear(hidden_dim, hidden_dim // 2), nn.LeakyReLU(0.2),
                                 nn.Linear(hidden_dim // 2, hidden_dim // 4),
                                 nn.Linear(hidden_dim // 4, 2))

    def forward(self, h, s, z):
        return self.fc1(torch.cat([h, s, z], dim=1))


# ----------------------------------------------------------------------------- utils/linear_models.py:9-20
def predict_cv(obsv, n_next):
    n_past = obsv.shape[1]
    if n_past > 2:
        my_vel = (obsv[:, -1] - obsv[:, -3]) / 2.
    else:
        my_vel = (obsv[:, -1] - obsv[:, -2])
    for si in range(n_next):
        pred_hat = obsv[:, -1] + my_vel
        obsv = torch.cat((obsv, pred_hat.unsqueeze(1)), dim=1)
    return obsv[:, n_past:, :]


# ----------------------------------------------------------------------------- utils/parse_utils.py:11-76
class Scale(object):
    def __init__(self):
        self.min_x, self.max_x = +np.inf, -np.inf
        self.min_y, self.max_y = +np.inf, -np.inf
        self.sx, self.sy = 1, 1

    def calc_scale(self, keep_ratio=True):
        self.sx = 1 / (self.max_x - self.min_x)
        self.sy = 1 / (self.max_y - self.min_y)
        if keep_ratio:
            if self.sx > self.sy:
                self.sx = self.sy
            else:
                self.sy = self.sx

    def normalize(self, data, shift=True, inPlace=True):
        out = data if inPlace else np.copy(data)
        out[..., 0] = (data[..., 0] - self.min_x * shift) * self.sx
        out[..., 1] = (data[..., 1] - self.min_y * shift) * self.sy
        return out

    def denormalize(self, data, shift=True, inPlace=False):
        out = data if inPlace else np.copy(data)
        out[..., 0] = data[..., 0] / self.sx + self.min_x * shift
        out[..., 1] = data[..., 1] / self.sy + self.min_y * shift
        return out


def load_and_normalise(obsvs, preds, batches):
    """train.py:89-124: 4/5 scene split and keep-ratio min-max normalisation (in float32, like the
    reference which normalises the float32 npz arrays in place)."""
    obsvs = np.array(obsvs, dtype=np.float32, copy=True)
    preds = np.array(preds, dtype=np.float32, copy=True)
    the_batches = np.asarray(batches)
    train_size = max(1, (len(the_batches) * 4) // 5)
    n_train_samples = int(the_batches[train_size - 1][1])
    n_test_samples = obsvs.shape[0] - n_train_samples
    if n_test_samples == 0:
        n_test_samples = 1
        the_batches = np.array([the_batches[0], the_batches[0]])
    scale = Scale()
    scale.max_x = max(np.max(obsvs[:, :, 0]), np.max(preds[:, :, 0]))
    scale.min_x = min(np.min(obsvs[:, :, 0]), np.min(preds[:, :, 0]))
    scale.max_y = max(np.max(obsvs[:, :, 1]), np.max(preds[:, :, 1]))
    scale.min_y = min(np.min(obsvs[:, :, 1]), np.min(preds[:, :, 1]))
    scale.calc_scale(keep_ratio=True)
    obsvs = scale.normalize(obsvs)
    preds = scale.normalize(preds)
    return dict(obsv=torch.FloatTensor(obsvs), pred=torch.FloatTensor(preds), the_batches=the_batches,
                train_size=train_size, n_train_samples=n_train_samples, n_test_samples=n_test_samples,
                scale=scale, ss=scale.sx)


# ----------------------------------------------------------------------------- train.py:370-389, 392-560
class SocialWaysOracle:
    """Module construction order, predict(), train() step body and test() of train.py restated as
    an object (the reference keeps all of this in module globals)."""

    def __init__(self, n_next, hidden_size=64, lr_g=1e-4, lr_d=1e-3, n_unrolling_steps=1,
                 use_social=True, social="blockdiag", use_info_loss=True, loss_info_w=0.5,
                 n_latent_codes=2, use_l2_loss=False, use_variety_loss=False, loss_l2_w=0.5, variety_k=20):
        self.n_next = n_next
        self.hidden_size = hidden_size
        self.noise_len = hidden_size // 2                                    # train.py:81
        self.n_lstm_layers = 1                                               # train.py:82
        self.use_social = use_social                                         # train.py:83 (flag)
        self.social = social
        self.n_unrolling_steps = n_unrolling_steps
        self.use_info_loss = use_info_loss
        self.loss_info_w = loss_info_w
        self.n_latent_codes = n_latent_codes
        self.use_l2_loss = use_l2_loss                                       # train.py:67-69
        self.use_variety_loss = use_variety_loss     # False | True = train.py:527-536 as written | "fixed" = best-of-K
        self.variety_k = variety_k
        self.loss_l2_w = loss_l2_w
        # construction order fixes the RNG -> init mapping (train.py:370-385)
        self.encoder = EncoderLstm(hidden_size, self.n_lstm_layers)
        self.feature_embedder = EmbedSocialFeatures(3, hidden_size)
        self.attention = AttentionPooling(hidden_size, hidden_size)
        self.decoder = DecoderFC(hidden_size + hidden_size + self.noise_len)
        predictor_params = chain(self.attention.parameters(), self.feature_embedder.parameters(),
                                 self.encoder.parameters(), self.decoder.parameters())
        self.predictor_optimizer = opt.Adam(predictor_params, lr=lr_g, betas=(0.9, 0.999))
        self.D = Discriminator(n_next, hidden_size, n_latent_codes)
        self.D_optimizer = opt.Adam(self.D.parameters(), lr=lr_d, betas=(0.9, 0.999))
        self.mse_loss = nn.MSELoss()
        self.last = {}

    # -- train.py:392-432
    def predict(self, obsv_p, noise, n_next, sub_batches=[]):
        bs = obsv_p.shape[0]
        enc = self.encoder
        obsv_4d = get_traj_4d(obsv_p, [])
        enc.init_lstm(torch.zeros(self.n_lstm_layers, bs, enc.hidden_size),
                      torch.zeros(self.n_lstm_layers, bs, enc.hidden_size))
        enc(obsv_4d)
        if len(sub_batches) == 0:
            sub_batches = [[0, obsv_p.size(0)]]
        hT = enc.lstm_h[0].squeeze(0)        # reference .squeeze() (train.py:411); bs==1 safe here
        if self.use_social:
            if self.social == "faithful":
                features = SocialFeatures(obsv_4d, sub_batches)
                emb_features = self.feature_embedder(features, sub_batches)
                weighted_features = self.attention(emb_features, hT, sub_batches)
            else:
                weighted_features = social_pool_blockdiag(obsv_4d[:, -1], hT, sub_batches,
                                                          self.feature_embedder, self.attention)
        else:
            weighted_features = torch.zeros_like(hT)
        pred_4ds = []
        last_obsv = obsv_4d[:, -1]
        for ii in range(n_next):
            new_v = self.decoder(enc.lstm_h[0].view(bs, -1), weighted_features.view(bs, -1), noise).view(bs, 2)
            new_p = new_v + last_obsv[:, :2]
            last_obsv = torch.cat([new_p, new_v], dim=1)
            pred_4ds.append(last_obsv)
            enc(pred_4ds[-1])
        self.last["hT"], self.last["S"] = hT, weighted_features
        return torch.stack(pred_4ds, 1)

    # -- train.py:458-554: one packed batch
    def train_step(self, obsv, pred, sub_batches, zeros_val, ones_val, noise, ss=1.0, record=None, variety_noise=None):
        """Step body.  `zeros_val`/`ones_val` are the two label-noise scalars (train.py:471-472),
        `noise` the (B,noise_len) latent (train.py:473); the caller draws them so RNG streams can be
        shared with the implementation under test.  Returns the 9 MSE values in reference order
        [d_fake,d_info,d_real]x(unroll+1), [g_l2,g_fool,g_info] and the ADE/FDE partial sums."""
        D, mse = self.D, self.mse_loss
        n_next, nl = self.n_next, self.n_latent_codes
        bs = obsv.shape[0]
        obsv_4d, pred_4d = get_traj_4d(obsv, pred)
        zeros = torch.zeros(bs, 1) + zeros_val
        ones = torch.ones(bs, 1) * ones_val
        losses = []
        backup = None
        for u in range(self.n_unrolling_steps + 1):
            D.zero_grad()
            with torch.no_grad():
                pred_hat_4d = self.predict(obsv, noise, n_next, sub_batches)
            fake_labels, code_hat = D(obsv_4d, pred_hat_4d)
            d_loss_fake = mse(fake_labels, zeros)
            d_loss_info = mse(code_hat.squeeze(), noise[:, :nl])
            real_labels, code_hat = D(obsv_4d, pred_4d)
            d_loss_real = mse(real_labels, ones)
            d_loss = d_loss_fake + d_loss_real
            if self.use_info_loss:
                d_loss = d_loss + self.loss_info_w * d_loss_info
            d_loss.backward()
            if record is not None:
                record.setdefault("d_grads", []).append(
                    {k: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone())
                     for k, p in D.named_parameters()})
                if u == 0:
                    record["fake_labels0"] = fake_labels.detach().clone()
                    record["code_hat_real0"] = code_hat.detach().clone()
                    record["real_labels0"] = real_labels.detach().clone()
            self.D_optimizer.step()
            losses += [d_loss_fake.item(), d_loss_info.item(), d_loss_real.item()]
            if u == 0 and self.n_unrolling_steps > 0:
                backup = copy.deepcopy(D)
        D.zero_grad()
        self.predictor_optimizer.zero_grad()
        pred_hat_4d = self.predict(obsv, noise, n_next, sub_batches)
        gen_labels, code_hat = D(obsv_4d, pred_hat_4d)
        g_loss_l2 = mse(pred_hat_4d[:, :, :2], pred)
        g_loss_fooling = mse(gen_labels, ones)
        g_loss_info = mse(code_hat.squeeze(), noise[:, :nl])
        g_loss = g_loss_fooling
        if self.use_info_loss:
            g_loss = g_loss + self.loss_info_w * g_loss_info
        if self.use_l2_loss:                                                 # train.py:525-526
            g_loss = g_loss + self.loss_l2_w * g_loss_l2
        variety = None
        if self.use_variety_loss == "fixed":
            # What train.py:527-536 evidently means (NOT reference behaviour - the reference's loop is
            # buggy, see the branch below): KV rollouts with independent noise (sample 0 = this step's own noise),
            # per agent the smallest mean squared error over the samples, averaged over the batch.  torch.min
            # routes the gradient to the arg-min sample only.
            KV = self.variety_k
            vn = variety_noise.view(KV - 1, bs, noise.shape[1])
            l2_k = [((pred_hat_4d[:, :, :2] - pred) ** 2).mean(dim=(1, 2))]
            for k in range(1, KV):
                ph_k = self.predict(obsv, vn[k - 1], n_next, sub_batches)
                l2_k.append(((ph_k[:, :, :2] - pred) ** 2).mean(dim=(1, 2)))
            l2_k = torch.stack(l2_k)                                         # (KV, bs)
            l2_min, k_min = torch.min(l2_k, dim=0)
            variety = l2_min.mean()
            g_loss = g_loss + self.loss_l2_w * variety
            if record is not None:
                record["variety_kmin"] = k_min.clone()
                record["variety_l2"] = l2_k.detach().clone()
        elif self.use_variety_loss:
            # train.py:527-536 AS WRITTEN: KV=20 predict() calls with the SAME noise (identical values,
            # SURVEY §0.11), loss k compares AGENT k (not sample k) with its ground truth, and only the
            # last one (k = 19) is appended -> the "variety" term is the L2 of agent 19 alone.
            KV = 20
            pred_hat_k = self.predict(obsv, noise, n_next, sub_batches)      # the k = KV-1 call; the other 19 are dead
            variety = mse(pred_hat_k[KV - 1, :, :2], pred[KV - 1])
            g_loss = g_loss + self.loss_l2_w * variety
        if record is not None:
            pred_hat_4d.retain_grad()
        g_loss.backward()
        if record is not None:
            record["pred_hat_4d"] = pred_hat_4d.detach().clone()
            record["dpred_hat_4d"] = pred_hat_4d.grad.detach().clone()
            record["gen_labels"] = gen_labels.detach().clone()
            record["gen_code_hat"] = code_hat.detach().clone()
            record["hT"] = self.last["hT"].detach().clone()
            record["S"] = self.last["S"].detach().clone()
            if variety is not None:
                record["variety"] = float(variety.item())
            record["g_grads"] = {}
            for name, mod in (("attention", self.attention), ("feature_embedder", self.feature_embedder),
                              ("encoder", self.encoder), ("decoder", self.decoder)):
                for k, p in mod.named_parameters():
                    record["g_grads"][name + "." + k] = (torch.zeros_like(p) if p.grad is None
                                                         else p.grad.detach().clone())
        self.predictor_optimizer.step()
        losses += [g_loss_l2.item(), g_loss_fooling.item(), g_loss_info.item()]
        if self.n_unrolling_steps > 0:
            D.load(backup)
        with torch.no_grad():                                                # train.py:546-551
            err_all = torch.pow((pred_hat_4d[:, :, :2] - pred) / ss, 2).sum(dim=2).sqrt()
            ade_sum = err_all.sum().item() / n_next
            fde_sum = err_all[:, -1].sum().item()
        return losses, ade_sum, fde_sum

    # -- train.py:439-557: one epoch incl. the greedy scene packing (train.py:446-456)
    def train_epoch(self, data, batch_size, draw=None, record_steps=None):
        the_batches, train_size = data["the_batches"], data["train_size"]
        train_batches = the_batches[:train_size]
        train_ADE = train_FDE = 0.0
        all_losses, shapes = [], []
        acc, subs = 0, []
        for ii, batch_i in enumerate(train_batches):
            acc += int(batch_i[1] - batch_i[0])
            subs.append(batch_i)
            if ii >= train_size - 1 or \
                    acc + int(the_batches[ii + 1][1] - the_batches[ii + 1][0]) > batch_size:
                a, b = int(subs[0][0]), int(subs[-1][1])
                obsv, pred = data["obsv"][a:b], data["pred"][a:b]
                sb = np.asarray(subs) - a
                if draw is None:
                    zv = np.random.uniform(0, 0.1)                           # train.py:471
                    ov = np.random.uniform(0.9, 1.0)                         # train.py:472
                    noise = torch.rand(acc, self.noise_len)                  # train.py:473
                else:
                    zv, ov, noise = draw(acc)
                rec = {} if (record_steps is not None and len(all_losses) in record_steps) else None
                losses, ade, fde = self.train_step(obsv, pred, sb, zv, ov, noise, data["ss"], rec)
                if rec is not None:
                    record_steps[len(all_losses)] = rec
                all_losses.append(losses)
                shapes.append((acc, len(subs)))
                train_ADE += ade
                train_FDE += fde
                acc, subs = 0, []
        return (train_ADE / data["n_train_samples"], train_FDE / data["n_train_samples"],
                all_losses, shapes)

    # -- train.py:563-616
    def test(self, data, n_gen_samples=20, just_one=False, times=None, collect=None):
        tb = data["the_batches"][data["train_size"]:]
        ss = data["ss"]
        ade_avg = fde_avg = ade_min = fde_min = 0.0
        for ii, batch_i in enumerate(tb):
            obsv = data["obsv"][batch_i[0]:batch_i[1]]
            pred = data["pred"][batch_i[0]:batch_i[1]]
            bs = int(batch_i[1] - batch_i[0])
            with torch.no_grad():
                errs, preds_k = [], []
                linear_preds = predict_cv(obsv, self.n_next)
                for kk in range(n_gen_samples):
                    noise = torch.rand(bs, self.noise_len)
                    pred_hat_4d = self.predict(obsv, noise, self.n_next)
                    preds_k.append(pred_hat_4d.unsqueeze(0))
                    err = torch.pow((pred_hat_4d[:, :, :2] - pred) / ss, 2).sum(dim=2, keepdim=True).sqrt()
                    errs.append(err.unsqueeze(0))
                errs = torch.cat(errs)
                if collect is not None:
                    sc = data["scale"]
                    collect.append(dict(
                        timestamp=None if times is None else times[batch_i[0]],
                        obsvs=sc.denormalize(obsv[:, :, :2].numpy()),
                        preds_our=sc.denormalize(torch.cat(preds_k)[:, :, :, :2].numpy()),
                        preds_gtt=sc.denormalize(pred[:, :, :2].numpy()),
                        preds_lnr=sc.denormalize(linear_preds[:, :, :2].numpy())))
                fde_min += errs[:, :, -1].min(0, keepdim=True)[0].sum().item()
                ade_min += errs.mean(2).min(0, keepdim=True)[0].sum().item()
                fde_avg += errs[:, :, -1].mean(0, keepdim=True).sum().item()
                ade_avg += errs.mean(2).mean(0, keepdim=True).sum().item()
            if just_one:
                break
        n = data["n_test_samples"]
        return ade_avg / n, fde_avg / n, ade_min / n, fde_min / n

    # -- train.py:651-663
    def checkpoint(self, epoch):
        return {'epoch': epoch,
                'attentioner_dict': self.attention.state_dict(),
                'feature_embedder_dict': self.feature_embedder.state_dict(),
                'encoder_dict': self.encoder.state_dict(),
                'decoder_dict': self.decoder.state_dict(),
                'pred_optimizer': self.predictor_optimizer.state_dict(),
                'D_dict': self.D.state_dict(),
                'D_optimizer': self.D_optimizer.state_dict()}

    def load_state(self, sd):
        self.attention.load_state_dict(sd['attentioner_dict'])
        self.feature_embedder.load_state_dict(sd['feature_embedder_dict'])
        self.encoder.load_state_dict(sd['encoder_dict'])
        self.decoder.load_state_dict(sd['decoder_dict'])
        self.D.load_state_dict(sd['D_dict'])


# ----------------------------------------------------------------------------- synthetic inputs
def synth_dataset(n_scenes, agents, n_past=8, n_next=12, seed=1234):
    """SURVEY.md §8d generator: p0~U[0,10)^2, v = N(0,0.3^2) + cumsum_t N(0,0.05^2),
    track = p0 + cumsum_t v, float32; scenes contiguous.  `agents` is an int or a per-scene list."""
    rng = np.random.default_rng(seed)
    sizes = [agents] * n_scenes if np.isscalar(agents) else list(agents)
    N, T = int(np.sum(sizes)), n_past + n_next
    p0 = rng.uniform(0, 10, size=(N, 1, 2))
    v = rng.normal(0, 0.3, size=(N, 1, 2)) + np.cumsum(rng.normal(0, 0.05, size=(N, T, 2)), axis=1)
    track = (p0 + np.cumsum(v, axis=1)).astype(np.float32)
    ends = np.cumsum(sizes)
    batches = np.stack([ends - np.asarray(sizes), ends], axis=1).astype(np.int64)
    times = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    return dict(obsvs=track[:, :n_past], preds=track[:, n_past:], times=times, batches=batches)


# ----------------------------------------------------------------------------- calc_statistics.py:7-66
def _mean_l2(a, b, obsv_len):
    """D[k][i][j] = mean_t ||a[i,k,t] - b[j,k,t]|| over t >= obsv_len, in fp32 like the reference's
    numpy expression on fp32 inputs (calc_statistics.py:30-31, 58-59)."""
    diff = a[:, None, :, obsv_len:, :] - b[None, :, :, obsv_len:, :]           # (Na, Nb, nPed, T', 2)
    d = np.sqrt(np.sum(np.power(diff, 2), axis=-1)).mean(axis=-1)              # (Na, Nb, nPed)
    return np.transpose(d, (2, 0, 1)).astype(np.float64)


def compute_1nn(reals, fakes, obsv_len=2):
    """Leave-one-out 1-NN two-sample test per pedestrian (calc_statistics.py:7-44): returns
    [overall accuracy, real recall, fake recall].  Self-distance is 1000 (the matrix initial value)."""
    n_r, n_f, n_ped = reals.shape[0], fakes.shape[0], reals.shape[1]
    mixed = np.concatenate([reals, fakes], axis=0)
    D = _mean_l2(mixed, mixed, obsv_len)
    labels = np.concatenate([np.ones(n_r), -np.ones(n_f)])
    real_pos = fake_pos = 0
    for k in range(n_ped):
        Dk = D[k].copy()
        iu = np.triu_indices(n_r + n_f, 1)
        Dk[(iu[1], iu[0])] = Dk[iu]              # the reference fills (i<j) and mirrors it
        np.fill_diagonal(Dk, 1000.0)
        nn_ind = np.argmin(Dk, axis=1)
        same = labels[nn_ind] == labels
        real_pos += int(np.sum(same & (labels == 1)))
        fake_pos += int(np.sum(same & (labels == -1)))
    return np.array([(real_pos + fake_pos) / ((n_r + n_f) * n_ped), real_pos / (n_r * n_ped), fake_pos / (n_f * n_ped)])


def emd_cost_matrix(D):
    """calc_statistics.py:56-60 writes `D[ii, jj], D[jj, ii] = dij, dij` for EVERY (ii, jj) of a
    real x fake matrix, so later iterations overwrite earlier ones: the matrix that reaches the
    assignment solver is the LOWER triangle of the true distances mirrored to the upper one.
    Kept as is (needs n_reals == n_fakes, as in the reference's use)."""
    L = np.tril(D)
    return L + np.tril(D, -1).T


def compute_wasserstein(reals, fakes, obsv_len=2):
    """Per-pedestrian optimal assignment cost between real and generated futures, averaged
    (calc_statistics.py:47-66)."""
    import scipy.optimize as sopt
    n_r, n_ped = reals.shape[0], reals.shape[1]
    D = _mean_l2(reals, fakes, obsv_len)
    cost = 0.0
    for k in range(n_ped):
        Dk = emd_cost_matrix(D[k])
        r, c = sopt.linear_sum_assignment(Dk)
        cost += Dk[r, c].sum()
    return cost / (n_r * n_ped)
