"""Toy-set mode-coverage statistics of calc_statistics.py: the leave-one-out 1-NN
two-sample test and the per-pedestrian assignment cost ("EMD") between real and generated futures.
The O(K^2 T) distance matrices are computed on the GPU (`sw_traj_dist`), the nearest-neighbour votes
with device reductions; the K x K assignment problems go to scipy's Hungarian solver on the host, as in
the reference (calc_statistics.py:62)."""
import os

import numpy as np
import torch

from . import _lib as L


def _dev(x, device):
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32)
    return t.to(device).contiguous()


def traj_dist(a, b, obsv_len=2, device="cuda"):
    """D[k,i,j] = mean_t ||a[i,k,t] - b[j,k,t]|| over t >= obsv_len; a (Na,nPed,T,2), b (Nb,nPed,T,2)."""
    a, b = _dev(a, device), _dev(b, device)
    L.require_gpu(a)
    Na, nPed, T = a.shape[0], a.shape[1], a.shape[2]
    if b.shape[1:] != a.shape[1:] or a.shape[3] != 2:
        raise ValueError("sample sets must be (N, nPed, T, 2) with equal nPed and T: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    D = torch.empty(nPed, Na, b.shape[0], device=a.device)
    L.call("sw_traj_dist", L.ptr(a), L.ptr(b), Na, b.shape[0], nPed, T, int(obsv_len), L.ptr(D), L.stream())
    return D


def compute_1nn(reals, fakes, obsv_len=2, device="cuda"):
    """calc_statistics.py:7-44 -> np.array([accuracy, real recall, fake recall])."""
    reals, fakes = _dev(reals, device), _dev(fakes, device)
    n_r, n_f, n_ped = reals.shape[0], fakes.shape[0], reals.shape[1]
    D = traj_dist(torch.cat([reals, fakes]), torch.cat([reals, fakes]), obsv_len, device)
    D.diagonal(dim1=1, dim2=2).fill_(1000.0)                     # self-distance = the matrix' initial value
    nn_ind = torch.argmin(D, dim=2)                              # first minimum, like np.argmin
    is_real = torch.arange(n_r + n_f, device=D.device) < n_r
    same = is_real[nn_ind] == is_real[None, :]
    real_pos = int((same & is_real[None, :]).sum())
    fake_pos = int((same & ~is_real[None, :]).sum())
    return np.array([(real_pos + fake_pos) / ((n_r + n_f) * n_ped), real_pos / (n_r * n_ped), fake_pos / (n_f * n_ped)])


def compute_wasserstein(reals, fakes, obsv_len=2, device="cuda"):
    """calc_statistics.py:47-66.  The reference's loop writes every distance to D[ii,jj] AND D[jj,ii]
    of the real x fake matrix, so what reaches the solver is the lower triangle mirrored upwards; kept
    for drop-in parity (requires as many fakes as reals, like the reference's use)."""
    import scipy.optimize as sopt
    reals, fakes = _dev(reals, device), _dev(fakes, device)
    if reals.shape[0] != fakes.shape[0]:
        raise ValueError("compute_wasserstein needs as many generated as real samples (calc_statistics.py:57-60)")
    D = traj_dist(reals, fakes, obsv_len, device)
    C = (torch.tril(D) + torch.tril(D, -1).transpose(1, 2)).double().cpu().numpy()
    cost = 0.0
    for k in range(C.shape[0]):
        r, c = sopt.linear_sum_assignment(C[k])
        cost += C[k][r, c].sum()
    return cost / (reals.shape[0] * reals.shape[1])


def calc_and_store_stats(main_dir, real_samples, n_past=2, n_next=2, stats_file=None, device="cuda", min_ped=6):
    """calc_statistics.py:70-125 without the plotting: for every `<main_dir>/<epoch>/*.npz` written by
    `SocialWaysTrainer.test(write_to_file=...)` (keys obsvs, preds_our) compare the K real samples with
    the first K generated ones; returns ({epoch: 1nn accuracy}, {epoch: EMD}) and writes the reference's
    `stats_1nn` / `stats_wst` arrays (epoch order) to `stats_file` if given.
    real_samples: (K, nPed, n_past+n_next, 2)."""
    real_samples = np.asarray(real_samples, dtype=np.float32)
    K = real_samples.shape[0]
    stats_1nn, stats_wst = {}, {}
    for dirpath, _, filenames in sorted(os.walk(main_dir)):
        cur = os.path.basename(dirpath)
        if not cur.isdigit():
            continue
        s1 = sw = 0.0
        n_files = 0
        for f in sorted(filenames):
            if "npz" not in f:
                continue
            fake = np.load(os.path.join(dirpath, f))
            obsvs, preds = fake["obsvs"], fake["preds_our"]
            n_ped = obsvs.shape[0]
            if n_ped < min_ped:
                continue
            fo = np.broadcast_to(obsvs[None], (K,) + obsvs.shape)
            fake_samples = np.concatenate((fo, preds[:K]), axis=2).astype(np.float32)
            real = real_samples.reshape(K, n_ped, n_past + n_next, 2)
            s1 += compute_1nn(real, fake_samples, n_past, device)[0]
            sw += compute_wasserstein(real, fake_samples, n_past, device)
            n_files += 1
        if n_files:
            stats_1nn[int(cur)], stats_wst[int(cur)] = s1 / n_files, sw / n_files
    if stats_file is not None:
        np.savez(stats_file, stats_1nn=[stats_1nn[k] for k in sorted(stats_1nn)],
                 stats_wst=[stats_wst[k] for k in sorted(stats_wst)])
    return stats_1nn, stats_wst
