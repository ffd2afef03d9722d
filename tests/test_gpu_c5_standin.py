"""BASELINE config 5 stand-in ("ETH+UCY all five splits, data-parallel"; the recordings are not in this image):
FIVE synthetic BIWI-format recordings of different density go, one after the other, through the dataset pipeline
(obsmat.txt -> parse -> windows -> scenes -> Scale, utils/parse_utils.py:231-320, 457-508, train.py:89-120) and two epochs of
train() (train.py:439-557) on TWO data-parallel ranks (scene-aligned shards, three gradient all-reduces per step; both ranks
share the test box's GPU; splits 1 and 3 use the library's direct exchange, the others the process group's all-reduce),
against the CPU oracle running the same recording in ONE process on identical draws: per-step MSE terms, epoch ADE / FDE,
replicas bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
SPLITS = [(11, 16, 60), (12, 24, 70), (13, 32, 60), (14, 20, 80), (15, 40, 60)]     # (seed, pedestrians, frames)
BATCH = 48


def _recording(tmp, k):
    import socialways_amd as sw
    seed, n_ped, n_frames = SPLITS[k]
    path = os.path.join(tmp, "split%d_obsmat.txt" % k)
    if not os.path.exists(path):
        fr, ids, pos, vel = sw.data.synth_crowd_frames(n_frames=n_frames, n_ped=n_ped, interval=6, seed=seed)
        sw.data.write_biwi_obsmat(path, fr, ids, pos, vel)
    return sw.data.biwi_to_npz(path, os.path.join(tmp, "split%d_r%d.npz" % (k, os.getpid())))


def _draws(k, epoch):
    rng = np.random.default_rng(1000 * k + epoch)
    return lambda bs: (float(rng.uniform(0, 0.1)), float(rng.uniform(0.9, 1.0)), torch.from_numpy(rng.random((bs, 32), dtype=np.float32)))


def _worker(rank, world, port, tmp, ret):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for k in range(len(SPLITS)):
        if k in (1, 3):
            os.environ["SW_ALLREDUCE"] = "direct"
        else:
            os.environ.pop("SW_ALLREDUCE", None)
        obsvs, preds, times, batches = _recording(tmp, k)
        data = sw.SceneDataset(obsvs, preds, batches, times, device="cuda:0")
        torch.manual_seed(100 + k)                     # every rank the same seed here; rank 0's replica is broadcast anyway
        tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", process_group=dist.group.WORLD)
        w0 = {kk: {a: b.cpu() for a, b in v.items()} for kk, v in tr.checkpoint().items() if kk.endswith("_dict")}
        eps = []
        for e in range(2):
            ade, fde, losses, sizes = tr.train_epoch(data, BATCH, draw=_draws(k, e))
            eps.append((ade, fde, np.asarray(losses).tolist(), [s[0] for s in sizes]))
        out.append((eps, w0 if rank == 0 else None, tr.G._flat_all.double().sum().item(), tr.D._flat.double().sum().item(),
                    tr._direct.status() if tr._direct is not None else 0, tr._direct is not None))
        tr.release_graphs()
        if tr._direct is not None:
            tr._direct.close()
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_five_recordings_two_ranks_follow_the_oracle(tmp_path):
    import torch.multiprocessing as mp
    import socialways_amd as sw
    import sw_oracle as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for k in range(len(SPLITS)):          # the five obsmat.txt files exist before the ranks start (they only read them)
        _recording(str(tmp_path), k)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    n_steps = 0
    for k in range(len(SPLITS)):
        r0, r1 = ret[0][k], ret[1][k]
        assert r0[2] == r1[2] and r0[3] == r1[3], "split %d: replicas diverged" % k
        assert r0[4] == 0 and r1[4] == 0 and r0[5] is (k in (1, 3))
        obsvs, preds, times, batches = _recording(str(tmp_path), k)
        assert int(np.diff(batches, axis=1).max()) <= 64
        odata = O.load_and_normalise(obsvs, preds, batches)
        orc = O.SocialWaysOracle(12, use_social=True)
        orc.load_state(r0[1])
        for e in range(2):
            oade, ofde, olosses, oshapes = orc.train_epoch(odata, BATCH, draw=_draws(k, e))
            ade, fde, losses, sizes = r0[0][e]
            assert sizes == [s_[0] for s_ in oshapes]
            n_steps += len(sizes)
            assert_close(np.asarray(losses), np.asarray(olosses), 5e-4, 5e-6, "split %d epoch %d: MSE terms" % (k, e))
            assert abs(ade - oade) < 1e-4 and abs(fde - ofde) < 1e-4, (k, e, ade, oade, fde, ofde)
            assert r1[0][e][0] == ade and r1[0][e][1] == fde
    assert n_steps >= 20
