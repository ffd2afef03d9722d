"""N > 1 path on CPU: 2 processes, gloo.  The data-parallel recipe of trainer.py - scene-aligned shards
(never split a scene), every rank normalises its loss by the GLOBAL batch size, packed gradient
buffers all-reduced with SUM - must reproduce the single-process gradients exactly (up to fp32
summation order).  The compute here is the CPU oracle; the GPU kernels get the same treatment in
trainer.step()."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sw_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _losses(o, obsv, pred, sb, zv, ov, noise, Bg):
    """D-update and G-update losses of train.py:484-494 / 512-523 as sums over the local rows / Bg."""
    obsv_4d, pred_4d = O.get_traj_4d(obsv, pred)
    with torch.no_grad():
        ph = o.predict(obsv, noise, 12, sb)
    fake, code = o.D(obsv_4d, ph)
    real, _ = o.D(obsv_4d, pred_4d)
    d_loss = ((fake - zv) ** 2).sum() / Bg + ((real - ov) ** 2).sum() / Bg + 0.5 * ((code - noise[:, :2]) ** 2).sum() / (2 * Bg)
    ph = o.predict(obsv, noise, 12, sb)
    gl, gc = o.D(obsv_4d, ph)
    g_loss = ((gl - ov) ** 2).sum() / Bg + 0.5 * ((gc - noise[:, :2]) ** 2).sum() / (2 * Bg)
    return d_loss, g_loss


def _flat_grads(o, d_loss, g_loss):
    dg = torch.autograd.grad(d_loss, list(o.D.parameters()))
    gp = [p for m in (o.attention, o.feature_embedder, o.encoder, o.decoder) for p in m.parameters()]
    gg = torch.autograd.grad(g_loss, gp, allow_unused=True)
    gg = [torch.zeros_like(p) if g is None else g for p, g in zip(gp, gg)]
    return torch.cat([g.reshape(-1) for g in dg]), torch.cat([g.reshape(-1) for g in gg])


def _worker(rank, world, port, ret):
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(0)                       # identical replicas + identical global RNG stream on every rank
    o = O.SocialWaysOracle(12, use_social=True)
    sizes = [3, 8, 1, 5, 16, 2, 7, 4, 9, 6]
    tr = O.synth_dataset(len(sizes), sizes, seed=5)
    data = O.load_and_normalise(tr["obsvs"], tr["preds"], tr["batches"])
    sb = np.asarray(tr["batches"])
    B = int(sb[-1, 1])
    noise = torch.rand(B, 32)                  # drawn for the whole packed batch, sliced per rank
    zv, ov = 0.04, 0.93
    lo, hi = sw.shard_scenes(sb, world)[rank]
    r0, r1 = int(sb[lo, 0]), int(sb[hi - 1, 1])
    d_l, g_l = _losses(o, data["obsv"][r0:r1], data["pred"][r0:r1], sb[lo:hi] - r0, zv, ov, noise[r0:r1], float(B))
    dgrad, ggrad = _flat_grads(o, d_l, g_l)
    dist.all_reduce(dgrad)
    dist.all_reduce(ggrad)
    loss = torch.stack([d_l.detach(), g_l.detach()])
    dist.all_reduce(loss)
    d_f, g_f = _losses(o, data["obsv"][:B], data["pred"][:B], sb, zv, ov, noise, float(B))
    dref, gref = _flat_grads(o, d_f, g_f)
    ok = (torch.allclose(dgrad, dref, rtol=2e-4, atol=1e-7) and torch.allclose(ggrad, gref, rtol=2e-4, atol=1e-7)
          and torch.allclose(loss, torch.stack([d_f.detach(), g_f.detach()]), rtol=1e-5))
    # distributed plumbing of the trainer itself on CPU: a rank without scenes still joins the 3 all-reduces
    t = sw.SocialWaysTrainer(12, use_social=True, device="cpu", process_group=dist.group.WORLD, fused_adam=False)
    w_before = t.D._flat.clone()
    out = t._empty_step()
    ok = ok and torch.equal(t.D._flat, w_before) and float(out.abs().sum()) == 0.0 and t.world == world
    # replicas seeded DIFFERENTLY are made identical at construction (rank 0's weights / optimizer state are broadcast)
    # and follow rank 0's RNG stream after sync_rng() (label-noise scalars and z of train.py:471-473)
    torch.manual_seed(100 + rank)
    np.random.seed(100 + rank)
    t2 = sw.SocialWaysTrainer(12, use_social=True, device="cpu", process_group=dist.group.WORLD, fused_adam=False)
    t2.sync_rng()
    sig = torch.cat([t2.G._flat_all.double().sum().view(1), t2.D._flat.double().sum().view(1),
                     torch.rand(3).double(), torch.tensor([np.random.uniform(0, 0.1)], dtype=torch.float64)])
    hi_, lo_ = sig.clone(), sig.clone()
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    ok = ok and torch.equal(hi_, lo_)
    ret[rank] = bool(ok), float((dgrad - dref).abs().max()), float((ggrad - gref).abs().max()), (lo, hi)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_gradients_equal_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok, ed, eg, shard = ret[r]
        assert ok, (r, ed, eg, shard)
    assert ret[0][3][1] == ret[1][3][0]        # contiguous scene shards
