"""The library's own gradient all-reduce (csrc/sw_comm.hip, socialways_amd/comm.py): W processes that share the
test box's ONE GPU map each other's exchange buffers through hipIpc and all-reduce flat fp32 buffers of the step's bucket
sizes - against the sum taken in rank order on the host, bit for bit, eagerly and replayed from a hipGraph; then the
data-parallel training step with SW_ALLREDUCE=direct against the same step on the process group's own all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZES = (27942, 86124, 4096, 1031, 86124, 2048, 27942)   # D's and G's packed gradients (train.py:379-385), round and odd sizes;
                                                        # alternating sizes change the chunking between consecutive calls


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ar_worker(rank, world, port, ret):
    import torch.distributed as dist
    from socialways_amd.comm import DirectAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = DirectAllReduce(dist.group.WORLD, "cuda:0", max(SIZES))
    ok, worst = True, 0.0
    gen = torch.Generator().manual_seed(100 + rank)
    for it in range(6):
        for n in SIZES:
            x = torch.randn(n, generator=gen) * (10.0 ** (it - 3))
            parts = [torch.empty(n) for _ in range(world)]
            dist.all_gather(parts, x)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p                       # the kernel's order: rank 0, 1, 2, ...
            g = x.cuda()
            ar(g)
            got = g.cpu()
            ok = ok and torch.equal(got, want)
            worst = max(worst, float((got - want).abs().max()))
    # the three buckets of a training step, recorded in ONE hipGraph and replayed (arguments fixed, epochs advance on the device)
    bufs = [torch.zeros(n, device="cuda") for n in (27939, 27939, 86122)]
    srcs = [torch.randn(n, generator=gen).cuda() for n in (27939, 27939, 86122)]
    wants = []
    for s_ in srcs:
        parts = [torch.empty(s_.numel()) for _ in range(world)]
        dist.all_gather(parts, s_.cpu())
        w = parts[0].clone()
        for p in parts[1:]:
            w += p
        wants.append(w)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for b_, s_ in zip(bufs, srcs):
                b_.copy_(s_)
                ar(b_)
    graph_ok = True
    for rep in range(5):
        g.replay()
        torch.cuda.synchronize()
        graph_ok = graph_ok and all(torch.equal(b_.cpu(), w) for b_, w in zip(bufs, wants))
    ret[rank] = (ok, worst, graph_ok, ar.status())
    ar.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,cached", [(2, False), (4, False), (8, False), (4, True)])
def test_direct_allreduce_equals_the_sum_in_rank_order(world, cached, monkeypatch):
    """cached: the exchange buffers in ordinary cached device memory (SW_COMM_CACHED=1) instead of uncached - the hand-off
    then rests on the kernel's release / acquire fences alone (workgroups of the processes sit on different XCDs)."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("SW_COMM_CACHED", "1" if cached else "0")
    ret = mp.Manager().dict()
    mp.spawn(_ar_worker, args=(world, _port(), ret), nprocs=world, join=True)
    for r in range(world):
        ok, worst, graph_ok, status = ret[r]
        assert status == 0, "rank %d: a wait on a peer timed out" % r
        assert ok, "rank %d: eager all-reduce differs from the rank-order sum (max |diff| %.3g)" % (r, worst)
        assert graph_ok, "rank %d: replayed all-reduces differ" % r


def _dp_worker(rank, world, port, ret, mode):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if mode in ("direct", "auto"):
        os.environ["SW_ALLREDUCE"] = mode
    else:
        os.environ.pop("SW_ALLREDUCE", None)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = sw.synth_tracks(24, [5, 1, 9, 16, 3, 2, 2, 2, 7, 8, 4, 6] * 2, 8, 12, seed=5)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    torch.manual_seed(3)
    tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", process_group=dist.group.WORLD)
    gen = torch.Generator().manual_seed(8)
    out = []
    for e in range(6):              # eager steps, capture, replays (direct: ONE graph with the three exchange kernels in it)
        z = torch.rand(data.n_train_samples, 32, generator=gen)
        ade, fde, losses, sizes = tr.train_epoch(data, data.n_train_samples, draw=lambda bs: (0.01 * (e + 1), 0.95, z))
        out.append((ade, fde, np.asarray(losses[0]).tolist()))
    ret[(mode, rank)] = (out, tr.G._flat_all.cpu().clone(), tr.D._flat.cpu().clone(), tr._graph_collectives,
                         tr._direct.status() if tr._direct is not None else 0, tr.exchange_probe)
    tr.release_graphs()
    if tr._direct is not None:
        tr._direct.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_auto_mode_probes_both_exchanges_and_trains_like_the_chosen_one():
    """SW_ALLREDUCE=auto: every rank builds the direct exchange, checks it against the group's all-reduce and times both;
    the choice is collective.  Here (gloo stages CUDA tensors through the host) the direct form must win, and the trajectory
    must equal the SW_ALLREDUCE=direct run bit for bit."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    for mode in ("auto", "direct"):
        mp.spawn(_dp_worker, args=(2, _port(), ret, mode), nprocs=2, join=True)
    for r in (0, 1):
        pr = ret[("auto", r)][5]
        assert pr["chosen"] == "direct" and len(pr["group_us"]) == 2 and sum(pr["direct_us"]) < sum(pr["group_us"]), pr
        assert ret[("auto", r)][0] == ret[("direct", r)][0]
        assert torch.equal(ret[("auto", r)][1], ret[("direct", r)][1]) and torch.equal(ret[("auto", r)][2], ret[("direct", r)][2])
    assert ret[("auto", 0)][5]["direct_us"] == ret[("auto", 1)][5]["direct_us"]        # the ranks decided on the same numbers


@pytest.mark.timeout(900)
def test_data_parallel_step_on_the_direct_allreduce_equals_the_process_groups():
    """2 ranks, scene-sharded ragged batches: with two ranks a sum has one order, so the trajectory on the direct all-reduce
    (recorded inside the step graph) must equal the one on the group's own all-reduce (graph segments) bit for bit."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    for mode in ("group", "direct"):
        mp.spawn(_dp_worker, args=(2, _port(), ret, mode), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[("direct", r)][4] == 0 and ret[("direct", r)][3] is True and ret[("group", r)][3] is False
        assert ret[("direct", r)][0] == ret[("group", r)][0], "losses / ADE / FDE differ on rank %d" % r
        assert torch.equal(ret[("direct", r)][1], ret[("group", r)][1]) and torch.equal(ret[("direct", r)][2], ret[("group", r)][2])
    assert torch.equal(ret[("direct", 0)][1], ret[("direct", 1)][1]) and torch.equal(ret[("direct", 0)][2], ret[("direct", 1)][2])


def _dp4_worker(rank, world, port, ret, mode):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if mode == "direct":
        os.environ["SW_ALLREDUCE"] = "direct"
    else:
        os.environ.pop("SW_ALLREDUCE", None)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # packed batches of 3 .. 5 scenes over 4 ranks: in most steps at least one rank has NO scene and takes part in the three
    # exchanges through _empty_step (plain exchange + separate update) while its peers run the fused exchange + Adam launch
    t = sw.synth_tracks(15, [9, 2, 7, 12, 3, 5, 8, 1, 6, 4, 10, 2, 7, 3, 5], 8, 12, seed=21)
    data = sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")
    torch.manual_seed(3)
    tr = sw.SocialWaysTrainer(12, use_social=True, device="cuda:0", process_group=dist.group.WORLD)
    gen = torch.Generator().manual_seed(8)
    out, empties = [], 0
    for e in range(3):
        draws = []
        ade, fde, losses, sizes = tr.train_epoch(data, 24, draw=lambda bs: (0.01 * (e + 1), 0.95, torch.rand(bs, 32, generator=gen)))
        out.append((ade, fde, np.asarray(losses).tolist()))
    ret[(mode, rank)] = (out, tr.G._flat_all.cpu().clone(), tr.D._flat.cpu().clone(),
                         tr._direct.status() if tr._direct is not None else 0)
    tr.release_graphs()
    if tr._direct is not None:
        tr._direct.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_four_ranks_with_idle_ranks_direct_exchange_follows_the_process_groups():
    """4 ranks and packed batches of a few scenes: ranks without a scene join the exchanges through _empty_step (the plain
    sw_allreduce_direct + a separate update) while their peers run sw_allreduce_direct_adam - the two forms must interoperate.
    Replicas stay bit-identical in each mode; the direct trajectory (rank-order sums) follows the process group's (gloo's
    order) within fp32 summation noise."""
    import torch.multiprocessing as mp
    from socialways_amd import data as D
    sb = np.asarray([[0, 9], [9, 11], [11, 18]])
    assert any(hi <= lo for lo, hi in D.shard_scenes(sb, 4)), "the test needs a rank without scenes"
    ret = mp.Manager().dict()
    for mode in ("group", "direct"):
        mp.spawn(_dp4_worker, args=(4, _port(), ret, mode), nprocs=4, join=True)
    for mode in ("group", "direct"):
        for r in range(1, 4):
            assert torch.equal(ret[(mode, r)][1], ret[(mode, 0)][1]) and torch.equal(ret[(mode, r)][2], ret[(mode, 0)][2]), \
                "%s: replicas diverged (rank %d)" % (mode, r)
            assert ret[(mode, r)][3] == 0
    a, b = ret[("direct", 0)], ret[("group", 0)]
    for e in range(3):
        np.testing.assert_allclose(a[0][e][2], b[0][e][2], rtol=2e-4, atol=1e-6)
        assert abs(a[0][e][0] - b[0][e][0]) < 1e-5 and abs(a[0][e][1] - b[0][e][1]) < 1e-5
    lr = 1e-3
    assert float((a[2] - b[2]).abs().max()) <= 2.2 * lr * 3 * 8 and float((a[1] - b[1]).abs().max()) <= 2.2 * 1e-4 * 3 * 8


def _timeout_worker(rank, world, port, ret):
    import torch.distributed as dist
    from socialways_amd.comm import DirectAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["SW_COMM_TIMEOUT_S"] = "0.5"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = DirectAllReduce(dist.group.WORLD, "cuda:0", 27942)
    g = torch.full((27942,), float(rank + 1), device="cuda")
    ar(g)
    torch.cuda.synchronize()
    first_ok = bool((g == 3.0).all()) and ar.status() == 0
    dist.barrier()
    untouched = later_untouched = True
    if rank == 0:                     # rank 1 never joins this call: the wait gives up after 0.5 s
        g2 = torch.full((27942,), 7.0, device="cuda")
        ar(g2)
        torch.cuda.synchronize()
        untouched = bool((g2 == 7.0).all())          # no stale / partial sum was written over the gradient
        st = ar.status()
        g3 = torch.full((27942,), 9.0, device="cuda")
        t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0[0].record()
        ar(g3)                                        # the exchange is dead: returns at once, writes nothing
        t0[1].record()
        torch.cuda.synchronize()
        later_untouched = bool((g3 == 9.0).all()) and t0[0].elapsed_time(t0[1]) < 100.0
    else:
        st = ar.status()
    dist.barrier()
    ret[rank] = (first_ok, untouched, later_untouched, st, ar.status_all())
    ar.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_a_timed_out_wait_publishes_nothing_and_the_status_is_collective():
    """rank skew beyond the time-out.  The rank whose wait gives up must not write a partly stale sum over its
    gradient (nor publish it to its peers with a valid tag), later calls on the dead exchange return at once, and
    status_all() tells EVERY rank (the healthy one too) that the epoch is invalid."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_timeout_worker, args=(2, _port(), ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0], "the first (complete) call must work"
    assert ret[0][1], "rank 0: the timed-out call wrote into the gradient buffer"
    assert ret[0][2], "rank 0: a call on the dead exchange wrote / waited"
    assert ret[0][3] == 1 and ret[1][3] == 0
    assert ret[0][4] == 1 and ret[1][4] == 1, "status_all must report the time-out on every rank"


def _asym_worker(rank, world, port, ret):
    import torch.distributed as dist
    from socialways_amd import _lib as L
    from socialways_amd.comm import DirectAllReduce, probe
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rank == 1:
        os.environ["SW_COMM_FAULT_INJECT"] = "2"        # this rank cannot map its peer's buffer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raised = None
    try:
        DirectAllReduce(dist.group.WORLD, "cuda:0", 27942)
    except L.SocialWaysHipError as e:
        raised = str(e)
    # the group's collectives still line up after the failure (nobody is left in a barrier / all_gather)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ar, rep = probe(dist.group.WORLD, "cuda:0", [27942, 86124])
    t2 = torch.tensor([float(rank + 1)])
    dist.all_reduce(t2)
    ret[rank] = (raised, float(t.item()), ar is None, rep.get("chosen"), rep.get("reason"), float(t2.item()))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_construction_failure_on_one_rank_raises_on_every_rank():
    """DirectAllReduce.__init__ is collective; a rank that fails locally (here: rank 1's peer mapping is refused)
    must not leave its peers in a barrier - every rank raises, and comm.probe falls back to the group on every rank."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_asym_worker, args=(2, _port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        raised, s1, none, chosen, reason, s2 = ret[r]
        assert raised is not None and "rank 1" in raised, (r, raised)
        assert s1 == 3.0 and s2 == 3.0
        assert none and chosen == "group" and reason and "rank 1" in reason
