"""Randomised data-parallel sweep: 2 processes (gloo, sharing cuda:0 - plumbing, not speed) train on scene-aligned shards of
random RAGGED datasets (batches with a single scene leave one rank empty: _empty_step) and must follow the single process:
epoch ADE / FDE / losses within fp32 summation order, replicas bit-identical.  python tools/dbg/fuzz_dp.py [n] [seed]"""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch


def _make(cfg):
    import socialways_amd as sw
    t = sw.synth_tracks(len(cfg["sizes"]), cfg["sizes"], cfg["To"], cfg["Tp"], seed=cfg["s_t"])
    return sw.SceneDataset(t["obsvs"], t["preds"], t["batches"], device="cuda:0")


def _epochs(tr, data, cfg):
    out = []
    for ep in range(3):
        torch.manual_seed(cfg["s_r"] + ep); np.random.seed(cfg["s_r"] + ep)
        ade, fde, losses, sizes = tr.train_epoch(data, cfg["bs"])
        out.append((ade, fde, np.asarray(losses)))
    return out


def _flat(tr):
    return torch.cat([p.detach().double().reshape(-1) for p in list(tr.G.parameters()) + list(tr.D.parameters())]).cpu()


def _worker(rank, world, port, cfg, ret):
    import torch.distributed as dist
    import socialways_amd as sw
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)
    tr = sw.SocialWaysTrainer(cfg["Tp"], hidden_size=cfg["H"], device="cuda:0", process_group=dist.group.WORLD, **cfg["kw"])
    tr.load_checkpoint(torch.load(cfg["ck"], map_location="cuda:0"))
    res = _epochs(tr, _make(cfg), cfg)
    ret[rank] = (res, _flat(tr))
    tr.release_graphs()
    dist.destroy_process_group()


def run(N=6, seed=0):
    import torch.multiprocessing as mp
    import socialways_amd as sw
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(N):
        amax = int(rng.choice([2, 8, 20, 64]))
        sizes = [int(rng.integers(1, amax + 1)) for _ in range(int(rng.choice([10, 30, 60])))]
        cfg = dict(sizes=sizes, To=int(rng.choice([3, 8])), Tp=int(rng.choice([2, 12])), H=int(rng.choice([64, 64, 128, 80])),
                   bs=int(rng.choice([8, 64, 256])), s_t=int(rng.integers(1 << 30)), s_r=int(rng.integers(1 << 30)),
                   kw=dict(use_social=bool(rng.random() < 0.85), n_unrolling_steps=int(rng.choice([0, 1, 2]))))
        t0 = time.perf_counter()
        torch.manual_seed(int(rng.integers(1 << 30)))
        one = sw.SocialWaysTrainer(cfg["Tp"], hidden_size=cfg["H"], device="cuda:0", **cfg["kw"])
        cfg["ck"] = "/tmp/fuzz_dp_%d.pt" % os.getpid()
        one.save(cfg["ck"], epoch=0)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(2, port, cfg, ret), nprocs=2, join=True)
        ref = _epochs(one, _make(cfg), cfg)
        w1 = _flat(one)
        same = bool(torch.equal(ret[0][1], ret[1][1]))
        dw = float((ret[0][1] - w1).abs().max())
        e = 0.0
        for ep in range(3):
            for r in (0, 1):
                ade, fde, l = ret[r][0][ep]
                e = max(e, abs(ade - ref[ep][0]) / max(ref[ep][0], 1e-9), abs(fde - ref[ep][1]) / max(ref[ep][1], 1e-9),
                        float(np.max(np.abs(l - ref[ep][2]) / (np.abs(ref[ep][2]) + 1e-5))))
        ok = same and e < 2e-3 and dw < 5e-3
        print("%s #%02d %s %d scenes / %d agents (max %d) To %d Tp %d H %d bs %d %s | replicas identical %s, vs one process: "
              "epochs %.1e, weights %.1e | %.1fs" % ("ok  " if ok else "FAIL", it, type(one).__name__, len(sizes), sum(sizes),
              max(sizes), cfg["To"], cfg["Tp"], cfg["H"], cfg["bs"], cfg["kw"], same, e, dw, time.perf_counter() - t0), flush=True)
        fails += 0 if ok else 1
        del one
    print("%d configurations, %d failures" % (N, fails))
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
