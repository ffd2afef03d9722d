"""bench.py - GAN train steps/s of the Social Ways inner loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload m1|c2|c4] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the body of the reference's train() for one packed batch (train.py:458-554): generator
rollout with the pairwise social block, 2 discriminator updates, 1 generator update (LSGAN + InfoGAN
losses, n_unrolling_steps=1, use_social=True), the three Adam steps, D.load(backup) and the ADE/FDE
sums, on synthetic tracks already resident in HBM (label noise and z are drawn on the host and
copied each step, as the reference does).  Workload m1 (default, the metric's shape): 256 scenes x 8
agents x (8 obs + 12 pred) = 2048 agents per step per GPU.  N > 1: one process per GPU, every rank
trains on its own 256-scene shard of a 256*N-scene global batch, gradients all-reduced with RCCL
three times per step (weak scaling); `value` counts 256-scene batches processed per second by the
whole job.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {   # name -> (scenes per GPU step, agents per scene, To, Tp)
    "m1": (256, 8, 8, 12),     # BASELINE metric shape: --batch-size 2048
    "c2": (32, 8, 8, 12),      # BASELINE config 2: --batch-size 256
    "c4": (512, 64, 8, 12),    # dense crowd: 32768 agents, 2.1M pairs
}
PEAK_HBM_BPS = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / VALU fp32 peak
N_BATCHES = 8                  # distinct packed batches cycled through


def alg_flops(B, P, To, Tp):
    """Algorithmic work of SURVEY.md §8d (MAC = 2 FLOP, backward = 2 x forward)."""
    G = B * ((To + Tp) * (256 + 32768) + Tp * 41680) + B * 4096 + P * 6368
    Dd = B * (To * 17408 + 2048 + 1024 + 4 * Tp * 32 + 1024 + 4096 + 32 + 64)
    return dict(step=2.0 * (3 * G + 15 * Dd),
                # data-gradient pass of the decode loop: dX = W^T dY has the MAC count of the forward
                sw_dec_rollout_bwd=2.0 * B * (Tp * 41680 + (Tp - 1) * 33024),
                sw_dec_rollout_fwd=2.0 * B * (Tp * 41680 + (Tp - 1) * 33024),
                sw_enc_lstm_fwd=2.0 * B * To * 33024,
                sw_gen_wgrad=2.0 * B * ((To + Tp - 1) * 33024 + Tp * 41680))


def cpu_baseline(tracks, S, A, To, Tp, budget_s=12.0):
    """The CPU oracle (oracle/sw_oracle.py, block-diagonal social block, the reference's own call
    structure incl. its three predict() calls) timed on this host's cores on the same batch shape."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sw_oracle as O
    torch.manual_seed(0)
    data = O.load_and_normalise(tracks["obsvs"], tracks["preds"], tracks["batches"])
    orc = O.SocialWaysOracle(Tp, use_social=True, social="blockdiag")
    B = S * A
    sb = data["the_batches"][:S]
    obsv, pred = data["obsv"][:B], data["pred"][:B]
    noise = torch.rand(B, 32)
    orc.train_step(obsv, pred, sb, 0.05, 0.95, noise, data["ss"])       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(obsv, pred, sb, 0.05, 0.95, noise, data["ss"])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    return {"value": n / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of one %dx%d-agent packed batch (To=%d, Tp=%d), torch CPU fp32, "
                      "block-diagonal social block, reference call structure" % (n, S, A, To, Tp)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="m1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dominant", default="sw_dec_rollout_bwd", help="C-ABI call timed with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                         % (args.gpus, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)               # SW_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1 or os.environ.get("SW_FORCE_DIST", "") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SW_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    import socialways_amd as sw
    from socialways_amd import _lib as L
    S, A, To, Tp = WORKLOADS[args.workload]
    B = S * A
    P = S * A * A if A > 1 else 0
    torch.manual_seed(0)                      # identical replicas on every rank
    np.random.seed(0)
    tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, process_group=pg)
    tracks = sw.synth_tracks(S * N_BATCHES, A, To, Tp, seed=1234 + rank)
    data = sw.SceneDataset(tracks["obsvs"], tracks["preds"], tracks["batches"], device=dev)
    sb = np.stack([np.arange(S) * A, (np.arange(S) + 1) * A], axis=1).astype(np.int64)
    Bg = B * world
    last = [None]

    KG = int(os.environ.get("SW_BENCH_STEPS_PER_LAUNCH", "4"))   # steps per graph launch (step_many)

    def draw(i):
        a = (i % N_BATCHES) * B
        zv = np.random.uniform(0, 0.1)                         # train.py:471-473, same host RNG use
        ov = np.random.uniform(0.9, 1.0)
        noise = torch.rand(B, tr.noise_len)                    # host generator, copied to HBM inside the step
        return data.obsv[a:a + B], data.pred[a:a + B], zv, ov, noise

    def one_step(i):
        o, p, zv, ov, noise = draw(i)
        last[0] = tr.step(o, p, sb, zv, ov, noise, data.ss, global_B=Bg, out=False)

    def run_steps(i0, n):
        """n training steps, KG per graph launch where possible (identical work: see SocialWaysTrainer.step_many)."""
        i = i0
        while i < i0 + n:
            if KG > 1 and tr.use_graph and i + KG <= i0 + n:
                last[0] = tr.step_many([draw(i + j) for j in range(KG)], sb, data.ss, global_B=Bg, out=False)[-1]
                i += KG
            else:
                one_step(i)
                i += 1

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # untimed: the first 3 calls of a launch shape run eagerly / capture the graphs - both shapes (KG steps per
    # launch and single steps) are primed here so that no capture falls into the timed region, then W warm-up steps
    for rep in range(3):
        if KG > 1:
            run_steps(0, KG)
        one_step(rep)
    run_steps(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    # Roofline leg: the timed region replays hipGraphs, where no per-kernel event can be placed, so the
    # dominant kernel is timed right here with HIP events around its C-ABI call (on the launch stream)
    # over a few EAGER steps of the same workload; profiles/ holds the rocprofv3 trace of the graph run.
    n_ev = max(4, min(args.steps, 20))
    tr.use_graph = False
    for i in range(2):
        one_step(i)
    L.TIMING = {"names": {args.dominant, args.dominant + "_aux"}, "events": []}   # the step calls the _aux entry of the same kernel
    for i in range(n_ev):
        one_step(i)
    fence()
    timing, L.TIMING = L.TIMING, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(last[0]).all(), "non-finite losses"
    replicas_identical = None
    if world > 1:      # data-parallel replicas must hold bit-identical weights after the same all-reduced updates
        chk = torch.stack([tr.G._flat_all.double().sum(), tr.D._flat.double().sum()])
        hi, lo = chk.clone(), chk.clone()
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        replicas_identical = bool(torch.equal(hi, lo))

    if rank == 0:
        fl = alg_flops(B, P, To, Tp)
        kern_ms = [e0.elapsed_time(e1) for _, e0, e1 in timing["events"]]
        kern_s = float(np.mean(kern_ms)) * 1e-3 if kern_ms else float("nan")
        achieved = fl.get(args.dominant, float("nan")) / kern_s / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_%s.json" % args.workload)
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get(args.dominant, {}).get("hbm_bytes_per_launch")
        res = {
            "metric": "GAN train steps/sec (%d scenes x %d agents x %d+%d T per GPU step; fp32; social block on)" % (S, A, To, Tp),
            "value": args.steps * world / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d scenes x %d agents x (%d obs + %d pred) per GPU step = reference "
                                   "--batch-size %d; use_social=True, n_unrolling_steps=1, info loss on"
                                   % (args.workload, S, A, To, Tp, B),
                       "global_batch_scenes": S * world, "parallelism": "dp%d" % world, "steps_per_graph_launch": KG,
                       "collectives": (None if pg is None else "in-graph" if tr._graph_collectives else "between graph segments"),
                       "replicas_identical": replicas_identical,
                       "step_alg_gflop": fl["step"] / 1e9,
                       "step_frac_of_fp32_peak": fl["step"] / (dt / args.steps) / (PEAK_FP32_TFLOPS * 1e12)},
            "roofline": {"bound": "mfma", "kernel": args.dominant.replace("sw_", "") + "_kernel", "achieved": achieved,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "avg_launch_ms": kern_s * 1e3, "launches": len(kern_ms), "traffic": traffic,
                         # north_star also asks for the HBM view: measured bytes / launch time vs 8 TB/s
                         "hbm_GBps": (traffic / kern_s / 1e9) if traffic else None,
                         "hbm_frac": (traffic / kern_s / PEAK_HBM_BPS) if traffic else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(tracks, S if args.workload != "c4" else 16, A, To, Tp)
        import ctypes
        ctypes.CDLL(None).fflush(None)          # RCCL's version banner sits in the C stdio buffer: keep the JSON line last
        print(json.dumps(res), flush=True)
    if pg is not None:
        tr.release_graphs()                     # recorded collectives go before their communicator
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
