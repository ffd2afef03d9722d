"""bench.py - GAN train steps/s of the Social Ways inner loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload m1|c2|c4] [--scaling weak|strong]
                    [--no-cpu-baseline] [--no-other-workloads]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: the N ranks are started by self_launch() the same way)

One "step" = the body of the reference's train() for one packed batch (train.py:458-554): generator
rollout with the pairwise social block, 2 discriminator updates, 1 generator update (LSGAN + InfoGAN
losses, n_unrolling_steps=1, use_social=True), the three Adam steps, D.load(backup) and the ADE/FDE
sums, on synthetic tracks already resident in HBM (label noise and z are drawn on the host and
copied each step, as the reference does).  Workload m1 (default, the metric's shape): 256 scenes x 8
agents x (8 obs + 12 pred) = 2048 agents per step per GPU.

N > 1: one process per GPU, gradients all-reduced with RCCL three times per step.
  --scaling weak   (default) every rank trains on its own 256-scene shard of a 256*N-scene global batch;
                   `value` counts 256-scene batches processed per second by the whole job.
  --scaling strong ONE packed batch of --global-scenes scenes (default 2048 = 8 x m1) is sharded over the
                   ranks by data.shard_scenes (scene aligned); `value` = global steps/s, total work fixed.
Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  config.repeats          min / median / max ms_per_step over R further blocks of K steps (spread of this box),
  config.other_workloads  short legs of the other BASELINE shapes (c2 = --batch-size 256, c4 = dense crowd), the 1-rank
                          RCCL form of m1 and the K = 20 variety step; each the FASTEST of three regions of its n steps,
  roofline                the kernel that takes the most time per step, timed live with HIP events around every launch
                          of an eager pass (sw_kernel_timing; fp32 MFMA peak), `roofline.kernels` = the top-6 table,
                          `roofline.step_traffic` = HBM bytes per step from the committed PMC pass of these kernel sources,
  cpu_baseline            the CPU oracle on this host: block-diagonal port (all threads, 1 thread) and the
                          reference's own dense + per-agent-loop formulation at a reduced batch.
"""
import argparse
import ctypes
import glob
import hashlib
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
    # BASELINE config 3's SHAPE (ETH-hotel: <= 8 agents per scene; the recordings themselves are not in the image): one
    # packed batch of 2048 agents in scenes of 1..8 agents (data.ragged_scene_sizes), A = None marks the ragged layout
    "c3_ragged": (None, None, 8, 12),
}
OTHER_STEPS = {"m1": (100, 12), "c2": (100, 12), "c4": (24, 8), "c3_ragged": (100, 12)}     # (steps, warm-up) of a short leg
PEAK_HBM_BPS = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / VALU fp32 peak
N_BATCHES = 8                  # distinct packed batches cycled through
REPEATS = 5                    # further timed blocks of K steps (spread)


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


def kernel_alg_flops(B, P, To, Tp, one_launch_d=False, dfuse=None):
    """Algorithmic FLOPs PER STEP of each kernel of the step (all of its launches together), MAC = 2 FLOP, from the
    per-agent / per-pair MAC counts of SURVEY.md §8a (data-gradient passes = forward MACs, weight gradients = forward
    MACs).  U + 1 = 2 discriminator updates + the generator-phase D pass."""
    lstm, dec = 33024, 41680                              # EncoderLstm step (embed + LSTM), DecoderFC
    d_lstm, d_obs, d_br = 17408, 2048 + 1024, 4 * Tp * 32 + 1024 + 4096 + 32 + 64     # D: LSTM step, obs fc, one branch
    gen_rows = (To + Tp - 1) * lstm + Tp * dec
    soc = B * 4096 + P * 6368
    d_all = To * d_lstm + d_obs + 2 * d_br                 # a whole D pass on both branches (forward = data-gradient MACs)
    if dfuse is None:     # the generator-phase D pass runs inside the decode BPTT launch (ops.DFUSE, sw_dec_rollout_bwd_dfuse)
        from socialways_amd import ops as _ops
        dfuse = _ops.DFUSE
    g_phase = 2.0 * B * (To * d_lstm + d_obs + 2 * d_br)     # one branch forward + its heads backward
    rides = (B + 15) // 16 <= 128     # pass 1's observation LSTM rides in the decode launch while that leaves CUs idle (ops.D_OBS_MAX_TILES)
    if one_launch_d:      # sw_disc_update: pass 1 + pass 2 (forward + backward each); disc_fwd = the G phase only
        d_kernels = {"disc_update_kernel": 2.0 * B * (((d_obs + 2 * d_br) if rides else d_all) + d_all + 2 * d_all),
                     "disc_fwd_kernel": g_phase}
    else:                 # 3 disc_fwd launches (pass 1, pass 2, G phase) + 2 disc_bwd launches
        d_kernels = {"disc_fwd_kernel": 2.0 * B * (((d_obs + 2 * d_br) if rides else d_all) + d_all) + g_phase,
                     "disc_bwd_kernel": 2.0 * 2 * B * d_all}
    if dfuse:
        d_kernels["disc_fwd_kernel"] -= g_phase
        if d_kernels["disc_fwd_kernel"] <= 0:
            del d_kernels["disc_fwd_kernel"]
    return {
        **d_kernels,
        "enc_lstm_fwd_kernel": 2.0 * B * To * lstm,
        "enc_lstm_fwd8_kernel": 2.0 * B * To * lstm,      # the same launch on eight waves (up to one tile per CU)
        "enc_lstm_bwd_kernel": 2.0 * B * To * lstm,
        "dec_rollout_fwd_kernel": 2.0 * B * (Tp * dec + (Tp - 1) * lstm)
                                  + (2.0 * B * To * d_lstm if rides else 0.0),   # + D's first obs LSTM (rides here)
        "dec_rollout_bwd_kernel": 2.0 * B * (Tp * dec + (Tp - 1) * lstm) + (g_phase if dfuse else 0.0),   # + the G-phase D pass
        "social_pool_fwd_kernel": 2.0 * soc,
        "social_pool_bwd_rows_kernel": 2.0 * 2 * soc,     # recomputes the pair MLP + its data gradients
        "social_pool_bwd_kernel": 2.0 * 3 * soc,          # ... + the pair-MLP weight gradients in registers
        # 3 launches: two D passes + the generator's (incl. the social block's rows when they are deferred)
        "wgrad_partial_kernel": 2.0 * (2 * B * (To * d_lstm + d_obs + 2 * d_br) + B * gen_rows + soc),
    }


def alg_bytes(B, To, Tp):
    """Compulsory HBM bytes of a step: tracks + z + outputs, parameter / gradient / Adam traffic."""
    return B * ((To + Tp) * 8 + 32 * 4 + Tp * 16) + 32 * (86122 + 2 * 27939)


def kernel_src_sha16():
    """Identity of the kernel sources a PMC pass was taken with (profiles/*_pmc_*.json carries the same)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "socialways_amd", "csrc", "*.h*"))):
        # the wide path's kernels (hidden sizes > 64) and the data-parallel exchange are in no measured single-process step
        if os.path.basename(f) not in ("sw_wide.hip", "sw_comm.hip"):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(workload, weak=True):
    """The committed PMC pass of this workload (profiles/r*_pmc_<workload>.json, newest round first) IF it was taken with
    exactly these kernel sources (sha256 of csrc/*.h*); else None.  It carries HBM bytes per launch / per step and the
    EXECUTED matrix FLOP per launch / per step (64 x SQ_VALU_MFMA_BUSY_CYCLES, tools/pmc_traffic.py)."""
    sha = kernel_src_sha16()
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)), reverse=True):
        rec = json.load(open(pmc))
        meta = rec.get("_meta", {})
        if meta.get("kernel_src_sha16") == sha and weak:
            rec["_file"] = os.path.relpath(pmc, ROOT)
            return rec
    return None


def _time_oracle(orc, obsv, pred, sb, noise, ss, budget_s, max_steps):
    orc.train_step(obsv, pred, sb, 0.05, 0.95, noise, ss)               # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(obsv, pred, sb, 0.05, 0.95, noise, ss)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= max_steps:
            return n, dt


def cpu_baseline(tracks, S, A, To, Tp, budget_s=10.0):
    """The CPU oracle (oracle/sw_oracle.py, the reference's own call structure incl. its three predict() calls per
    step) timed on this host's cores, bounded samples of the same workload:
      value             block-diagonal social block - the fair CPU baseline - at the fastest of the
                        sampled thread counts {1, 16, all}; `by_threads` lists every sample incl. the 1-thread figure;
      faithful          the reference's formulation - dense B x B pair tensors + the per-agent Python loop
                        (train.py:153-175, 229-241), O(B^3) - on 64 scenes x 8 agents (B = 512): one m1-size step of
                        it took ~295 s on 8 cores in the survey container and is not run here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sw_oracle as O
    torch.manual_seed(0)
    data = O.load_and_normalise(tracks["obsvs"], tracks["preds"], tracks["batches"])
    B = S * A
    sb = data["the_batches"][:S]
    obsv, pred = data["obsv"][:B], data["pred"][:B]
    noise = torch.rand(B, 32)
    threads = torch.get_num_threads()
    orc = O.SocialWaysOracle(Tp, use_social=True, social="blockdiag")
    # torch's CPU kernels do not scale to all cores of a large host on tensors this small (measured here: 1 thread
    # beats 128): a few thread counts are sampled and the FASTEST is the reported baseline, the rest are listed
    variants = {}
    for nt in sorted({1, min(16, threads), threads}):
        torch.set_num_threads(nt)
        try:
            n, dt = _time_oracle(orc, obsv, pred, sb, noise, data["ss"], budget_s / 2, 20)
        finally:
            torch.set_num_threads(threads)
        variants[nt] = (n / dt, n)
    best = max(variants, key=lambda k: variants[k][0])
    res = {"value": variants[best][0], "unit": "steps/s", "cores": best, "kind": "port",
           "sample": "%d steps of one %dx%d-agent packed batch (To=%d, Tp=%d), torch CPU fp32, block-diagonal social "
                     "block, reference call structure; fastest of the sampled thread counts" % (variants[best][1], S, A, To, Tp),
           "by_threads": {str(k): {"value": v[0], "steps": v[1]} for k, v in variants.items()}, "host_threads": threads,
           "host_cores": os.cpu_count()}
    Sf = min(S, 512 // A) if A <= 512 else 1
    Bf = Sf * A
    orf = O.SocialWaysOracle(Tp, use_social=True, social="faithful")
    nf, dtf = _time_oracle(orf, obsv[:Bf], pred[:Bf], sb[:Sf], noise[:Bf], data["ss"], budget_s, 10)
    res["faithful"] = {"value": nf / dtf, "unit": "steps/s", "cores": threads,
                       "sample": "%d steps of a %dx%d-agent batch (B = %d; dense B^2 pair tensors + per-agent loop, O(B^3): "
                                 "not comparable with `value` at B = %d)" % (nf, Sf, A, Bf, B)}
    return res


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher's environment (WORLD_SIZE unset): re-run this command line as
    N ranks of ONE node under torch.distributed.run - one process per GPU, rendezvous on 127.0.0.1 at a free port -
    and return its exit code; rank 0 of that job prints the JSON line.  RCCL needs every rank on its own device, so a
    box with fewer than N GPUs is refused here (SW_BENCH_BACKEND=gloo lets ranks share devices: a plumbing rehearsal,
    never a measurement)."""
    import socket
    import subprocess
    backend = os.environ.get("SW_BENCH_BACKEND", "nccl")
    if backend == "nccl" and "--launch-check" not in sys.argv and torch.cuda.device_count() < n:
        print("bench.py: --gpus %d needs %d visible GPUs (found %d); RCCL ranks cannot share a device"
              % (n, n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")      # the launcher would set 1: z is drawn on the host every step
    return subprocess.call(cmd, env=env)


def launch_check(world, rank, args):
    """--launch-check: the N-rank launch, rendezvous, barrier and max-over-ranks plumbing of this file with NO training
    step (runs without a GPU, gloo): rank 0 prints one JSON line with value null - a rehearsal of the driver's command,
    not a measurement."""
    backend = os.environ.get("SW_BENCH_BACKEND", "gloo" if not torch.cuda.is_available() else "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            torch.distributed.init_process_group(backend)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        torch.distributed.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ranks_seen = int(t.item())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    else:
        ranks_seen = 1
    if rank == 0:
        print(json.dumps({"metric": "launch check (no training step ran)", "value": None, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "launch_check": True, "backend": backend,
                          "ranks_seen": ranks_seen}), flush=True)
    return 0


class Leg:
    """One workload on this rank: trainer, resident synthetic batches and the stepping closures."""

    def __init__(self, name, dev, pg, world, rank, scaling, global_scenes, KG, **trainer_kw):
        import socialways_amd as sw
        self.name, self.world, self.KG = name, world, KG
        S, A, To, Tp = WORKLOADS[name]
        self.To, self.Tp, self.A = To, Tp, A
        self.strong = scaling == "strong"
        torch.manual_seed(0)                      # identical replicas on every rank
        np.random.seed(0)
        self.tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev, process_group=pg, **trainer_kw)
        if A is None:      # ragged layout: every packed batch has the same scene sizes (one graph layout), different tracks
            sizes = sw.ragged_scene_sizes(2048, 8, seed=77)
            tracks = sw.synth_tracks(len(sizes) * N_BATCHES, sizes * N_BATCHES, To, Tp, seed=1234 + rank)
            self.tracks = tracks
            self.data = sw.SceneDataset(tracks["obsvs"], tracks["preds"], tracks["batches"], device=dev)
            self.S_local, self.B = len(sizes), int(np.sum(sizes))
            self.Bg, self.S_global, self.row0, self.stride = self.B * world, len(sizes) * world, 0, self.B
            self.P = int(sum(a * a for a in sizes if a > 1))
            self.sb = np.asarray(tracks["batches"][:len(sizes)], dtype=np.int64)
            self.sizes = sizes
            self.last = None
            self._zring = [torch.empty(self.B, self.tr.noise_len) for _ in range(2 * max(KG, 1) + 2)]
            self._zi = 0
            return
        if scaling == "strong":
            # the same global dataset on every rank; this rank's rows = its scene-aligned shard of every packed batch
            Sg = global_scenes
            tracks = sw.synth_tracks(Sg * N_BATCHES, A, To, Tp, seed=1234)
            sb_g = np.stack([np.arange(Sg) * A, (np.arange(Sg) + 1) * A], axis=1).astype(np.int64)
            lo, hi = sw.shard_scenes(sb_g, world)[rank]
            self.S_local, self.Bg, self.S_global = hi - lo, Sg * A, Sg
            self.row0, self.stride = lo * A, Sg * A
        else:
            tracks = sw.synth_tracks(S * N_BATCHES, A, To, Tp, seed=1234 + rank)
            self.S_local, self.Bg, self.S_global = S, S * A * world, S * world
            self.row0, self.stride = 0, S * A
        self.tracks = tracks
        self.data = sw.SceneDataset(tracks["obsvs"], tracks["preds"], tracks["batches"], device=dev)
        Sl = self.S_local
        self.B = Sl * A
        self.P = Sl * A * A if A > 1 else 0
        self.sb = np.stack([np.arange(Sl) * A, (np.arange(Sl) + 1) * A], axis=1).astype(np.int64)
        self.last = None
        # z is drawn into a ring of preallocated host buffers (same generator calls, same values as torch.rand(B, Z)):
        # a fresh 256 KB tensor per step went through the allocator 2400 times a second
        self._zring = [torch.empty(self.Bg if self.strong else self.B, self.tr.noise_len) for _ in range(2 * max(KG, 1) + 2)]
        self._zi = 0

    def z_resident(self, n=16):
        """From here on z (train.py:473) is an input that is ALREADY in HBM when a timed region starts (the bench contract's
        definition of the inputs): a ring of n device tensors drawn once (device generator), handed to the trainer as they
        are - the staging kernel reads them through their address (sw_stage_step_zdev), no PCIe read of z inside the step."""
        assert not self.strong
        self._zdev = [torch.rand(self.B, self.tr.noise_len, device=self.data.obsv.device) for _ in range(n)]

    def z_host(self):
        """Back to z drawn on the host every step and pulled from pinned memory inside the step (the PCIe-inclusive form)."""
        self._zdev = None

    def draw(self, i):
        a = (i % N_BATCHES) * self.stride + self.row0
        zv = np.random.uniform(0, 0.1)                         # train.py:471-473, same host RNG use
        ov = np.random.uniform(0.9, 1.0)
        if getattr(self, "_zdev", None):
            self._zi += 1
            return self.data.obsv[a:a + self.B], self.data.pred[a:a + self.B], zv, ov, self._zdev[self._zi % len(self._zdev)]
        buf = self._zring[self._zi % len(self._zring)]
        self._zi += 1
        torch.rand(buf.shape, out=buf)                         # train.py:473, host generator; copied to HBM inside the step
        noise = buf[self.row0:self.row0 + self.B] if self.strong else buf   # strong: drawn for the whole packed batch and sliced
        return self.data.obsv[a:a + self.B], self.data.pred[a:a + self.B], zv, ov, noise

    def one_step(self, i):
        o, p, zv, ov, noise = self.draw(i)
        self.last = self.tr.step(o, p, self.sb, zv, ov, noise, self.data.ss, global_B=self.Bg, out=False)

    def plan(self, n, cold=False):
        """Launch sizes for n consecutive steps (identical work in any split: SocialWaysTrainer.step_many).  `cold`: the GPU
        is idle (right behind a fence) - ONE single-step launch gets it going after one step's host preparation, the rest
        follows in launches of up to KG steps that the host prepares while the GPU works (a launch boundary costs ~13 us, the
        fence at the end of a hipGraph; [1, 19] for the driver's 20-step region: 0.3700 ms per step where [1, 1, 2, 4, 4, 4, 4]
        gave 0.3728)."""
        if not self.tr.use_graph or self.KG <= 1:
            return [1] * n
        if not getattr(self, "_zdev", None):
            # z drawn on the host per step (train.py:473): preparing a step costs ~0.15 ms (4 MB of z at c4: ~2 ms), so the
            # launches stay small - 1, 1, 2 behind a fence, then 4 steps each
            out = [1, 1, 2] if cold and n >= 8 else []
            n -= sum(out)
            out += [4] * (n // 4) + [1] * (n % 4)
            return out
        if os.environ.get("SW_BENCH_RAMP") and cold:          # experiments: explicit first launches
            out = [int(x) for x in os.environ["SW_BENCH_RAMP"].split(",")]
            n -= sum(out)
        else:
            out = [1] if cold and n > 1 else []
            n -= len(out)
        while n >= self.KG:
            out.append(self.KG)
            n -= self.KG
        if n > 0:
            out.append(n)
        return out

    def run_steps(self, i0, n, cold=False):
        i, tr = i0, self.tr
        for k in self.plan(n, cold):
            if k > 1:
                self.last = tr.step_many([self.draw(i + j) for j in range(k)], self.sb, self.data.ss, global_B=self.Bg,
                                         out=False)[-1]
            else:
                self.one_step(i)
            i += k

    def prime(self, counts=()):
        """Untimed: the first 3 calls of a launch shape run eagerly / capture the graphs - every launch size the regions of
        `counts` steps will use is primed so that no capture falls into a timed region, and called twice more so that BOTH
        alternating executables of a shape have been launched once (the first launch of an executable costs ~2 ms on this
        runtime: the first timed region used to carry two of them)."""
        shapes = {1, self.KG}
        for j, n in enumerate(counts):      # counts = (steps of a timed region, untimed warmup steps in front of it)
            shapes |= set(self.plan(n, cold=(j == 0)))
        shapes = sorted(shapes, reverse=True)
        for rep in range(5):
            for k in shapes:
                if k > 1:
                    self.last = self.tr.step_many([self.draw(j) for j in range(k)], self.sb, self.data.ss, global_B=self.Bg,
                                                  out=False)[-1]
                else:
                    self.one_step(rep)

    def timed(self, fence, i0, steps):
        fence()
        t0 = time.perf_counter()
        self.run_steps(i0, steps, cold=True)
        fence()
        return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="m1", choices=[k for k in sorted(WORKLOADS) if WORKLOADS[k][1] is not None])   # c3_ragged: a side leg only
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-scenes", type=int, default=2048, help="--scaling strong: scenes of the ONE global packed batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg")
    ap.add_argument("--launch-check", action="store_true",
                    help="only launch / rendezvous / reduce over the N ranks (no GPU needed), print a JSON line with value null")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))        # plain `python bench.py --gpus N`: start the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))
    if args.launch_check:
        return launch_check(world, rank, args)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)               # SW_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg, backend = None, None
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

    from socialways_amd import _lib as L
    KG = int(os.environ.get("SW_BENCH_STEPS_PER_LAUNCH", "32"))   # most steps per graph launch (step_many; Leg.plan)
    os.environ.setdefault("SW_MAX_GRAPHS", "16")                  # launch sizes x input modes of one layout

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        return x

    import gc
    leg = Leg(args.workload, dev, pg, world, rank, args.scaling, args.global_scenes, KG)
    # The headline step is the reference's: z = torch.rand(bs, noise_len) drawn on the HOST every step and copied
    # to the device inside the step (train.py:473); the tracks are resident in HBM.  SW_BENCH_HOST_Z=0 makes the all-resident
    # form (z a ring of device tensors drawn beforehand) the headline instead; by default it is the secondary figure
    # config.inputs_resident.  --scaling strong always draws z for the global batch on the host and slices it.
    HOST_Z = os.environ.get("SW_BENCH_HOST_Z", "1") != "0" or args.scaling == "strong"
    RESIDENT = "resident in HBM when the timed region starts (tracks; z: a ring of 16 device tensors drawn beforehand)"
    if not HOST_Z:
        leg.z_resident()
    tr, To, Tp, A = leg.tr, leg.To, leg.Tp, leg.A
    # host-side noise sources of a 20-step window: the collector (a gen-2 pass over torch's module graph is ~10 ms) is off
    # inside timed regions; the noise ring (Leg) keeps the allocator out of the loop.  The collection runs BEFORE the
    # priming and warmup steps: tens of idle milliseconds right in front of the timed region cost its first launches ~0.4 ms
    gc.collect()
    gc.disable()
    UNTIMED_BEFORE = 5 * sum({1, KG} | set(leg.plan(args.steps, cold=True)) | set(leg.plan(args.warmup)))
    leg.prime((args.steps, args.warmup))
    # The runtime's one-time stalls (16-55 ms host blocks seen in the FIRST timed region of a young process in ~1 of 10 runs,
    # never in 180 later regions: round-3 per-launch host / GPU time stamps) are let happen in untimed steps: ~0.4 s of the same graph launches
    # in front of the W warmup steps.  Reported as config.settle_steps.
    SETTLE = int(os.environ.get("SW_BENCH_SETTLE_STEPS", "1000")) // KG * KG
    leg.run_steps(0, SETTLE)
    leg.run_steps(0, args.warmup)
    # ... and two untimed REHEARSALS of the region itself (same launch sizes behind a fence): a graph executable that has not
    # been launched for ~0.4 s (the settle steps use other launch sizes) replays ~70 us slower the first time - 3.6 us per step
    # of a 20-step region (measured: first region 0.3750, the following five 0.3701 .. 0.3711)
    REHEARSALS = 2
    UNTIMED_BEFORE += SETTLE + args.warmup + REHEARSALS * args.steps
    for r in range(REHEARSALS):
        leg.timed(fence, args.warmup, args.steps)
    dt = max_over_ranks(leg.timed(fence, args.warmup, args.steps))          # THE timed region: exactly K steps
    reps = [max_over_ranks(leg.timed(fence, args.warmup + (r + 1) * args.steps, args.steps)) for r in range(REPEATS)]
    if os.environ.get("SW_BENCH_VERBOSE"):
        print("regions (ms per step): first %.4f, then %s" % (1e3 * dt / args.steps, ["%.4f" % (1e3 * r / args.steps) for r in reps]),
              file=sys.stderr)
    # a sustained leg (>= ~2 s of back-to-back steps): long enough for an external GPU-busy sampler to see the device
    sustained = None
    if not args.no_sustained:
        n_sus = 1 + max(args.steps, int(2.2 / max(dt / args.steps, 1e-5))) // KG * KG      # [1, KG, KG, ...]: no new launch size
        d_sus = max_over_ranks(leg.timed(fence, 0, n_sus))
        sustained = {"steps": n_sus, "seconds": d_sus, "steps_s": n_sus * (world if args.scaling == "weak" else 1) / d_sus,
                     "ms_per_step": 1e3 * d_sus / n_sus}
    # the PCIe-inclusive form of the same K steps: z drawn on the host every step (train.py:473), copied into the pinned slot and
    # pulled inside the step (small launches: the host needs ~0.15 ms per step to prepare)
    pcie = None
    if not HOST_Z:
        leg.z_host()
        leg.prime((args.steps, args.warmup))
        leg.run_steps(0, args.warmup)
        d_pc = min(max_over_ranks(leg.timed(fence, args.warmup + r * args.steps, args.steps)) for r in range(3))
        pcie = {"steps_s": args.steps * (world if args.scaling == "weak" else 1) / d_pc, "ms_per_step": 1e3 * d_pc / args.steps,
                "what": "z drawn on the host per step (train.py:473) and pulled from pinned memory inside the step; fastest of 3 regions"}
        leg.z_resident()
    # ... and, when the headline is the reference's host-drawn z, the all-resident form of the same K steps as the secondary figure
    resident = None
    if HOST_Z and args.scaling == "weak":
        leg.z_resident()
        leg.prime((args.steps, args.warmup))
        leg.run_steps(0, args.warmup)
        d_rs = min(max_over_ranks(leg.timed(fence, args.warmup + r * args.steps, args.steps)) for r in range(3))
        resident = {"steps_s": args.steps * world / d_rs, "ms_per_step": 1e3 * d_rs / args.steps,
                    "what": "z too resident in HBM (a ring of 16 device tensors drawn beforehand, read by address: no draw, no "
                            "PCIe read inside the step); fastest of 3 regions"}
        leg.z_host()
    gc.enable()
    # Roofline leg: the timed region replays hipGraphs, where no per-kernel event can be placed, so EVERY kernel launch
    # of a few EAGER steps of the same workload is bracketed by HIP events on its launch stream (sw_kernel_timing); a
    # spin kernel queued in front lets the host run ahead, so the launches reach the GPU back to back and the event
    # intervals hold no host gaps.  profiles/ holds the rocprofv3 trace of the graph-replayed run for comparison.
    n_ev = max(4, min(args.steps, 20))
    tr.use_graph = False
    for i in range(2):
        leg.one_step(i)
    # inputs of the timed eager steps are placed on the device beforehand: a host-to-device copy inside the pass would
    # make the host wait for the stream and the launches behind it would reach an idle GPU one by one
    from socialways_amd import ops as sw_ops
    scenes = sw_ops.SceneIndex.get(leg.sb, leg.B, dev)
    ins = []
    for i in range(n_ev):
        o, p_, zv, ov, nz = leg.draw(i)
        ins.append((o.contiguous(), p_.contiguous(), torch.tensor([zv, ov], dtype=torch.float32).to(dev),
                    tr._pad_z(nz.to(dev)).contiguous()))
    part = torch.zeros(tr.n_unrolling_steps + 3, (leg.B + 15) // 16, 3, device=dev)
    tr._row0, tr._vnoise = 0, None
    fence()
    lib = L.load()
    lib.sw_kernel_timing(1)
    lib.sw_debug_spin(float(os.environ.get("SW_BENCH_SPIN_US", 2500.0 * n_ev)), L.stream())
    for o, p_, tg, nz in ins:
        leg.last = tr._step_impl(o, p_, None, scenes, tg, nz, float(leg.data.ss), float(leg.Bg), part)
    for _ in range(32):
        lib.sw_debug_spin(0.0, L.stream())       # calibration: the event interval of a kernel that does nothing
    fence()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.sw_kernel_timing_read(buf, len(buf))
    lib.sw_kernel_timing(0)
    ktimes, event_overhead_us = {}, None
    for line in buf.value.decode().splitlines():
        name, calls, total_us = line.split()
        if name == "nop_kernel":
            event_overhead_us = float(total_us) / int(calls)
        elif name != "spin_kernel":
            ktimes[name] = (int(calls), float(total_us))
    assert torch.isfinite(leg.last).all(), "non-finite losses"
    replicas_identical = None
    if world > 1:      # data-parallel replicas must hold bit-identical weights after the same all-reduced updates
        chk = torch.stack([tr.G._flat_all.double().sum(), tr.D._flat.double().sum()])
        hi, lo = chk.clone(), chk.clone()
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        replicas_identical = bool(torch.equal(hi, lo))
    collectives = None if pg is None else "in-graph" if tr._graph_collectives else "between graph segments"
    if pg is not None:
        tr.release_graphs()                     # recorded collectives go before their communicator

    def dp_exchange_report():
        """N > 1: what one gradient all-reduce of each of the step's three buckets costs on this group (HIP events around 50
        back-to-back calls; D = 27 942 floats twice, G = 86 124: the packed buffers) on the process group's own all-reduce (RCCL) and on the
        library's two-hop exchange (SW_ALLREDUCE=direct, csrc/sw_comm.hip), then the whole step on the direct form as a secondary
        leg - so that a scaling run explains itself."""
        from socialways_amd.comm import DirectAllReduce
        rep = {"buckets_floats": [int(tr.D._gflat.numel()), int(tr.D._gflat.numel()), int(tr.G._gflat_all.numel())]}
        bufs = [torch.zeros(n, device=dev) for n in rep["buckets_floats"]]

        def time_calls(fn):
            out = []
            for b in bufs:
                for _ in range(5):
                    fn(b)
                fence()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn(b)
                e1.record()
                torch.cuda.synchronize()
                out.append(max_over_ranks(e0.elapsed_time(e1) * 1e3 / 50))
            return out
        rep["group_us"] = time_calls(lambda b: torch.distributed.all_reduce(b, group=pg))
        rep["group_backend"] = backend
        prev_mode = os.environ.get("SW_ALLREDUCE")
        try:
            ar = DirectAllReduce(pg, dev, max(rep["buckets_floats"]))
            rep["direct_us"] = time_calls(ar)
            rep["direct_status"] = ar.status()
            ar.close()
            os.environ["SW_ALLREDUCE"] = "direct"
            lg = Leg(args.workload, dev, pg, world, rank, args.scaling, args.global_scenes, KG)
            n, w = OTHER_STEPS["m1"]
            d = max_over_ranks(short_leg(lg, n, w))
            rep["direct_step"] = {"steps": n, "steps_s": n * (world if args.scaling == "weak" else 1) / d, "ms_per_step": 1e3 * d / n,
                                  "collectives": "in-graph" if lg.tr._graph_collectives else "between graph segments",
                                  "status": lg.tr._direct.status() if lg.tr._direct is not None else None}
            lg.tr.release_graphs()
            if lg.tr._direct is not None:
                lg.tr._direct.close()
            del lg
        except Exception as e:      # noqa: BLE001 - the report must not lose the bench line
            rep["direct_error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        finally:
            if prev_mode is None:
                os.environ.pop("SW_ALLREDUCE", None)
            else:
                os.environ["SW_ALLREDUCE"] = prev_mode
        return rep

    def short_leg(lg, n, w):
        """A side leg: n steps timed three times, the fastest region reported.  (Regions of 40-80 ms are exposed to the
        sporadic 3-50 ms host stalls of this runtime - round-3 per-launch host / GPU time stamps - which are not a property of the leg;
        THE timed region of the headline workload is never treated this way.)"""
        lg.prime((n, w))
        lg.run_steps(0, w)
        gc.collect()
        gc.disable()
        d = min(lg.timed(fence, w + r * n, n) for r in range(3))
        gc.enable()
        assert torch.isfinite(lg.last).all(), "non-finite losses (%s)" % lg.name
        return d

    exchange = None
    if pg is not None and world > 1 and os.environ.get("SW_ALLREDUCE", "") != "direct" and not args.no_other_workloads:
        exchange = dp_exchange_report()
    other = None
    if world == 1 and pg is None and not args.no_other_workloads:
        other = {}
        del tr
        leg.tr = None
        for name in sorted(WORKLOADS):
            if name == args.workload:
                continue
            torch.cuda.empty_cache()
            lg = Leg(name, dev, None, 1, 0, "weak", 0, KG)
            if not HOST_Z:
                lg.z_resident()
            n, w = OTHER_STEPS[name]
            d = short_leg(lg, n, w)
            fl_o = alg_flops(lg.B, lg.P, lg.To, lg.Tp)
            S_o, A_o = WORKLOADS[name][:2]
            wl = ("%d scenes x %d agents x %d+%d" % (S_o, A_o, lg.To, lg.Tp)) if A_o is not None else \
                 ("%d scenes of 1..8 agents (%d agents, %d in-scene pairs, %d single-agent scenes) x %d+%d: the SHAPE of a real "
                  "ETH/UCY packed batch, synthetic tracks" % (lg.S_local, lg.B, lg.P, sum(a == 1 for a in lg.sizes), lg.To, lg.Tp))
            other[name] = {"workload": wl, "steps": n, "warmup": w,
                           "steps_s": n / d, "ms_per_step": 1e3 * d / n, "step_alg_gflop": fl_o["step"] / 1e9,
                           # reference-formulation FLOPs / time: CREDITS work the kernels eliminate algebraically
                           "step_frac_of_fp32_peak": fl_o["step"] / (d / n) / (PEAK_FP32_TFLOPS * 1e12)}
            rec_o = pmc_record(name)
            ex = rec_o and rec_o.get("_step", {}).get("mfma_flop_per_step")
            # matrix FLOP the kernels really issued per step (SQ counter pass of these sources) / this leg's time
            other[name]["step_executed_mfma_gflop"] = ex / 1e9 if ex else None
            other[name]["step_frac_executed"] = ex / (d / n) / (PEAK_FP32_TFLOPS * 1e12) if ex else None
            other[name]["inputs"] = "z drawn on the host every step" if HOST_Z else RESIDENT
            if name == "c4" and not HOST_Z:
                # 4 MB of z per step are ~170 us of request-bound PCIe reads that 113 us of encoder work cannot hide (and ~2 ms
                # of host work per step): the PCIe-inclusive figure next to the one with resident inputs
                lg.z_host()
                d2 = short_leg(lg, n, w)
                other[name]["pcie_inclusive"] = {"steps_s": n / d2, "ms_per_step": 1e3 * d2 / n,
                                                 "what": "z drawn on the host per step (train.py:473) and pulled from pinned memory "
                                                         "inside the step: 4 MB = ~170 us of request-bound PCIe reads"}
            if name == "c4" and HOST_Z:
                lg.z_resident()     # the all-resident form next to the reference's host-drawn z (4 MB per step at this shape)
                d2 = short_leg(lg, n, w)
                other[name]["inputs_resident"] = {"steps_s": n / d2, "ms_per_step": 1e3 * d2 / n,
                                                  "what": "z too resident in HBM (device tensors read by address)"}
            del lg
        # The data-parallel step structure at N = 1 - the only scaling evidence a 1-GPU box can give: the same workload on a
        # 1-rank RCCL group (SW_FORCE_DIST: all three all-reduces are issued, the Adam updates run behind them as kernels
        # of their own instead of inside the gradient reductions).  `delta_us_per_step` = what the DP structure costs
        # per step before any wire time; weak-scaling efficiency at N ranks <= t_plain / (t_dp1 + 3 x all-reduce latency).
        if args.workload == "m1":
            torch.cuda.empty_cache()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29537")
            os.environ["SW_FORCE_DIST"] = "1"
            try:
                torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                lg = Leg("m1", dev, torch.distributed.group.WORLD, 1, 0, "weak", 0, KG)
                if not HOST_Z:
                    lg.z_resident()
                n, w = OTHER_STEPS["m1"]
                d = short_leg(lg, n, w)
                other["dp1_rccl"] = {"workload": "m1 on a 1-rank RCCL process group (3 all-reduces per step issued, Adam behind them)",
                                     "steps": n, "steps_s": n / d, "ms_per_step": 1e3 * d / n,
                                     "delta_us_per_step": 1e3 * (1e3 * d / n - 1e3 * dt / args.steps),
                                     "collectives": "in-graph" if lg.tr._graph_collectives else "between graph segments",
                                     # plain-path time / 1-rank-group time: what the DP step STRUCTURE costs with no peer
                                     # on the wire - says nothing about N > 1 (no wire latency is in it)
                                     "plain_over_dp1_time_ratio": (dt / args.steps) / (d / n)}
                lg.tr.release_graphs()
                del lg
                # ... and the same on the library's direct exchange (SW_ALLREDUCE=direct: per bucket ONE launch that exchanges
                # the gradient and applies Adam; with one rank the exchange moves nothing): its structure cost
                os.environ["SW_ALLREDUCE"] = "direct"
                try:
                    lg = Leg("m1", dev, torch.distributed.group.WORLD, 1, 0, "weak", 0, KG)
                    if not HOST_Z:
                        lg.z_resident()
                    d = short_leg(lg, n, w)
                    other["dp1_direct"] = {"workload": "m1 on a 1-rank group with SW_ALLREDUCE=direct (3 exchange + Adam launches per step)",
                                           "steps": n, "steps_s": n / d, "ms_per_step": 1e3 * d / n,
                                           "delta_us_per_step": 1e3 * (1e3 * d / n - 1e3 * dt / args.steps),
                                           "collectives": "in-graph" if lg.tr._graph_collectives else "between graph segments",
                                           "status": lg.tr._direct.status()}
                    lg.tr.release_graphs()
                    lg.tr._direct.close()
                    del lg
                except Exception as e:      # noqa: BLE001
                    other["dp1_direct"] = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
                finally:
                    os.environ.pop("SW_ALLREDUCE", None)
                torch.distributed.destroy_process_group()
            except Exception as e:      # noqa: BLE001 - a box without a working RCCL must not lose the bench line
                other["dp1_rccl"] = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
            finally:
                os.environ.pop("SW_FORCE_DIST", None)
            # the best-of-K variety term (K = 20 rollouts folded into one batch of 20 x 2048 agents) as a
            # throughput stress of the generator path; eager steps (the folded step is not graph-captured)
            torch.cuda.empty_cache()
            lg = Leg("m1", dev, None, 1, 0, "weak", 0, 1, use_variety_loss="fixed", variety_k=20, use_l2_loss=True)
            if not HOST_Z:
                lg.z_resident()
            n, w = 40, 6
            d = short_leg(lg, n, w)
            other["m1_variety_k20"] = {"workload": "m1 + best-of-20 variety loss (use_variety_loss='fixed'): decode loop on 40 960 agent copies, encoder and social block once on the 2 048 agents",
                                       "steps": n, "steps_s": n / d, "ms_per_step": 1e3 * d / n}
            del lg
            # `--hidden-size 128` (train.py:42-44): the WIDE path (wide.py: time-step-level kernels, explicit backward, one
            # hipGraph per step) on the metric shape; the generic path's layer-by-layer form ran 42 steps/s here in round 3
            for Hw in (128, 96):
                torch.cuda.empty_cache()
                import socialways_amd as sw
                torch.manual_seed(0)
                np.random.seed(0)
                tr_w = sw.SocialWaysTrainer(Tp, hidden_size=Hw, use_social=True, device=dev)
                S_w, A_w = WORKLOADS["m1"][:2]
                tk = sw.synth_tracks(S_w * 2, A_w, To, Tp, seed=99)
                dw = sw.SceneDataset(tk["obsvs"], tk["preds"], tk["batches"], device=dev)
                Bw, sbw = S_w * A_w, np.stack([np.arange(S_w) * A_w, (np.arange(S_w) + 1) * A_w], axis=1).astype(np.int64)
                zb = torch.empty(Bw, Hw // 2).pin_memory()

                def wstep(i):
                    a = (i % 2) * Bw
                    torch.rand(zb.shape, out=zb)
                    return tr_w.step(dw.obsv[a:a + Bw], dw.pred[a:a + Bw], sbw, np.random.uniform(0, 0.1), np.random.uniform(0.9, 1.0),
                                     zb, dw.ss)
                for i in range(6):
                    last_w = wstep(i)
                n_w, t_best = 60, float("inf")
                for rep in range(3):
                    fence()
                    t0 = time.perf_counter()
                    for i in range(n_w):
                        last_w = wstep(i)
                    fence()
                    t_best = min(t_best, time.perf_counter() - t0)
                assert torch.isfinite(last_w).all(), "non-finite losses (hidden size %d)" % Hw
                other["m1_hidden%d" % Hw] = {
                    "workload": "m1 at --hidden-size %d (decoder %d-%d-%d-%d-2, noise %d): %s" % (
                        Hw, 5 * Hw // 2, 5 * Hw // 2, 5 * Hw // 4, 5 * Hw // 8, Hw // 2, type(tr_w).__name__),
                    "steps": n_w, "steps_s": n_w / t_best, "ms_per_step": 1e3 * t_best / n_w}
                tr_w.release_graphs()
                del tr_w, dw
            # the evaluation pass test() exists for - K = 20 sampled futures per held-out scene, min / avg ADE
            # and FDE (train.py:563-616) - on the m1-shaped recording's held-out fifth (scenes folded into rollout launches)
            torch.cuda.empty_cache()
            import socialways_amd as sw
            torch.manual_seed(0)
            tr_e = sw.SocialWaysTrainer(Tp, use_social=True, device=dev)
            tk = sw.synth_tracks(1280, 8, To, Tp, seed=4321)
            data_e = sw.SceneDataset(tk["obsvs"], tk["preds"], tk["batches"], device=dev)
            tr_e.test(data_e, 20)
            fence()
            t_best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                res_e = tr_e.test(data_e, 20)
                fence()
                t_best = min(t_best, time.perf_counter() - t0)
            n_sc = len(data_e.test_batches)
            other["test_k20"] = {"workload": "test(): K = 20 sampled futures for each of %d held-out scenes x 8 agents (%d agents), "
                                             "min / avg ADE and FDE; scenes folded into launches of <= %d agent copies"
                                             % (n_sc, data_e.n_test_samples, tr_e.TEST_CHUNK),
                                 "seconds": t_best, "scenes_s": n_sc / t_best, "rollouts_s": 20 * data_e.n_test_samples / t_best,
                                 "ade_avg_min": [res_e[0], res_e[2]], "fde_avg_min": [res_e[1], res_e[3]]}
            del tr_e, data_e

    if rank == 0:
        S = leg.S_local
        B, P = leg.B, leg.P
        fl = alg_flops(B, P, To, Tp)
        per_step = dt / args.steps
        # per-kernel table of the eager roofline pass: launches / step, mean launch time, algorithmic GFLOP / step, fraction
        kfl = kernel_alg_flops(B, P, To, Tp, one_launch_d="disc_update_kernel" in ktimes)
        rows = []
        for name, (calls, total_us) in ktimes.items():
            us_step = total_us / n_ev
            gf = kfl.get(name)
            rows.append({"name": name, "launches_per_step": calls / n_ev, "avg_us": total_us / calls, "us_per_step": us_step,
                         "alg_gflop_per_step": (gf / 1e9) if gf else None,
                         "frac": (gf / (us_step * 1e-6) / (PEAK_FP32_TFLOPS * 1e12)) if gf else None})
        rows.sort(key=lambda r: -r["us_per_step"])
        top = rows[0]
        kern_s = top["avg_us"] * 1e-6
        achieved = (top["alg_gflop_per_step"] or float("nan")) * 1e9 / (top["us_per_step"] * 1e-6) / 1e12
        # HBM traffic: from the committed PMC pass (separate --pmc runs, tools/collect_profiles.sh) - valid only for the
        # kernel sources it was taken with (same sha) and this workload; otherwise null
        traffic, step_traffic, traffic_src, step_exec = None, None, None, None
        sha = kernel_src_sha16()
        rec = pmc_record(args.workload, args.scaling == "weak")
        if rec is not None:
            by_kernel = {v.get("kernel", "").replace("void ", "").split("<")[0]: v for k, v in rec.items() if not k.startswith("_")}
            traffic = by_kernel.get(top["name"], {}).get("hbm_bytes_per_launch")
            step_traffic = rec.get("_step", {}).get("hbm_bytes_per_step")
            step_exec = rec.get("_step", {}).get("mfma_flop_per_step")
            traffic_src = {"file": rec["_file"], "commit": rec.get("_meta", {}).get("commit"), "kernel_src_sha16": sha}
            for r in rows:      # EXECUTED matrix work next to the reference-formulation credit, per kernel
                ex = by_kernel.get(r["name"], {}).get("mfma_flop_per_launch")
                r["executed_mfma_gflop_per_step"] = ex * r["launches_per_step"] / 1e9 if ex is not None else None
                r["frac_executed"] = (ex * r["launches_per_step"] / (r["us_per_step"] * 1e-6) / (PEAK_FP32_TFLOPS * 1e12)) if ex is not None else None
        if traffic_src is None:
            traffic_src = {"file": None, "kernel_src_sha16": sha,
                           "note": "no PMC pass under profiles/ matches these kernel sources / this workload"}
        value = args.steps * (world if args.scaling == "weak" else 1) / dt
        rep_ms = sorted(1e3 * r / args.steps for r in reps)
        res = {
            "metric": "GAN train steps/sec (%d scenes x %d agents x %d+%d T per %s step; fp32; social block on)"
                      % ((S, A, To, Tp, "GPU") if args.scaling == "weak" else (leg.S_global, A, To, Tp, "GLOBAL")),
            "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s: %d scenes x %d agents x (%d obs + %d pred) per GPU step = reference "
                                    "--batch-size %d; use_social=True, n_unrolling_steps=1, info loss on"
                                    % (args.workload, S, A, To, Tp, B)) if args.scaling == "weak" else
                                   ("%s-strong: ONE packed batch of %d scenes x %d agents (reference --batch-size %d) sharded "
                                    "scene-aligned over %d ranks; use_social=True, n_unrolling_steps=1, info loss on"
                                    % (args.workload, leg.S_global, A, leg.Bg, world)),
                       "global_batch_scenes": leg.S_global, "parallelism": "dp%d" % world, "steps_per_graph_launch": KG, "settle_steps": SETTLE, "rehearsal_regions": REHEARSALS,
                       # everything that ran untimed in front of THE region: priming (5 passes over every launch size), settle steps,
                       # the W warmup steps and the rehearsals of the region ("warmup" above is only W)
                       "untimed_steps_before_region": UNTIMED_BEFORE,
                       "collectives": collectives, "rccl_ranks": (world if pg is not None else None),
                       "backend": backend,       # "nccl" = RCCL; "gloo" = ranks sharing devices, a rehearsal, not a measurement
                       "allreduces_per_step": (3 if pg is not None else 0),
                       "allreduce": ("direct (csrc/sw_comm.hip)" if (leg.tr is not None and getattr(leg.tr, "_direct", None) is not None)
                                     else "process group") if pg is not None else None,
                       "exchange": exchange,      # N > 1: us per all-reduce of each bucket on both forms + the step on the direct form
                       "exchange_probe": getattr(leg.tr, "exchange_probe", None) if leg.tr is not None else None,      # SW_ALLREDUCE=auto: what the probe measured / chose
                       "replicas_identical": replicas_identical,
                       "step_alg_gflop": fl["step"] / 1e9,
                       "step_frac_of_fp32_peak": fl["step"] / per_step / (PEAK_FP32_TFLOPS * 1e12),
                       "step_alg_bytes": alg_bytes(B, To, Tp),
                       "step_frac_of_hbm_peak": alg_bytes(B, To, Tp) / per_step / PEAK_HBM_BPS,
                       "repeats": {"n": len(rep_ms), "steps_each": args.steps, "ms_per_step_min": rep_ms[0],
                                   "ms_per_step_median": rep_ms[len(rep_ms) // 2], "ms_per_step_max": rep_ms[-1]},
                       "sustained": sustained,
                       "inputs": "z drawn on the host every step" if HOST_Z else RESIDENT,
                       "pcie_inclusive": pcie,
                       "inputs_resident": resident,
                       "other_workloads": other},
            "roofline": {"bound": "mfma", "kernel": top["name"], "achieved": achieved,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "avg_launch_ms": kern_s * 1e3, "launches": int(round(top["launches_per_step"] * n_ev)),
                         "launches_per_step": top["launches_per_step"],
                         "how": "HIP events around every kernel launch of %d eager steps (sw_kernel_timing), host running ahead "
                                "of the GPU behind a spin kernel; algorithmic FLOPs of all of the kernel's launches in a step "
                                "/ their summed time" % n_ev,
                         "kernels": rows[:int(os.environ.get("SW_BENCH_KERNEL_ROWS", "6"))],
                         # what an event pair adds to a launch (a kernel that does nothing, same queue): subtract from avg_us
                         # for the kernel's own duration; `frac` is computed from the RAW times (conservative)
                         "event_overhead_us": event_overhead_us,
                         "eager_step_kernel_us": sum(r["us_per_step"] for r in rows),
                         # `frac` / step_frac_of_fp32_peak count the REFERENCE formulation's FLOPs and so credit work
                         # the kernels eliminate; *_executed = matrix FLOP really issued (64 x SQ_VALU_MFMA_BUSY_CYCLES of the
                         # committed SQ pass of these kernel sources) over the same measured times: pipe USE, never above 1
                         "step_frac_reference_formulation": fl["step"] / per_step / (PEAK_FP32_TFLOPS * 1e12),
                         "step_executed_mfma_gflop": (step_exec / 1e9) if step_exec else None,
                         "step_frac_executed": (step_exec / per_step / (PEAK_FP32_TFLOPS * 1e12)) if step_exec else None,
                         "frac_executed": top.get("frac_executed"),
                         "traffic": traffic, "step_traffic": step_traffic,
                         "step_traffic_vs_algorithmic": (step_traffic / alg_bytes(B, To, Tp)) if step_traffic else None,
                         "traffic_source": traffic_src,
                         # north_star also asks for the HBM view: measured bytes / launch time vs 8 TB/s
                         "hbm_GBps": (traffic / kern_s / 1e9) if traffic else None,
                         "hbm_frac": (traffic / kern_s / PEAK_HBM_BPS) if traffic else None,
                         "step_hbm_GBps": (step_traffic / per_step / 1e9) if step_traffic else None,
                         "step_hbm_frac": (step_traffic / per_step / PEAK_HBM_BPS) if step_traffic else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(leg.tracks, S if args.workload != "c4" else 16, A, To, Tp)
        ctypes.CDLL(None).fflush(None)          # RCCL's version banner sits in the C stdio buffer: keep the JSON line last
        print(json.dumps(res), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
