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
    # the natural per-GPU shard of an 8 192-scene global batch (train.py:446-456 packs --batch-size agents): 8 192 agents =
    # 512 sixteen-agent tiles, two per CU - a shape on which the step's three gradient exchanges amortise
    "m4": (1024, 8, 8, 12),
}
OTHER_STEPS = {"m1": (100, 12), "c2": (100, 12), "c4": (24, 8), "c3_ragged": (100, 12), "m4": (60, 10)}     # (steps, warm-up) of a short leg
RESIDENT = "resident in HBM when the timed region starts (tracks; z: a ring of 16 device tensors drawn beforehand)"
SIDE_GROUPS = ("shapes", "dp1", "extra")     # side legs of the N = 1 line, one child process per group (run_side_children)
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
        "dec_rollout_fwd2_kernel": 2.0 * B * (Tp * dec + (Tp - 1) * lstm),       # two column blocks per workgroup (> 256 tiles: no riders)
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



def _err(e):
    return {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}


def _write_json(path, obj):
    """Atomic (a reader never sees half a file): side legs rewrite their record after every leg, so whatever finished before a
    crash or a time-out of the child survives it."""
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)


def _read_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


_CHILD = {"proc": None}     # the side-leg child that is running (the SIGTERM handler of the parent ends it)
LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_NAME",
                "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SW_FORCE_DIST", "SW_ALLREDUCE")


def run_child(cmd, timeout_s, env_extra=None):
    """A side leg as a process (group) of its own: whatever happens in it - an exception, a GPU memory fault that aborts the
    process, a wait that never ends - costs that leg, never the ranks that hold the headline measurement.  Returns
    (return code or None after a time-out, last lines of its stderr)."""
    import signal
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV and not k.startswith("TORCHELASTIC_")}
    env["SW_BENCH_CHILD"] = "1"
    env.update(env_extra or {})
    with tempfile.TemporaryFile(mode="w+") as errf:
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=errf, start_new_session=True)
        _CHILD["proc"] = proc
        try:
            rc = proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            rc = None
        finally:
            _CHILD["proc"] = None
        if rc is None or rc != 0:            # its own process group (start_new_session): the launcher's workers go with it
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.wait()
        errf.seek(0)
        tail = errf.read()[-1500:]
    return rc, tail


def child_timeout(default_s):
    return float(os.environ.get("SW_BENCH_CHILD_TIMEOUT_S", default_s))


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


def short_leg(lg, n, w, fence):
    """A side leg: n steps timed three times, the fastest region reported.  (Regions of 40-80 ms are exposed to the
    sporadic 3-50 ms host stalls of this runtime - round-3 per-launch host / GPU time stamps - which are not a property of the leg;
    THE timed region of the headline workload is never treated this way.)"""
    import gc
    lg.prime((n, w))
    lg.run_steps(0, w)
    gc.collect()
    gc.disable()
    d = min(lg.timed(fence, w + r * n, n) for r in range(3))
    gc.enable()
    assert torch.isfinite(lg.last).all(), "non-finite losses (%s)" % lg.name
    return d


def init_group(backend, dev, world=None, rank=None):
    kw = {} if world is None else {"world_size": world, "rank": rank}
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", device_id=dev, **kw)
    else:
        torch.distributed.init_process_group(backend, **kw)
    return torch.distributed.group.WORLD


def side_legs_main(args, dev, KG, HOST_Z):
    """`bench.py --side-legs GROUP --side-out FILE` - a CHILD of the N = 1 bench (run_side_children): the side legs of one
    group, each under its own guard ({"error": ...} in its place), the record rewritten after every leg."""
    import socialways_amd as sw
    out = {}
    ref_ms = args.ref_ms                      # the parent's headline ms per step (dp1 legs report their distance to it)

    def fence():
        torch.cuda.synchronize()

    def put(key, fn):
        try:
            out[key] = fn()
        except Exception as e:      # noqa: BLE001 - a leg that fails is reported in its place; the others still run
            out[key] = _err(e)
            torch.cuda.synchronize()
        _write_json(args.side_out, out)
        torch.cuda.empty_cache()

    def shape_leg(name):
        lg = Leg(name, dev, None, 1, 0, "weak", 0, KG)
        if not HOST_Z:
            lg.z_resident()
        n, w = OTHER_STEPS[name]
        d = short_leg(lg, n, w, fence)
        fl_o = alg_flops(lg.B, lg.P, lg.To, lg.Tp)
        S_o, A_o = WORKLOADS[name][:2]
        wl = ("%d scenes x %d agents x %d+%d" % (S_o, A_o, lg.To, lg.Tp)) if A_o is not None else \
             ("%d scenes of 1..8 agents (%d agents, %d in-scene pairs, %d single-agent scenes) x %d+%d: the SHAPE of a real "
              "ETH/UCY packed batch, synthetic tracks" % (lg.S_local, lg.B, lg.P, sum(a == 1 for a in lg.sizes), lg.To, lg.Tp))
        rec = {"workload": wl, "steps": n, "warmup": w, "agents": lg.B,
               "steps_s": n / d, "ms_per_step": 1e3 * d / n, "agent_steps_s": lg.B * n / d, "step_alg_gflop": fl_o["step"] / 1e9,
               # reference-formulation FLOPs / time: CREDITS work the kernels eliminate algebraically
               "step_frac_of_fp32_peak": fl_o["step"] / (d / n) / (PEAK_FP32_TFLOPS * 1e12)}
        rec_o = pmc_record(name)
        ex = rec_o and rec_o.get("_step", {}).get("mfma_flop_per_step")
        # matrix FLOP the kernels really issued per step (SQ counter pass of these sources) / this leg's time
        rec["step_executed_mfma_gflop"] = ex / 1e9 if ex else None
        rec["step_frac_executed"] = ex / (d / n) / (PEAK_FP32_TFLOPS * 1e12) if ex else None
        rec["inputs"] = "z drawn on the host every step" if HOST_Z else RESIDENT
        if name in ("c4", "m4"):
            # z in the other form next to it (c4: 4 MB of z per step, m4: 1 MB)
            if HOST_Z:
                lg.z_resident()
            else:
                lg.z_host()
            d2 = short_leg(lg, n, w, fence)
            rec["inputs_resident" if HOST_Z else "pcie_inclusive"] = {
                "steps_s": n / d2, "ms_per_step": 1e3 * d2 / n,
                "what": "z too resident in HBM (device tensors read by address)" if HOST_Z else
                        "z drawn on the host per step (train.py:473) and pulled from pinned memory inside the step"}
        lg.tr.close()
        return rec

    def dp1_leg(direct):
        """The data-parallel step STRUCTURE at N = 1 - the only scaling evidence a 1-GPU box can give: the same workload on a
        1-rank RCCL group (SW_FORCE_DIST: all three all-reduces are issued, the Adam updates run behind them as kernels of
        their own instead of inside the gradient reductions); `direct`: the same on the library's exchange (per bucket ONE
        launch that exchanges the gradient and applies Adam; with one rank the exchange moves nothing).  `delta_us_per_step`
        = what the structure costs per step before any wire time - it says nothing about N > 1."""
        os.environ["SW_FORCE_DIST"] = "1"
        if direct:
            os.environ["SW_ALLREDUCE"] = "direct"
        else:
            os.environ.pop("SW_ALLREDUCE", None)
        try:
            lg = Leg("m1", dev, torch.distributed.group.WORLD, 1, 0, "weak", 0, KG)
            if not HOST_Z:
                lg.z_resident()
            n, w = OTHER_STEPS["m1"]
            d = short_leg(lg, n, w, fence)
            rec = {"workload": ("m1 on a 1-rank group with SW_ALLREDUCE=direct (3 exchange + Adam launches per step)" if direct else
                                "m1 on a 1-rank RCCL process group (3 all-reduces per step issued, Adam behind them)"),
                   "steps": n, "steps_s": n / d, "ms_per_step": 1e3 * d / n,
                   "delta_us_per_step": (1e3 * (1e3 * d / n - ref_ms)) if ref_ms else None,
                   "collectives": "in-graph" if lg.tr._graph_collectives else "between graph segments"}
            if direct:
                rec["status"] = lg.tr._direct.status()
            elif ref_ms:
                rec["plain_over_dp1_time_ratio"] = ref_ms / (1e3 * d / n)
            lg.tr.close()
            return rec
        finally:
            os.environ.pop("SW_FORCE_DIST", None)
            os.environ.pop("SW_ALLREDUCE", None)

    def variety_leg():
        # the best-of-K variety term (K = 20 rollouts folded into one batch of 20 x 2048 agents) as a
        # throughput stress of the generator path; eager steps (the folded step is not graph-captured)
        lg = Leg("m1", dev, None, 1, 0, "weak", 0, 1, use_variety_loss="fixed", variety_k=20, use_l2_loss=True)
        if not HOST_Z:
            lg.z_resident()
        n, w = 40, 6
        d = short_leg(lg, n, w, fence)
        return {"workload": "m1 + best-of-20 variety loss (use_variety_loss='fixed'): decode loop on 40 960 agent copies, encoder and social block once on the 2 048 agents",
                "steps": n, "steps_s": n / d, "ms_per_step": 1e3 * d / n}

    def wide_leg(Hw):
        # `--hidden-size 128` (train.py:42-44): the WIDE path (wide.py: time-step-level kernels, explicit backward, one
        # hipGraph per step) on the metric shape; the generic path's layer-by-layer form ran 42 steps/s here in round 3
        S_w, A_w, To, Tp = WORKLOADS["m1"]
        torch.manual_seed(0)
        np.random.seed(0)
        tr_w = sw.SocialWaysTrainer(Tp, hidden_size=Hw, use_social=True, device=dev)
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
        rec = {"workload": "m1 at --hidden-size %d (decoder %d-%d-%d-%d-2, noise %d): %s" % (
                   Hw, 5 * Hw // 2, 5 * Hw // 2, 5 * Hw // 4, 5 * Hw // 8, Hw // 2, type(tr_w).__name__),
               "steps": n_w, "steps_s": n_w / t_best, "ms_per_step": 1e3 * t_best / n_w}
        tr_w.release_graphs()
        return rec

    def test_leg():
        # the evaluation pass test() exists for - K = 20 sampled futures per held-out scene, min / avg ADE
        # and FDE (train.py:563-616) - on the m1-shaped recording's held-out fifth (scenes folded into rollout launches)
        To, Tp = WORKLOADS["m1"][2:]
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
        return {"workload": "test(): K = 20 sampled futures for each of %d held-out scenes x 8 agents (%d agents), "
                            "min / avg ADE and FDE; scenes folded into launches of <= %d agent copies"
                            % (n_sc, data_e.n_test_samples, tr_e.TEST_CHUNK),
                "seconds": t_best, "scenes_s": n_sc / t_best, "rollouts_s": 20 * data_e.n_test_samples / t_best,
                "ade_avg_min": [res_e[0], res_e[2]], "fde_avg_min": [res_e[1], res_e[3]]}

    if args.side_legs == "shapes":
        for name in sorted(WORKLOADS):
            if name != args.workload:
                put(name, lambda name=name: shape_leg(name))
    elif args.side_legs == "dp1":
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(sk.getsockname()[1])
        sk.close()
        try:
            init_group("nccl", dev, world=1, rank=0)
        except Exception as e:      # noqa: BLE001 - a box without a working RCCL
            out["dp1_rccl"] = _err(e)
            _write_json(args.side_out, out)
            return 0
        put("dp1_rccl", lambda: dp1_leg(False))
        put("dp1_direct", lambda: dp1_leg(True))
        torch.distributed.destroy_process_group()
    elif args.side_legs == "extra":
        put("m1_variety_k20", variety_leg)
        for Hw in (128, 96):
            put("m1_hidden%d" % Hw, lambda Hw=Hw: wide_leg(Hw))
        put("test_k20", test_leg)
    _write_json(args.side_out, out)
    return 0


def run_side_children(args, ref_ms, tmpdir):
    """The N = 1 line's side legs (config.other_workloads), one child process per group: the line cannot be lost to them."""
    other = {}
    base = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--ref-ms", "%.6f" % ref_ms]
    for group, t_s in zip(SIDE_GROUPS, (360.0, 240.0, 360.0)):
        path = os.path.join(tmpdir, "side_%s.json" % group)
        t0 = time.perf_counter()
        rc, tail = run_child(base + ["--side-legs", group, "--side-out", path], child_timeout(t_s))
        got = _read_json(path) or {}
        other.update(got)
        if rc != 0:      # whatever the child finished before it died is kept; the rest of its group is reported missing
            other["_%s_child" % group] = {"error": ("timed out after %.0f s" % (time.perf_counter() - t0)) if rc is None else "exit code %d" % rc,
                                          "legs_finished": sorted(got), "stderr_tail": tail[-600:]}
    return other


def exchange_child_main(args, world, rank, dev, pg, backend, KG):
    """`bench.py --gpus N --exchange-only --side-out FILE` under the launcher - a CHILD JOB of the N > 1 bench
    (run_exchange_child): what one gradient all-reduce of each of the step's three buckets costs on this node (HIP events
    around 50 back-to-back calls; D twice, G once: the packed buffers) on the process group's own all-reduce (RCCL) and on the
    library's two-hop exchange (SW_ALLREDUCE=direct, csrc/sw_comm.hip), then the whole step on the direct form.  Rank 0
    rewrites the record after every stage: what was measured before a fault survives it."""
    import socialways_amd as sw
    from socialways_amd.comm import DirectAllReduce
    rep = {}

    def fence():
        torch.distributed.barrier()
        torch.cuda.synchronize()

    def save():
        if rank == 0:
            _write_json(args.side_out, rep)

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())
    Tp = WORKLOADS[args.workload][3]
    probe_tr = sw.SocialWaysTrainer(Tp, use_social=True, device=dev)       # (no group: only for the bucket sizes)
    rep["buckets_floats"] = [int(probe_tr.D._gflat.numel()), int(probe_tr.D._gflat.numel()), int(probe_tr.G._gflat_all.numel())]
    del probe_tr
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
    rep["group_backend"] = backend
    rep["group_us"] = time_calls(lambda b: torch.distributed.all_reduce(b, group=pg))
    save()
    try:
        ar = DirectAllReduce(pg, dev, max(rep["buckets_floats"]))          # failure-symmetric: raises on every rank or none
        rep["direct_agrees_with_group"] = ar.check(rep["buckets_floats"][1:])
        rep["direct_us"] = time_calls(ar)
        rep["direct_status"] = ar.status_all()
        ar.close()
        save()
        os.environ["SW_ALLREDUCE"] = "direct"
        lg = Leg(args.workload, dev, pg, world, rank, args.scaling, args.global_scenes, KG)
        n, w = OTHER_STEPS[args.workload]
        d = max_over_ranks(short_leg(lg, n, w, fence))
        rep["direct_step"] = {"steps": n, "steps_s": n * (world if args.scaling == "weak" else 1) / d, "ms_per_step": 1e3 * d / n,
                              "collectives": "in-graph" if lg.tr._graph_collectives else "between graph segments",
                              "status": lg.tr._direct.status_all() if lg.tr._direct is not None else None}
        save()
        # ... and the HEADLINE's own procedure on the direct form (priming of every launch size, settle steps, W warmup steps,
        # two rehearsals, then EXACTLY K steps between fences, max over ranks): the number to hold against the line's `value`
        lg.prime((args.steps, args.warmup))
        lg.run_steps(0, int(os.environ.get("SW_BENCH_SETTLE_STEPS", "1000")) // KG * KG)
        lg.run_steps(0, args.warmup)
        for r in range(2):
            lg.timed(fence, args.warmup, args.steps)
        dh = max_over_ranks(lg.timed(fence, args.warmup, args.steps))
        rep["direct_headline"] = {"value": args.steps * (world if args.scaling == "weak" else 1) / dh, "unit": "steps/s",
                                  "steps": args.steps, "ms_per_step": 1e3 * dh / args.steps,
                                  "status": lg.tr._direct.status_all() if lg.tr._direct is not None else None,
                                  "what": "the headline's procedure with SW_ALLREDUCE=direct, in this child job"}
        lg.tr.close()
    except Exception as e:      # noqa: BLE001
        rep["direct_error"] = _err(e)["error"]
    save()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    return 0


def run_child_job(args, world, tmpdir, mode, tag, timeout_s):
    """Rank 0 of the N > 1 bench: a second job of `world` ranks on the same GPUs (`bench.py --gpus N --<mode>` under its own
    torch.distributed.run), the parent's ranks idle on the host meanwhile.  Whatever happens in it - a GPU fault, a failed peer
    mapping, a collective that never completes - ends the CHILD; the record keeps `error` and what was written before it."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    path = os.path.join(tmpdir, tag + ".json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(world), "--" + mode, "--side-out", path, "--workload", args.workload,
           "--scaling", args.scaling, "--global-scenes", str(args.global_scenes), "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    t0 = time.perf_counter()
    rc, tail = run_child(cmd, child_timeout(timeout_s), {"OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", "8")})
    rep = _read_json(path) or {}
    if rc != 0:
        rep["error"] = ("the child job timed out after %.0f s" % (time.perf_counter() - t0)) if rc is None else \
                       "the child job exited with code %d" % rc
        rep["stderr_tail"] = tail[-600:]
    rep["how"] = "a child job of %d ranks on the same GPUs (bench.py --%s), the parent's ranks idle; %.0f s" % (
        world, mode, time.perf_counter() - t0)
    return rep


def probe_child_main(args, world, rank, dev, pg, backend, KG):
    """`bench.py --gpus N --probe-only --side-out FILE` - a CHILD JOB run BEFORE the headline of an N > 1 bench: can this
    node replay the data-parallel step with the RCCL all-reduces recorded INSIDE the hipGraph?  The trainer probes that by
    itself (a captured all-reduce replayed and checked), but a captured collective that never completes would take the
    bench's ranks with it; here it takes a child.  The child then runs the real thing - eager steps, the capture, replays
    of 4-step launches - and checks the replicas.  No verdict (crash, time-out) = graph segments with eager collectives."""
    rep = {"backend": backend}
    os.environ.pop("SW_GRAPH_COLLECTIVES", None)       # the trainer's own probe decides
    lg = Leg(args.workload, dev, pg, world, rank, args.scaling, args.global_scenes, 4)
    lg.run_steps(0, 24)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(lg.last).all())
    chk = torch.stack([lg.tr.G._flat_all.double().sum(), lg.tr.D._flat.double().sum()])
    hi, lo = chk.clone(), chk.clone()
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    rep["replicas_identical"] = bool(torch.equal(hi, lo))
    rep["finite"] = ok
    rep["graph_collectives"] = bool(lg.tr._graph_collectives) and ok and rep["replicas_identical"]
    lg.tr.close()
    torch.distributed.barrier()
    if rank == 0:
        _write_json(args.side_out, rep)
    torch.distributed.destroy_process_group()
    return 0


def on_rank0(world, rank, dev, backend, tmpdir, tag, fn, wait_s):
    """fn() on rank 0 (it returns a JSON-able record); the other ranks wait for it ON THE HOST - a file, not a collective: a
    barrier's kernels would spin on the GPUs a child job measures on - and every rank returns the record."""
    path, done = os.path.join(tmpdir, tag + ".result.json"), os.path.join(tmpdir, tag + ".done")
    if rank == 0:
        try:
            rec = fn()
        except Exception as e:      # noqa: BLE001
            rec = _err(e)
        _write_json(path, rec)
        open(done, "w").close()
        return rec
    deadline = time.monotonic() + wait_s
    while not os.path.exists(done) and time.monotonic() < deadline:
        time.sleep(0.2)
    return _read_json(path) or {"error": "rank 0 did not report within %.0f s" % wait_s}


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
    # children of this file (run_side_children / run_exchange_child): the side legs run where they cannot take the line down
    ap.add_argument("--side-legs", choices=SIDE_GROUPS, help=argparse.SUPPRESS)
    ap.add_argument("--side-out", help=argparse.SUPPRESS)
    ap.add_argument("--ref-ms", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--exchange-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-only", action="store_true", help=argparse.SUPPRESS)
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
        pg = init_group(backend, dev)

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

    HOST_Z = os.environ.get("SW_BENCH_HOST_Z", "1") != "0" or args.scaling == "strong"
    if args.side_legs:
        return side_legs_main(args, dev, KG, HOST_Z)
    if args.exchange_only:
        return exchange_child_main(args, world, rank, dev, pg, backend, KG)
    if args.probe_only:
        return probe_child_main(args, world, rank, dev, pg, backend, KG)
    # N > 1 on RCCL: whether the all-reduces may be recorded INSIDE the step graph is decided by a child job that tries it
    # (probe_child_main) - a captured collective that never completes must not take the ranks that will hold the headline.
    # SW_GRAPH_COLLECTIVES=0 / 1 in the environment skips the child.
    tmpdir, probe = None, None
    if pg is not None and world > 1:
        import tempfile
        obj = [tempfile.mkdtemp(prefix="sw_bench_")] if rank == 0 else [None]
        torch.distributed.broadcast_object_list(obj, src=0, device=dev if backend == "nccl" else None)
        tmpdir = obj[0]
        # (SW_BENCH_FORCE_PROBE=1: also on gloo ranks sharing a device - a rehearsal of the plumbing, the verdict is "segments")
        if (backend == "nccl" or os.environ.get("SW_BENCH_FORCE_PROBE") == "1") and "SW_GRAPH_COLLECTIVES" not in os.environ:
            probe = on_rank0(world, rank, dev, backend, tmpdir, "probe",
                             lambda: run_child_job(args, world, tmpdir, "probe-only", "probe", 300.0), child_timeout(300.0) + 120.0)
            os.environ["SW_GRAPH_COLLECTIVES"] = "1" if probe.get("graph_collectives") is True and "error" not in probe else "0"
    import gc
    leg = Leg(args.workload, dev, pg, world, rank, args.scaling, args.global_scenes, KG)
    # The headline step is the reference's: z = torch.rand(bs, noise_len) drawn on the HOST every step and copied
    # to the device inside the step (train.py:473); the tracks are resident in HBM.  SW_BENCH_HOST_Z=0 makes the all-resident
    # form (z a ring of device tensors drawn beforehand) the headline instead; by default it is the secondary figure
    # config.inputs_resident.  --scaling strong always draws z for the global batch on the host and slices it.
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
    ktimes, event_overhead_us, roofline_error = {}, None, None
    try:
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
        for line in buf.value.decode().splitlines():
            name, calls, total_us = line.split()
            if name == "nop_kernel":
                event_overhead_us = float(total_us) / int(calls)
            elif name != "spin_kernel":
                ktimes[name] = (int(calls), float(total_us))
    except Exception as e:      # noqa: BLE001 - the per-kernel view is lost, the timed region is not
        if world > 1:
            raise               # (collectives inside the eager steps: a one-sided failure cannot be survived in place)
        roofline_error, ktimes = _err(e)["error"], {}
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
    # ---- the line exists from here on: the headline region, its repeats and the per-kernel pass are measured.  Everything
    # below decorates it and runs where it cannot take it down (side legs: child processes; the N > 1 exchange report: a child job)
    res = None
    allreduce_form = (("direct (csrc/sw_comm.hip)" if getattr(tr, "_direct", None) is not None else "process group")
                      if pg is not None else None)
    exchange_probe = getattr(tr, "exchange_probe", None)
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
        if not rows:        # the per-kernel pass failed (roofline.error says why): the roofline object keeps its keys, valued null
            rows = [{"name": None, "launches_per_step": 0.0, "avg_us": float("nan"), "us_per_step": float("nan"),
                     "alg_gflop_per_step": None, "frac": None}]
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
                       "allreduce": allreduce_form,
                       "collectives_probe": probe,      # N > 1 on RCCL: the child job that tried in-graph collectives before the headline
                       "exchange": None,          # N > 1: us per all-reduce of each bucket on both forms + the step on the direct form (below)
                       "exchange_probe": exchange_probe,      # SW_ALLREDUCE=auto: what the probe measured / chose
                       "agent_steps_s": value * B if args.scaling == "weak" else value * leg.Bg,      # agents stepped per second, whole job
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
                       "other_workloads": None},
            "roofline": {"bound": "mfma", "kernel": top["name"], "achieved": achieved, "error": roofline_error,
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

    import shutil
    import signal
    import tempfile

    def emit():
        ctypes.CDLL(None).fflush(None)          # RCCL's version banner sits in the C stdio buffer: keep the JSON line last
        print(json.dumps(res), flush=True)

    if rank == 0:
        def on_term(signum, frame):             # told to stop while a side leg runs: the line goes out as it stands
            res["config"]["terminated"] = "signal %d during the side legs: the line as it stood" % signum
            ch = _CHILD["proc"]
            if ch is not None:
                try:
                    os.killpg(ch.pid, signal.SIGKILL)
                except OSError:
                    pass
            emit()
            os._exit(0)
        signal.signal(signal.SIGTERM, on_term)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(leg.tracks, S if args.workload != "c4" else 16, A, To, Tp)
            except Exception as e:      # noqa: BLE001
                res["cpu_baseline"] = dict(_err(e), value=None, unit="steps/s", cores=None, kind="port", sample=None)
    if world == 1 and pg is None and not args.no_other_workloads:
        tr.close()
        del tr
        leg.tr = None
        torch.cuda.empty_cache()
        tmpdir = tempfile.mkdtemp(prefix="sw_bench_")
        try:
            res["config"]["other_workloads"] = run_side_children(args, 1e3 * dt / args.steps, tmpdir)
        except Exception as e:      # noqa: BLE001
            res["config"]["other_workloads"] = _err(e)
    if pg is not None and world > 1 and os.environ.get("SW_ALLREDUCE", "") != "direct" and not args.no_other_workloads:
        # N > 1: the exchange report is a CHILD JOB of rank 0 on the same GPUs (exchange_child_main)
        ex = on_rank0(world, rank, dev, backend, tmpdir, "exchange",
                      lambda: run_child_job(args, world, tmpdir, "exchange-only", "exchange", 420.0), child_timeout(420.0) + 120.0)
        if rank == 0:
            res["config"]["exchange"] = ex
    if rank == 0:
        emit()
    if pg is not None:
        try:
            tr.close()                          # (a direct exchange of the headline run: buffers and peer mappings)
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        except Exception:      # noqa: BLE001 - the line is out
            pass
    if tmpdir is not None and rank == 0:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
