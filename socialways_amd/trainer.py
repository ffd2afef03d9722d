"""train() / test() / checkpoint of the reference (train.py:439-668) over the HIP hot path.

`SocialWaysTrainer.step()` is the body of train.py:458-554 for one packed batch:
two discriminator updates (`n_unrolling_steps`+1), one generator update, the Linear-only restore of
the unrolled-GAN surrogate (train.py:498-499, 541-543) and the ADE/FDE partial sums.
Differences to the reference's *formulation*, none to its results:
  * the three `predict()` calls of a step use the same noise and the same generator weights
    (train.py:473,480,507), so the rollout is computed ONCE and its saved
    activations serve the generator backward;
  * the fake and the real branch of a D update share one observation-LSTM pass;
  * gradients are written by the kernels straight into packed buffers that are the `.grad`s of
    the parameters; Adam stays torch.optim.Adam (same parameter order as train.py:379-385, so the
    optimizer state_dicts interchange with the reference's checkpoint).
Data parallelism (one process per GPU): every rank runs `step()` on a scene-aligned shard of the
packed batch with the GLOBAL batch size in the loss normalisation and the packed gradient buffers
are all-reduced (sum) before each optimizer step - 3 RCCL all-reduces per training step.
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.optim as opt

from . import _lib as L
from . import ops
from .data import shard_scenes
from .model import Discriminator, Generator, predict_cv


class PackedAdam:
    """Adam (train.py:379-385: lr, betas (0.9, 0.999), eps 1e-8, no weight decay) over ONE packed parameter
    buffer, executed by torch's own fused multi-tensor Adam kernel (`torch._fused_adam_`, the op behind
    `torch.optim.Adam(fused=True)`) on chunk views of the buffer: one kernel launch per update.  The step
    counter is a device scalar the CALLER may supply (`step(step_tensor)`): a hipGraph-replayed training step
    gets its counters from the staging kernel instead of spending a graph node per update on incrementing
    them.  Reads and writes the reference's per-parameter optimizer state_dict (train.py:659,662 / 631,634):
    `slices` = [(offset, numel, shape)] in the reference's parameter order.  Padding floats have zero
    gradients and stay zero."""

    # torch's multi-tensor kernel gives one workgroup per (tensor, 64K chunk): feed it many small views - but at
    # most MAX_CHUNKS of them, beyond which the update is split into a second kernel launch (~5 us per graph node)
    MIN_CHUNK, MAX_CHUNKS = 2048, 30      # swept 8..36 / 512..4096 on one box: flat within noise

    def __init__(self, flat, gflat, slices, lr, betas=(0.9, 0.999), eps=1e-8, capturable=False, disc_tp=0):
        self.slices = slices
        self.flat = flat
        self.gflat = gflat
        # On the GPU the update is the library's own kernel (sw_adam_packed: torch's fused Adam restated operation by
        # operation, the arithmetic the gradient-finishing kernels apply in a single process) - one launch, and the
        # Discriminator's weight images (disc_tp = its horizon) follow the update.  torch's kernel remains for CPU
        # buffers and for a checkpoint that carries weight decay (the reference never sets it, train.py:379-385).
        self.native = flat.is_cuda
        self.disc_tp = int(disc_tp)
        n = flat.numel()
        c = self.CHUNK = max(self.MIN_CHUNK, (-(-n // self.MAX_CHUNKS) + 255) // 256 * 256)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        cut = lambda t: [t[o:min(o + c, n)] for o in range(0, n, c)]
        self.ps, self.gs, self.ms, self.vs = cut(flat), cut(gflat), cut(self.m), cut(self.v)
        self.t = 0                                              # updates applied so far
        self.step_t = torch.zeros((), device=flat.device)       # device copy of `t` for self-counted updates
        self.group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                          foreach=None, capturable=capturable, differentiable=False, fused=True)

    @torch.no_grad()
    def step(self, step_tensor=None):
        """One update.  `step_tensor` = device scalar holding the 1-based index of this update (the caller
        then also advances `self.t`); without it the optimizer counts itself."""
        if step_tensor is None:
            self.t += 1
            self.step_t.fill_(float(self.t))      # eager only: a captured update always gets its index from the caller
            step_tensor = self.step_t
        g = self.group
        if self.native and not g["weight_decay"]:
            L.call("sw_adam_packed", L.ptr(self.flat), L.ptr(self.gflat), L.ptr(self.m), L.ptr(self.v), self.flat.numel(),
                   L.ptr(step_tensor), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                   self.disc_tp, L.stream())
            return
        torch._fused_adam_(self.ps, self.gs, self.ms, self.vs, [], [step_tensor] * len(self.ps), amsgrad=False,
                           lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], weight_decay=g["weight_decay"],
                           eps=g["eps"], maximize=False, grad_scale=None, found_inf=None)

    @property
    def keeps_images(self):
        """Does step() keep registered Discriminator images current (sw_adam_packed does, torch's kernel cannot)?"""
        return bool(self.native and not self.group["weight_decay"] and self.disc_tp > 0)

    @property
    def fusable(self):
        """May a gradient-finishing kernel apply this update itself (sw_disc_bwd_gan_adam / sw_gen_wgrad_adam)?  Their
        arithmetic is Adam without weight decay."""
        return not self.group["weight_decay"]

    def fused_args(self, step_tensor=None):
        """(m, v, step scalar, lr, beta1, beta2, eps) for a kernel that applies this update itself
        (sw_disc_bwd_gan_adam) - the bookkeeping of `step()` without the launch."""
        if step_tensor is None:
            self.t += 1
            self.step_t.fill_(float(self.t))
            step_tensor = self.step_t
        g = self.group
        return (self.m, self.v, step_tensor, g["lr"], g["betas"][0], g["betas"][1], g["eps"])

    def zero_grad(self, set_to_none=False):
        for t in self.gs:
            t.zero_()

    @property
    def param_groups(self):
        return [self.group]

    def state_dict(self):
        g = dict(self.group)
        g["params"] = list(range(len(self.slices)))
        state = {}
        if self.t > 0:
            for i, sl in enumerate(self.slices):
                off, n, shape = sl[:3]
                pick = lambda buf: self._true(buf[off:off + n], sl).clone()
                state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": pick(self.m), "exp_avg_sq": pick(self.v)}
        return {"state": state, "param_groups": [g]}

    @staticmethod
    def _true(flat_slice, sl):
        """A parameter's slice of a packed buffer in the reference's shape: (offset, numel, kernel shape) alone, or
        followed by (true shape, index of the true entries) for zero-padded hidden sizes (model._Packed._true)."""
        if len(sl) == 3:
            return flat_slice.view(sl[2])
        return flat_slice[sl[4].to(flat_slice.device)].view(sl[3])

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in grp:
                self.group[k] = tuple(grp[k]) if k == "betas" else grp[k]
        state = sd.get("state", {})
        self.m.zero_()
        self.v.zero_()
        self.t = 0
        for i, sl in enumerate(self.slices):
            off, n = sl[0], sl[1]
            # torch Adam keeps state only for parameters that ever received a gradient: the unmodified reference
            # runs with use_social=False (train.py:83), so its checkpoints hold no entries for the attention /
            # feature-embedder parameters (optimizer indices 0..7).  Their moments stay zero here.
            e = state.get(i, state.get(str(i)))
            if e is None:
                continue
            for buf, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq")):
                val = e[key].to(self.flat.device).reshape(-1)
                if len(sl) == 3:
                    buf[off:off + n] = val
                else:                              # zero-padded hidden size: the true entries into their padded places
                    buf[off:off + n].index_copy_(0, sl[4].to(self.flat.device), val)
            self.t = max(self.t, int(float(e["step"])))       # one shared counter: every present entry was stepped together
        self.step_t.fill_(float(self.t))


class SocialWaysTrainer:
    STEPS_PER_LAUNCH = 4    # train_epoch: consecutive packed batches of one scene layout per graph launch (step_many)
    Z_COLS = 32             # noise columns of the kernels (hidden size 64: train.py:81); smaller models are zero-padded
    _direct = None          # the direct gradient exchange (SW_ALLREDUCE=direct / auto; the fused trainer only) and what the
    exchange_probe = None   # auto mode's probe measured - class-level defaults: the wider trainers have their own __init__

    def __new__(cls, n_next=None, hidden_size=64, *args, **kw):
        """Widths above the fused kernels' 64 units and latent-code counts other than 2 (train.py:42-44, 65) train on
        the generic-width path (generic.py: the same model layer by layer, same public surface)."""
        nl = kw.get("n_latent_codes", args[6] if len(args) > 6 else 2)      # positional slot 9 of __init__'s signature
        if cls is SocialWaysTrainer and (int(hidden_size) > 64 or int(nl) != 2):
            from .generic import GenericTrainer
            from .wide import WideTrainer
            # widths that are multiples of 32: the wide path (time-step-level kernels, explicit backward, one hipGraph per
            # step); anything else the reference accepts: the layer-by-layer generic path under torch's tape
            if os.environ.get("SW_WIDE", "1") != "0" and WideTrainer.supports(hidden_size, nl, kw.get("use_variety_loss", False)):
                return object.__new__(WideTrainer)
            return object.__new__(GenericTrainer)
        return object.__new__(cls)

    def _pad_z(self, z):
        return z if z.shape[-1] == self.Z_COLS else torch.nn.functional.pad(z, (0, self.Z_COLS - z.shape[-1]))

    def __init__(self, n_next, hidden_size=64, lr_g=1e-4, lr_d=1e-3, n_unrolling_steps=1, use_social=True,
                 use_info_loss=True, loss_info_w=0.5, n_latent_codes=2, device="cuda", process_group=None,
                 fused_adam=True, use_graph=None, use_l2_loss=False, use_variety_loss=False, loss_l2_w=0.5, variety_k=20):
        self.device = L.indexed_device(device)
        self.n_next = n_next
        self.noise_len = hidden_size // 2
        self.n_unrolling_steps = n_unrolling_steps
        self.use_info_loss = use_info_loss
        self.loss_info_w = loss_info_w
        # train.py:67-69.  `use_variety_loss` reproduces train.py:527-536 AS WRITTEN: its 20 predict()
        # calls reuse the same noise (identical values) and only the k = 19 term - the L2 of AGENT 19 of
        # the packed batch - enters the loss; the 20 redundant rollouts are not executed.
        # `use_variety_loss="fixed"` is the term the reference evidently MEANT (Social-GAN's best-of-K L2):
        # `variety_k` rollouts with independent z (the step's own z is sample 0), per agent the minimum over the samples
        # of the mean squared error, averaged over the batch; the gradient reaches the arg-min sample only.  The K
        # rollouts run as ONE folded batch of K*B agents (like test()), so the generator's work grows K-fold.
        if use_variety_loss not in (False, True, "fixed"):
            raise ValueError("use_variety_loss: False, True (train.py:527-536 as written) or 'fixed'")
        self.use_l2_loss, self.use_variety_loss, self.loss_l2_w = use_l2_loss, use_variety_loss, loss_l2_w
        self.variety_k = int(variety_k)
        self.last_variety = None             # fixed mode: (per-agent minimum (B,), arg-min sample (B,) int32) of the last step
        # construction order = train.py:370-385 (RNG -> init mapping, optimizer parameter order)
        self.G = Generator(hidden_size, 1, use_social=use_social, device=self.device)
        self.G.unify()
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        if use_graph is None:          # hipGraph replay of the step (segmented around the all-reduces when world > 1)
            use_graph = self.device.type == "cuda" and fused_adam
        self.use_graph = bool(use_graph)
        self.max_graphs = int(os.environ.get("SW_MAX_GRAPHS", "8"))    # captured packed-batch layouts (the rest of the steps run eagerly)
        self._graphs = {}
        self._force_dist = os.environ.get("SW_FORCE_DIST", "") == "1"   # 1-rank group still runs the collectives (tests)
        self._fuse_d_adam = True      # D's Adam inside the gradient reduction (single process; tests switch it off to compare)
        self._fuse_g_adam = True      # ... and the generator's (needs the weight images)
        # Data parallel: are the RCCL all-reduces recorded INSIDE the step graph (one launch for K steps) or run
        # eagerly between graph segments (3 segment boundaries per step)?  SW_GRAPH_COLLECTIVES=1 / 0 decides;
        # unset = probe once (a small captured all-reduce replayed twice and checked on every rank) and use the
        # in-graph form only if the whole group agrees that it works.
        gc_env = os.environ.get("SW_GRAPH_COLLECTIVES", "")
        self._graph_collectives = True if gc_env == "1" else False if gc_env == "0" else None
        packed = fused_adam and self.device.type == "cuda"
        if packed:
            self.predictor_optimizer = PackedAdam(self.G._flat_all, self.G._gflat_all, self.G.packed_slices(), lr_g)
        else:       # per-parameter torch Adam (CPU tests, literal autograd formulation)
            self.predictor_optimizer = opt.Adam(self.G.predictor_params(), lr=lr_g, betas=(0.9, 0.999))
        self.D = Discriminator(n_next, hidden_size, n_latent_codes, device=self.device)
        if packed:
            self.D_optimizer = PackedAdam(self.D._flat, self.D._gflat,
                                          [(off, k, tuple(p.shape)) + (self.D._true[i] if self.D._true is not None else ())
                                           for i, ((off, k), p) in enumerate(zip(self.D._slices, self.D.parameters()))], lr_d,
                                          disc_tp=n_next)
        else:
            self.D_optimizer = opt.Adam(self.D.parameters(), lr=lr_d, betas=(0.9, 0.999))
        self.pg = process_group
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        self.epoch = 0
        if self.pg is not None and self.world > 1:
            self.sync_replicas()
        # SW_ALLREDUCE=direct: the three gradient buckets of a step go through the library's own two-hop all-reduce over
        # hipIpc-mapped exchange buffers (csrc/sw_comm.hip) instead of RCCL's ring; everything else (epoch sums, broadcasts)
        # stays on the process group.  The kernel is an ordinary graph node: the step is captured as ONE graph.
        # SW_ALLREDUCE=auto: build it, check it against the group's all-reduce and time both on this node's links; use it if
        # every rank agrees that it is correct and faster (comm.probe; the verdict is kept in self.exchange_probe).
        self._direct, self.exchange_probe = None, None
        mode = os.environ.get("SW_ALLREDUCE", "")
        if (self.pg is not None and (self.world > 1 or self._force_dist) and self.device.type == "cuda"
                and mode in ("direct", "auto")):
            from . import comm
            buckets = [self.D._gflat.numel(), self.G._gflat_all.numel()]
            if mode == "direct":
                self._direct = comm.DirectAllReduce(self.pg, self.device, max(buckets))
                # no timing, but never unchecked: the exchange must reproduce the group's all-reduce on this node's links
                # before a weight depends on it (collective; SW_ALLREDUCE_CHECK=0 skips it)
                if os.environ.get("SW_ALLREDUCE_CHECK", "1") != "0" and not self._direct.check(buckets):
                    self._direct.close()
                    self._direct = None
                    raise L.SocialWaysHipError("SW_ALLREDUCE=direct: the exchange disagrees with the process group's all-reduce "
                                               "on this node (or a wait timed out); use SW_ALLREDUCE=auto or unset it")
            else:
                self._direct, self.exchange_probe = comm.probe(self.pg, self.device, buckets)
            if self._direct is not None and self._graph_collectives is None:
                self._graph_collectives = bool(self.use_graph)
        self.ws = ops.Workspaces(self.device)
        self._ws_version = 0
        # derived images of the generator's weights (composed input matrix, fc4 . fc3, transposed decoder matrices):
        # computed once per step by the staging launch instead of by every workgroup of four launches (sw_gen_images)
        self._gimg = torch.empty(L.load().sw_gen_image_floats(), device=self.device) if self.device.type == "cuda" else None
        # ... and of the discriminator's (A-operand images of weight_hh / its transpose, the transposed head matrices in the
        # backward kernels' LDS layout): scattered by the staging launch, kept current by the kernels that apply D's Adam
        # update (sw_disc_images; include/socialways_hip.h)
        self._dimg = self._dtab = None
        if self.device.type == "cuda":
            lib = L.load()
            tab = np.empty((self.D._flat.numel(), 2), dtype=np.int32)
            if lib.sw_disc_image_table(n_next, tab.ctypes.data) != 0:
                raise L.SocialWaysHipError("sw_disc_image_table(%d)" % n_next)
            self._dtab = torch.from_numpy(tab).to(self.device)
            self._dimg = torch.zeros(lib.sw_disc_image_floats(n_next), device=self.device)    # padding stays zero for good
        self._lin_mask = None
        self._lin_maskf = None
        self._noise_src = None

    # ------------------------------------------------------------------------------------------
    @property
    def use_social(self):
        return self.G.use_social

    def sync_replicas(self):
        """Data-parallel replicas must start from identical weights and optimizer state: rank 0's are broadcast (at
        construction and after load_checkpoint).  The reference has one process; here a user who does not seed every
        rank identically would otherwise train diverging replicas on all-reduced gradients."""
        dist = torch.distributed
        src = dist.get_global_rank(self.pg, 0)
        bufs = [self.G._flat_all, self.D._flat]
        for o in (self.predictor_optimizer, self.D_optimizer):
            if isinstance(o, PackedAdam):
                bufs += [o.m, o.v]
            else:
                for st in o.state.values():
                    bufs += [st["exp_avg"], st["exp_avg_sq"]]
        for b in bufs:
            dist.broadcast(b, src, group=self.pg)
        ts = torch.tensor([float(getattr(o, "t", 0)) for o in (self.predictor_optimizer, self.D_optimizer)] + [float(self.epoch)],
                          device=self.device if dist.get_backend(self.pg) == "nccl" else "cpu")
        dist.broadcast(ts, src, group=self.pg)
        for o, t in zip((self.predictor_optimizer, self.D_optimizer), ts.tolist()):
            if isinstance(o, PackedAdam):
                o.t = int(t)
                o.step_t.fill_(float(o.t))
        self.epoch = int(ts[2].item())

    def sync_rng(self):
        """train.py:471-473 draws the label-noise scalars and z from the process-global numpy / torch generators.
        Every rank must draw the SAME values for a packed batch (z is sliced per shard): rank 0's generator states
        are broadcast once per epoch, so the job as a whole follows rank 0's stream - the stream a single process
        seeded like rank 0 would follow."""
        dist = torch.distributed
        obj = [(np.random.get_state(), torch.get_rng_state())] if self.rank == 0 else [None]
        dist.broadcast_object_list(obj, src=dist.get_global_rank(self.pg, 0), group=self.pg,
                                   device=self.device if dist.get_backend(self.pg) == "nccl" else None)
        if self.rank != 0:
            np.random.set_state(obj[0][0])
            torch.set_rng_state(obj[0][1])

    def _allreduce(self, flat):
        if self.pg is not None and (self.world > 1 or self._force_dist):
            d = self._direct
            if (d is not None and flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
                    and 1024 <= flat.numel() <= d.max_floats):
                d(flat)        # SW_ALLREDUCE=direct: the two-hop exchange over peer-mapped buffers (comm.py), a plain kernel
            else:
                torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def _exchange_adam(self, opt, flat):
        """Does the direct exchange also apply this optimizer's update (sw_allreduce_direct_adam)?  A data-parallel step on
        SW_ALLREDUCE=direct with a packed, decay-free Adam."""
        return (self._direct is not None and (self.world > 1 or self._force_dist) and isinstance(opt, PackedAdam) and opt.native
                and opt.fusable and flat.numel() >= 1024)

    def _probe_graph_collectives(self):
        """Can this process group's all-reduce be recorded in a hipGraph and replayed?  Only RCCL ("nccl") is
        tried: a 4 KB SUM all-reduce is captured, replayed twice and compared with the known answer; every rank
        reports and the group takes the minimum, so all ranks choose the same step structure."""
        import ctypes
        import threading
        dev, W = self.device, self.world
        if torch.distributed.get_backend(self.pg) != "nccl":      # gloo & co. synchronise the stream: not capturable
            return False
        ok = 0.0
        # a collective that never completes would otherwise block forever in synchronize()
        guard = threading.Timer(180.0, lambda: (sys.stderr.write("socialways_amd: captured all-reduce probe hung; "
                                                                 "set SW_GRAPH_COLLECTIVES=0\n"), os._exit(3)))
        guard.daemon = True
        guard.start()
        x = torch.full((1024,), float(self.rank + 1), device=dev)
        y = torch.zeros_like(x)
        self._allreduce(y)          # eager first: connections / buffers of the communicator are set up outside capture
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    y.copy_(x)
                    self._allreduce(y)
                except Exception:
                    try:
                        g.capture_end()
                    except Exception:      # noqa: BLE001 - the capture is already invalid; make sure the stream left it
                        pass
                    hip, junk = ctypes.CDLL("libamdhip64.so"), ctypes.c_void_p()
                    hip.hipStreamEndCapture(ctypes.c_void_p(side.cuda_stream), ctypes.byref(junk))
                    hip.hipGetLastError()
                    raise
                g.capture_end()
                for _ in range(2):
                    g.replay()
            torch.cuda.synchronize()
            ok = float(bool((y == W * (W + 1) / 2.0).all()))
            del g
        except Exception as e:      # noqa: BLE001 - any failure means "do not record collectives"
            sys.stderr.write("socialways_amd: all-reduce not capturable here (%s: %s); using graph segments\n"
                             % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
        finally:
            guard.cancel()
        torch.cuda.synchronize()
        flag = torch.tensor([ok], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.pg)
        return bool(flag.item() > 0.5)

    def _resolve_collectives(self):
        """Every rank passes here at the start of every packed batch (step / step_many / _empty_step), so the
        probe's own collectives line up across the group even when some ranks have no scenes in a batch."""
        if self._graph_collectives is None and self.pg is not None and (self.world > 1 or self._force_dist):
            self._graph_collectives = bool(self.use_graph) and self._probe_graph_collectives()

    def release_graphs(self):
        """Drop every captured step graph (they hold the communicator's recorded collectives: release them before
        the process group is destroyed)."""
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        self._graphs.clear()
        self.ws.release_retired()
        self._ws_version = self.ws.version

    def close(self):
        """End of this trainer's life: captured graphs dropped, the direct exchange's buffers freed and its peer mappings
        closed (collective when a direct exchange exists: every rank calls it).  Idempotent."""
        self.release_graphs()
        d, self._direct = self._direct, None
        if d is not None:
            d.close()

    def _z_resident(self, batches):
        """Is the z of every step of this launch a contiguous fp32 device tensor of the kernels' width?  Then the
        staging kernel reads it from HBM through its pointer (sw_stage_step_zdev) instead of pulling its values over PCIe."""
        return all(nz.is_cuda and nz.dtype == torch.float32 and nz.is_contiguous() and nz.device == self.device
                   and tuple(nz.shape) == (b[0].shape[0], self.Z_COLS) for b in batches for nz in (b[4],))

    def _graph_key(self, scenes, To, ss, Bg, K, zdev=False):
        """Everything a captured step bakes in besides the buffer addresses: the scene layout, the loss switches and
        weights, the unrolling depth and the optimizer hyper-parameters (host-side scalars of the recorded launches)."""
        og, od = self.predictor_optimizer.param_groups[0], self.D_optimizer.param_groups[0]
        return (scenes.key, To, float(ss), float(Bg), self._row0, K, self.n_unrolling_steps, self.use_info_loss,
                self.loss_info_w, self.use_l2_loss, self.use_variety_loss, self.loss_l2_w, self.variety_k,
                og["lr"], tuple(og["betas"]), og["eps"], og.get("weight_decay", 0), od["lr"], tuple(od["betas"]), od["eps"],
                od.get("weight_decay", 0), bool(zdev))

    def _graphs_current(self):
        """A workspace outgrown since the last capture means captured graphs hold retired addresses: they stay valid
        (the retired tensors are alive) but pin memory, so all graphs are dropped and re-captured on the new buffers."""
        if self._ws_version != self.ws.version:
            self.release_graphs()

    def step(self, obsv, pred, sub_batches, zeros_val, ones_val, noise, ss=1.0, global_B=None, out=None, global_row0=0,
             variety_noise=None):
        """One packed batch (train.py:458-554) on this rank's rows.  obsv (B,To,2), pred (B,Tp,2) and
        noise (B,32) are tensors (noise normally lives on the host, like train.py:473); `global_B` = agents
        of the whole packed batch over all ranks.
        Returns a (U+3, 3) float64 device tensor: rows = the U+1 D updates, the G phase, ADE/FDE; loss rows
        hold SUMS of squared errors over the local rows [label_a, code, label_b], the last row
        [sum err / Tp, sum err[:, -1], sum err^2].  The kernels leave one partial triple per 16-agent tile
        (no reduction kernels on the critical path); they are summed here in float64.  `out=False` returns
        the raw (U+3, tiles, 3) partials without that extra reduction (bench loop).
        With `use_graph` the whole step - ~35 kernels incl. the input staging and the Adam updates - is
        captured once per batch layout into a hipGraph and replayed: per step the host only fills a pinned
        slot (pointers of the track slices, z, the two label-noise scalars) and launches the graph."""
        self._resolve_collectives()
        B = obsv.shape[0]
        Bg = float(global_B if global_B is not None else B)
        dev = self.device
        self._row0 = int(global_row0)       # first row of this rank's shard in the packed batch (variety term only)
        if self.use_variety_loss is True and Bg < 20:
            raise ValueError("use_variety_loss indexes agent 19 of the packed batch (train.py:531): batch of %d" % Bg)
        self._vnoise = None
        if self.use_variety_loss == "fixed":          # z of the samples 1..K-1: ((K-1)*B, 32), drawn like train.py:473 if not given
            vn = variety_noise if variety_noise is not None else torch.rand((self.variety_k - 1) * B, self.noise_len)
            if vn.shape != ((self.variety_k - 1) * B, self.noise_len):
                raise ValueError("variety_noise must be ((variety_k - 1) * B, %d)" % self.noise_len)
            self._vnoise = self._pad_z(vn.to(dev, non_blocking=True)).contiguous()
        part = None
        if not self._graphs and self.ws.retired:      # eager-only runs: nothing captured can reference an outgrown workspace
            self.ws.release_retired()                 # (stream-ordered allocator: kernels already queued on it stay valid)
        if self.use_graph and self.use_variety_loss != "fixed":    # the folded K-sample step runs eagerly
            # one graph set per packed-batch layout; datasets with ragged scenes produce many layouts, so the
            # number of captured layouts is capped and the rest of the steps run eagerly
            scenes = ops.SceneIndex.get(sub_batches, B, dev)
            self._graphs_current()
            zdev = self._z_resident([(obsv, pred, zeros_val, ones_val, noise)])
            if self._graph_key(scenes, obsv.shape[1], ss, Bg, 1, zdev) in self._graphs or len(self._graphs) < self.max_graphs:
                part = self._step_graph([(obsv, pred, zeros_val, ones_val, noise)], sub_batches, float(ss), Bg)[0]
        if part is None:
            part = torch.zeros(self.n_unrolling_steps + 3, (B + 15) // 16, 3, device=dev)     # one triple per 16-agent tile
            scenes = ops.SceneIndex.get(sub_batches, B, dev)
            noise = self._pad_z(noise.to(dev, non_blocking=True)).contiguous()
            # label-noise scalars of train.py:471-472 live in device memory: [zeros_val, ones_val]
            targets = torch.tensor([float(zeros_val), float(ones_val)], dtype=torch.float32).to(dev, non_blocking=True)
            self._step_impl(obsv.contiguous(), pred.contiguous(), None, scenes, targets, noise, float(ss), Bg, part)
        if out is False:           # caller reads the static partials before the next step overwrites them
            return part
        return part.sum(1, dtype=torch.float64)

    def step_many(self, batches, sub_batches, ss=1.0, global_B=None, out=None, global_row0=0):
        """K consecutive training steps on K packed batches of the SAME scene layout in ONE graph launch:
        `batches` = [(obsv, pred, zeros_val, ones_val, noise), ...].  Exactly the K `step()` calls in order (same
        kernels, same results); what it saves is the gap between two graph launches (~13 us, the system-scope
        fence at the end of a hipGraph) on K-1 of the K steps.  Returns the list of the K step results."""
        self._resolve_collectives()
        if not self.use_graph or len(batches) == 1 or self.use_variety_loss:
            return [self.step(o, p, sub_batches, zv, ov, nz, ss, global_B, out, global_row0) for o, p, zv, ov, nz in batches]
        B = batches[0][0].shape[0]
        Bg = float(global_B if global_B is not None else B)
        self._row0 = int(global_row0)
        scenes = ops.SceneIndex.get(sub_batches, B, self.device)
        self._graphs_current()
        key = self._graph_key(scenes, batches[0][0].shape[1], ss, Bg, len(batches), self._z_resident(batches))
        if key not in self._graphs and len(self._graphs) >= self.max_graphs:
            return [self.step(o, p, sub_batches, zv, ov, nz, ss, global_B, out, global_row0) for o, p, zv, ov, nz in batches]
        parts = self._step_graph(batches, sub_batches, float(ss), Bg)
        return parts if out is False else [q.sum(1, dtype=torch.float64) for q in parts]

    def _step_graph(self, batches, sub_batches, ss, Bg):
        K = len(batches)
        B, To, Tp = batches[0][0].shape[0], batches[0][0].shape[1], self.n_next
        dev = self.device
        scenes = ops.SceneIndex.get(sub_batches, B, dev)
        zdev = self._z_resident(batches)
        key = self._graph_key(scenes, To, ss, Bg, K, zdev)
        st = self._graphs.get(key)
        HDR = 8                                                   # SW_STAGE_HEADER words in front of z
        if st is None:
            st = self._graphs[key] = dict(
                n=0, graph=None, flip=0, scenes=scenes, obsv=torch.empty(B, To, 2, device=dev),
                pred=torch.empty(B, Tp, 2, device=dev), pred4=torch.empty(B, Tp, 4, device=dev),
                targets=torch.empty(4, device=dev), noise=torch.zeros(B, self.Z_COLS, device=dev),
                steps=torch.zeros(self.n_unrolling_steps + 2, device=dev),   # Adam step indices of the U+1 D updates, the G update
                outs=[torch.zeros(self.n_unrolling_steps + 3, (B + 15) // 16, 3, device=dev) for _ in range(K)],
                slots=[[torch.zeros(HDR + B * self.Z_COLS, dtype=torch.float32).pin_memory() for _ in range(K)]
                       for _ in range(2)],
                done=[torch.cuda.Event(), torch.cuda.Event()], keep=[None, None])
        # Inputs of a step travel through a pinned host slot that the step's first graph node (sw_stage_step)
        # reads itself: the device pointers of this batch's track slices, the two label-noise scalars, the Adam
        # counters and z.  A hipMemcpyAsync enqueued behind graph launches would block the host until the stream
        # drains; a kernel reading device-mapped host memory does not.  One slot per (graph executable, sub-step).
        k = (st["flip"] ^ 1) if st["graph"] is not None else 0
        st["done"][k].synchronize()                            # the replay that last read these slots has finished
        packed = isinstance(self.D_optimizer, PackedAdam)
        keep = []
        for j, (obsv, pred, zeros_val, ones_val, noise) in enumerate(batches):
            obsv, pred = obsv.contiguous(), pred.contiguous()
            keep.append((obsv, pred))                          # alive until the slot is rewritten
            hn = st["slots"][k][j].numpy()
            hn[:4].view(np.uint64)[:] = (obsv.data_ptr(), pred.data_ptr())
            hn[4], hn[5] = float(zeros_val), float(ones_val)
            if packed:        # updates applied so far: the staging kernel turns them into this step's Adam step indices
                hn[6], hn[7] = float(self.D_optimizer.t), float(self.predictor_optimizer.t)
                self.D_optimizer.t += self.n_unrolling_steps + 1
                self.predictor_optimizer.t += 1
            if zdev:          # z already in HBM: only its address travels
                hn[HDR:HDR + 2].view(np.uint64)[:] = noise.data_ptr()
                keep.append(noise)
                continue
            # (a hidden size below 64 draws fewer z columns: the kernels' remaining columns stay zero, like their weights)
            np.copyto(hn[HDR:].reshape(B, self.Z_COLS)[:, :self.noise_len], (noise.cpu() if noise.is_cuda else noise).numpy())
        st["keep"][k] = keep

        def stage(kk, j):
            slot = st["slots"][kk][j]
            L.call("sw_stage_step_zdev", slot.data_ptr(), B, To, Tp, L.ptr(st["obsv"]), L.ptr(st["pred"]),
                   L.ptr(st["pred4"]), L.ptr(st["targets"]), L.ptr(st["noise"]) if zdev else None, L.ptr(st["steps"]),
                   self.n_unrolling_steps + 1,
                   L.ptr(self.G.encoder._flat), L.ptr(self.G.decoder._flat), L.ptr(self.G.feature_embedder._flat),
                   L.ptr(self.G.attention._flat), L.ptr(self._gimg),
                   L.ptr(self.D._flat) if self._dimg is not None else None, L.ptr(self._dimg), L.ptr(self._dtab), int(zdev),
                   L.stream())
            # z in host memory: pulled by idle workgroups of the encoder launch; z in HBM: copied by this staging launch
            self._noise_src = None if zdev else slot.data_ptr() + 4 * HDR

        def args(j):
            return (st["obsv"], st["pred"], st["pred4"], scenes, st["targets"], st["noise"], ss, Bg, st["outs"][j],
                    st["steps"] if packed else None)

        def body(kk):          # the K steps of one executable, back to back
            for j in range(K):
                yield from self._step_gen(*args(j), pre=lambda j=j: stage(kk, j))
        if st["graph"] is not None:
            st["flip"] = k
            for g, buf in st["graph"][k]:
                g.replay()
                if buf is not None:
                    self._allreduce(buf)
        elif st["n"] < 2:          # first steps of a layout run eagerly (lazy inits, workspace growth)
            st["n"] += 1
            for j in range(K):
                stage(0, j)
                self._step_impl(*args(j))
        else:
            # Capture.  Single GPU: one graph for the K steps.  Data parallel: the same with the RCCL
            # all-reduces recorded in it when the group can do that (see __init__), else one graph per
            # segment between the all-reduce points with the collectives run eagerly in between, all
            # segments sharing one memory pool so intermediates stay alive.
            # Everything is captured TWICE and the two executables alternate: launching an executable
            # that is still running makes hipGraphLaunch wait for it, which would put the host-side
            # launch cost (~150 us for ~40 nodes) on the critical path of every step.
            torch.cuda.synchronize()
            pool, sets = None, []
            for kk in range(2):
                gen = body(kk)
                graphs, fin = [], False
                while not fin:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: other threads (the RCCL watchdog) may touch the runtime during capture
                    with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                        while True:
                            try:
                                buf = next(gen)
                            except StopIteration:
                                buf, fin = None, True
                            if fin:
                                break
                            if self.world > 1 or self._force_dist:
                                if not self._graph_collectives:
                                    break                      # segment boundary: the all-reduce runs eagerly
                                self._allreduce(buf)           # opt-in: RCCL all-reduce recorded inside the graph
                    graphs.append((g, buf))
                    pool = g.pool()
                sets.append(graphs)
            st["graph"], st["flip"] = sets, 0
            for g, buf in sets[0]:     # capture only records: this replay IS the step
                g.replay()
                if buf is not None:
                    self._allreduce(buf)
        st["done"][k].record()
        return st["outs"]

    def _step_impl(self, obsv, pred, pred4, scenes, targets, noise, ss, Bg, out, steps=None):
        """Eager step: run the segments, all-reducing the packed gradient buffer each one hands back."""
        for buf in self._step_gen(obsv, pred, pred4, scenes, targets, noise, ss, Bg, out, steps):
            self._allreduce(buf)
        return out

    def _step_gen(self, obsv, pred, pred4, scenes, targets, noise, ss, Bg, out, steps=None, pre=None):
        """Device-only body of the step (no host syncs, no host-dependent values: capturable) as a
        generator: it yields the packed gradient buffer at each of the 3 points where data-parallel
        ranks must all-reduce before the optimizer step (D, D, G).  `out` (U+3, tiles, 3) receives the
        per-tile loss / ADE partial sums; `steps` (U+2 device scalars) = the Adam step indices of this
        step's updates when the staging kernel provides them; `pre` = the input staging of a captured step."""
        G, D = self.G, self.D
        B, Tp = obsv.shape[0], self.n_next
        dev = self.device
        ws = self.ws
        U = self.n_unrolling_steps
        g_label = 1.0 / Bg
        g_code = (self.loss_info_w if self.use_info_loss else 0.0) / (2.0 * Bg)
        # One stream: inside a hipGraph every cross-stream edge costs 5-10 us of queue synchronisation on
        # this runtime - more than any of the small kernels that could be overlapped (measured).
        if pre is not None:
            pre()
        else:                            # eager step without a staging launch: derive the weight images here
            if self._gimg is not None:
                L.call("sw_gen_images", L.ptr(G.encoder._flat), L.ptr(G.decoder._flat), L.ptr(G.feature_embedder._flat),
                       L.ptr(G.attention._flat), L.ptr(self._gimg), L.stream())
            self._disc_images()
        try:
            yield from self._step_body(obsv, pred, pred4, scenes, targets, noise, ss, Bg, out, steps)
        finally:
            if self._gimg is not None:   # the generator's Adam step follows / has run: the images are stale
                L.call("sw_gen_images", None, None, None, None, None, None)
            if self._dimg is not None:   # D.load(backup) has run (train.py:541-542) / the weights are the caller's again
                L.call("sw_disc_images", None, None, None, 0, None)

    def _disc_images(self):
        """(Re-)scatter the packed D weights into their images and register them (sw_disc_images)."""
        if self._dimg is not None:
            L.call("sw_disc_images", L.ptr(self.D._flat), L.ptr(self._dimg), L.ptr(self._dtab), self.n_next, L.stream())

    def _step_body(self, obsv, pred, pred4, scenes, targets, noise, ss, Bg, out, steps):
        G, D = self.G, self.D
        B, Tp = obsv.shape[0], self.n_next
        dev = self.device
        ws = self.ws
        U = self.n_unrolling_steps
        g_label = 1.0 / Bg
        g_code = (self.loss_info_w if self.use_info_loss else 0.0) / (2.0 * Bg)
        noise_src, self._noise_src = self._noise_src, None      # set by the staging of this step (graph / warm-up path)
        if pred4 is None:          # real future as 4-d (train.py:470); the observation stays 2-d: kernels form (p, v) on the fly
            pred4 = torch.empty(B, Tp, 4, device=dev)
            o4_scratch = ws.get("o4", B * obsv.shape[1] * 4)
            L.call("sw_traj_4d", L.ptr(obsv), L.ptr(pred), B, obsv.shape[1], Tp, L.ptr(o4_scratch), L.ptr(pred4), L.stream())
        # ---- generator rollout, once (train.py:480/507 are identical) ---------------
        enc, emb, att, dec = G.encoder, G.feature_embedder, G.attention, G.decoder
        KV = self.variety_k if self.use_variety_loss == "fixed" else 1
        if KV > 1:
            # best-of-K variety term: the K rollouts are K copies of the scenes in one batch; copy 0 (the step's own z)
            # is the prediction every other loss term and the ADE/FDE sums see
            # (the encoder over the observed steps and the social pooling do not depend on z: once for all K copies)
            pred_hat_k, gctx = ops.gen_forward_k(enc._flat, emb._flat, att._flat, dec._flat, obsv,
                                                 torch.cat([noise, self._vnoise]), scenes, Tp, G.use_social, KV, ws=ws)
            pred_hat = pred_hat_k[:B]
            out[U + 2].zero_()
            L.call("sw_ade_fde", L.ptr(pred_hat), L.ptr(pred), B, Tp, 1.0 / float(ss), L.ptr(out[U + 2]),
                   L.ptr(ws.get("ade_scratch", 3 * L.RED_BLOCKS)), L.stream())
            d_pre = None
        else:
            # the decode kernel also leaves the ADE/FDE partial sums of the prediction (train.py:546-551)
            # ... and, while it leaves CUs idle, the observation LSTM of the first D pass (independent of the generator)
            d_pre = ops.d_obs_buffer(ws, B, obsv.shape[1], Tp) if obsv.shape[2] == 2 else None
            pred_hat, gctx = ops.gen_forward(enc._flat, emb._flat, att._flat, dec._flat, obsv, noise, scenes, Tp,
                                             G.use_social, save=True, ws=ws, ade=(pred, 1.0 / float(ss), out[U + 2]),
                                             noise_src=noise_src, d_obs=(D._flat, d_pre) if d_pre is not None else None)
        d_gflat = D.grad_views()
        backup = None
        # ---- discriminator updates (train.py:476-499) ------------------------------------------------
        for u in range(self.n_unrolling_steps + 1):
            if u == 1:     # deepcopy(D) after the first update (train.py:498-499) = the weights of this forward pass
                backup = ws.get("d_backup", D._flat.numel())
            # the loss gradients AND the reported loss sums (per-tile partials) are formed inside the backward kernel;
            # in a single process (no all-reduce between gradient and update) D's Adam step rides in the kernel that
            # finishes the gradients
            fuse = (self._fuse_d_adam and isinstance(self.D_optimizer, PackedAdam) and self.D_optimizer.fusable
                    and not (self.world > 1 or self._force_dist))
            adam = self.D_optimizer.fused_args(None if steps is None else steps[u]) if fuse else None
            if obsv.shape[2] == 2 and ops.disc_update_supported(D._flat, B, obsv.shape[1], Tp):
                # shapes that leave CUs idle: forward + loss gradients + backward of the pass in ONE launch (sw_disc_update)
                ops.disc_update(D._flat, obsv, [pred_hat, pred4], targets, (0, 1), noise, g_label, g_code, d_gflat, ws,
                                obs_pre=(u == 0 and d_pre is not None), w_snapshot=backup if u == 1 else None,
                                loss_part=out[u], adam=adam)
            else:
                labels, codes, dctx = ops.disc_forward(D._flat, obsv, [pred_hat, pred4], save=True, ws=ws,
                                                       save_lstm=2 if (u == 0 and d_pre is not None) else 1,
                                                       w_snapshot=backup if u == 1 else None)
                ops.disc_backward_gan(D._flat, dctx, labels, codes, targets, (0, 1), noise, g_label, g_code, d_gflat, (), ws=ws,
                                      loss_part=out[u], adam=adam)
            if self._exchange_adam(self.D_optimizer, d_gflat):
                # SW_ALLREDUCE=direct: gradient exchange + D's Adam update in ONE launch (no all-reduce point to hand back)
                self._direct.adam(d_gflat, self.D_optimizer, None if steps is None else steps[u])
                continue
            yield d_gflat
            if not fuse:
                self.D_optimizer.step() if steps is None else self.D_optimizer.step(steps[u])
                if not getattr(self.D_optimizer, "keeps_images", False):
                    self._disc_images()        # an update the image table did not see (torch's Adam): scatter again
        # ---- generator update (train.py:503-539) ----------------------------------------------------
        # D forward on the prediction + backward of its heads down to d(g_loss)/d(pred_hat), one launch, nothing saved
        # ... which is tile-local and feeds only the decode BPTT of the same tile: in the plain step (no extra terms on
        # d/d(pred_hat)) it runs inside that launch (ops.gen_backward(dfuse=...): one graph node less)
        dfuse = None
        if (ops.DFUSE and KV == 1 and not self.use_l2_loss and self.use_variety_loss is False and obsv.shape[2] == 2):
            dfuse = (D._flat, pred_hat, targets, 1, noise, g_label, g_code, out[U + 1])
        dpred = None if dfuse else ops.disc_dpred(D._flat, obsv, pred_hat, targets, 1, noise, g_label, g_code, loss_part=out[U + 1])
        if self.use_l2_loss:                                                 # train.py:525-526
            L.call("sw_l2_grad", L.ptr(pred_hat), L.ptr(pred), B, Tp, 0, B, self.loss_l2_w / (Bg * Tp), L.ptr(dpred), L.stream())
        if self.use_variety_loss is True:                                    # train.py:527-536 as written
            r = 19 - self._row0
            if 0 <= r < B:
                L.call("sw_l2_grad", L.ptr(pred_hat), L.ptr(pred), B, Tp, r, r + 1, self.loss_l2_w / Tp, L.ptr(dpred), L.stream())
        if KV > 1:                                                           # ... and with its intended semantics
            dk = torch.zeros(KV * B, Tp, 4, device=dev)
            dk[:B].copy_(dpred)
            l2min, kmin = torch.empty(B, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
            L.call("sw_variety_grad", L.ptr(pred_hat_k), L.ptr(pred), KV, B, Tp, self.loss_l2_w / (Bg * Tp), L.ptr(dk),
                   L.ptr(kmin), L.ptr(l2min), L.stream())
            self.last_variety = (l2min, kmin)
            dpred = dk
        restore = None
        if self.n_unrolling_steps > 0:     # D.load(backup) restores the Linear layers only (train.py:311-316, 541-542); D is
            if self._lin_maskf is None:    # not read again in this step: done by idle workgroups of the decode BPTT launch
                self._lin_maskf = D.linear_mask().float().contiguous()
            restore = (backup[:D._flat.numel()], D._flat, self._lin_maskf)
        G.grad_views()
        # Single process: the generator's Adam step rides in the two kernels that finish its gradients (the reduction of
        # the grouped GEMM and the composition back-propagation; the latter reads the step-start weight snapshot of the
        # image buffer).  Without social problems in the launch the attention / embedder weights would miss their
        # (zero-gradient) update: torch's kernel then.
        fuse = (KV == 1 and self._fuse_g_adam and self._gimg is not None and isinstance(self.predictor_optimizer, PackedAdam)
                and self.predictor_optimizer.fusable and not (self.world > 1 or self._force_dist)
                and (not G.use_social or gctx.scenes.P > 0 or gctx.scenes.NB > 0))
        adam = None
        if fuse:
            opt = self.predictor_optimizer
            adam = (G._flat_all, G._gflat_all) + opt.fused_args(None if steps is None else steps[U + 1])
        if KV > 1:
            ops.gen_backward_k(enc._flat, emb._flat, att._flat, dec._flat, gctx, dpred, enc._gflat, emb._gflat, att._gflat,
                               dec._gflat, ws=ws, aux=restore)
        else:
            ops.gen_backward(enc._flat, emb._flat, att._flat, dec._flat, gctx, dpred, enc._gflat, emb._gflat, att._gflat,
                             dec._gflat, ws=ws, aux=restore, tag="g", adam=adam, dfuse=dfuse)
        if self._exchange_adam(self.predictor_optimizer, G._gflat_all):
            self._direct.adam(G._gflat_all, self.predictor_optimizer, None if steps is None else steps[U + 1])
        else:
            yield G._gflat_all
            if not fuse:
                self.predictor_optimizer.step() if steps is None else self.predictor_optimizer.step(steps[U + 1])
        self.last_pred_hat = pred_hat
        self.last_pred_hat_k = pred_hat_k if KV > 1 else None

    # ------------------------------------------------------------------------------------------
    def losses_from(self, out, B_global, Tp=None, ss=1.0):
        """(n_steps, U+3, 3) sums -> the reference's MSE terms in its order
        [d_fake, d_info, d_real] x (U+1), [g_l2, g_fool, g_info] per step (train.py:484-488,512-516)."""
        o = out.detach().double().cpu().numpy()
        if o.ndim == 2:
            o = o[None]
        Bg = np.asarray(B_global, dtype=np.float64).reshape(-1, 1)
        U = self.n_unrolling_steps + 1
        Tp = Tp or self.n_next
        cols = []
        for u in range(U):
            cols += [o[:, u, 0:1] / Bg, o[:, u, 1:2] / (2 * Bg), o[:, u, 2:3] / Bg]
        cols += [o[:, U + 1, 2:3] * (ss * ss) / (2 * Tp * Bg), o[:, U, 0:1] / Bg, o[:, U, 1:2] / (2 * Bg)]
        return np.concatenate(cols, axis=1)

    def train_epoch(self, data, batch_size, draw=None):
        """train() (train.py:439-557).  `draw(bs)` -> (zeros_val, ones_val, noise_cpu) overrides the
        RNG draws of train.py:471-473 (tests feed the reference's recorded values)."""
        outs, sizes = [], []
        pend, pend_key = [], None        # consecutive packed batches of one layout share a graph launch
        if self.world > 1 and draw is None:
            self.sync_rng()
        if self._direct is not None and self.world > 1:
            # the exchange kernels WAIT for their peers on the device (bounded: SW_COMM_TIMEOUT_S): ranks enter an epoch
            # together, whatever one of them did alone in between (test(), save(), a capture)
            torch.distributed.barrier(group=self.pg)

        def flush():
            nonlocal pend, pend_key
            if pend:
                outs.extend(self.step_many([p[0] for p in pend], pend[0][1], data.ss, global_B=pend[0][2],
                                           global_row0=pend[0][3]))
                pend, pend_key = [], None
        for a, b, sb in data.packed_steps(batch_size):
            bs = b - a
            if draw is None:
                zv = np.random.uniform(0, 0.1)                               # train.py:471
                ov = np.random.uniform(0.9, 1.0)                             # train.py:472
                noise = torch.rand(bs, self.noise_len)                       # train.py:473 (CPU generator)
            else:
                zv, ov, noise = draw(bs)
            vn = None
            if self.use_variety_loss == "fixed":      # z of the extra samples, drawn for the whole packed batch
                vn = torch.rand(self.variety_k - 1, bs, self.noise_len)
            sizes.append((bs, len(sb)))
            r0, r1, sbl = 0, bs, sb
            if self.world > 1:      # this rank's scene-aligned shard of the packed batch (z / label noise: global draws, sliced)
                lo, hi = shard_scenes(sb, self.world)[self.rank]
                if hi <= lo:        # no scene for this rank: it still takes part in the step's three all-reduces
                    flush()
                    outs.append(self._empty_step())
                    continue
                r0, r1 = int(sb[lo, 0]), int(sb[hi - 1, 1])
                sbl = sb[lo:hi] - r0
            item = (data.obsv[a + r0:a + r1], data.pred[a + r0:a + r1], zv, ov, noise[r0:r1])
            if vn is not None:        # the folded K-sample step is not graph-captured: one step() per packed batch
                flush()
                outs.append(self.step(*item[:2], sbl, zv, ov, item[4], data.ss, global_B=bs, global_row0=r0,
                                      variety_noise=vn[:, r0:r1].reshape(-1, self.noise_len)))
                continue
            # consecutive packed batches with the same local layout share one graph launch (step_many)
            key = (bs, r0, np.asarray(sbl).tobytes())
            if key != pend_key or len(pend) == self.STEPS_PER_LAUNCH:
                flush()
                pend_key = key
            pend.append((item, sbl, bs, r0))
        flush()
        allo = torch.stack(outs)
        self._allreduce(allo)
        if self._direct is not None and self._direct.status_all() != 0:      # collective: every rank raises, none hangs
            raise L.SocialWaysHipError("the direct gradient exchange timed out waiting for a peer during this epoch on at "
                                       "least one rank (sw_comm_status): a rank whose wait times out publishes nothing and "
                                       "applies no update from it, so the replicas are no longer identical - restart from "
                                       "the last checkpoint (SW_COMM_TIMEOUT_S sets the wait, default 30 s)")
        o = allo.double().cpu().numpy()
        ade = float(o[:, -1, 0].sum() / data.n_train_samples)
        fde = float(o[:, -1, 1].sum() / data.n_train_samples)
        losses = self.losses_from(allo, [s[0] for s in sizes], data.n_next, data.ss)
        self.epoch += 1
        return ade, fde, losses, sizes

    def _empty_step(self):
        """A rank without scenes in this packed batch still takes part in the 3 all-reduces."""
        self._resolve_collectives()
        d_g = self.D.grad_views()
        self.G.grad_views()
        for u in range(self.n_unrolling_steps + 1):
            d_g.zero_()
            self._allreduce(d_g)
            self.D_optimizer.step()
            if u == 0 and self.n_unrolling_steps > 0:
                backup = self.D._flat.clone()
        self.G._gflat_all.zero_()
        self._allreduce(self.G._gflat_all)
        self.predictor_optimizer.step()
        if self.n_unrolling_steps > 0:
            if self._lin_mask is None:
                self._lin_mask = self.D.linear_mask() > 0
            self.D._flat.copy_(torch.where(self._lin_mask, backup, self.D._flat))
        return torch.zeros(self.n_unrolling_steps + 3, 3, device=self.device, dtype=torch.float64)

    # ------------------------------------------------------------------------------------------
    TEST_CHUNK = 16384      # agent copies (K x agents) per rollout launch of test(): scenes are folded up to this many

    def test(self, data, n_gen_samples=20, linear=False, write_to_file=None, just_one=False, collect=None):
        """test() (train.py:563-616): K sampled futures per held-out scene, avg / min-over-K ADE & FDE,
        optional prediction npz ('<epoch>-<t>.npz': timestamp, obsvs, preds_our, preds_gtt, preds_lnr,
        all denormalised - the schema visualize.py / calc_statistics.py read).  The K rollouts of a
        scene are independent given the noise, so they run as ONE batch of K*n agents with K copies
        of the scene, and consecutive held-out scenes are folded into the same launch (block-diagonal social
        block: a scene's rollout does not depend on its neighbours in the batch) up to TEST_CHUNK agent copies:
        identical rollouts, one launch sequence and one host sync per chunk instead of per scene.  The noise is
        drawn scene by scene, K draws of (n, noise_len) each, in the reference's order (train.py:584)."""
        ss, dev, K = data.ss, self.device, n_gen_samples
        sums = torch.zeros(4, dtype=torch.float64, device=dev)          # ade_avg, fde_avg, ade_min, fde_min
        batches = [(int(b[0]), int(b[1])) for b in data.test_batches]
        if just_one:
            batches = batches[:1]
        i = 0
        while i < len(batches):
            j, tot = i + 1, batches[i][1] - batches[i][0]
            while (j < len(batches) and batches[j][0] == batches[j - 1][1]
                   and (tot + batches[j][1] - batches[j][0]) * K <= self.TEST_CHUNK):
                tot += batches[j][1] - batches[j][0]
                j += 1
            lo, hi = batches[i][0], batches[j - 1][1]
            obsv, pred = data.obsv[lo:hi], data.pred[lo:hi]
            n = hi - lo
            with torch.no_grad():
                linear_preds = predict_cv(obsv, self.n_next)
                if linear and not write_to_file:
                    preds_k = linear_preds.unsqueeze(0)
                    errs = torch.pow((linear_preds[:, :, :2] - pred) / ss, 2).sum(dim=2, keepdim=True).sqrt().unsqueeze(0)
                else:
                    # copy k of scene s sits at rows k * n + (scene rows): K copies of the chunk, each copy its scenes
                    noise = torch.empty(K, n, self.noise_len)
                    for a, b in batches[i:j]:
                        for k in range(K):
                            noise[k, a - lo:b - lo] = torch.rand(b - a, self.noise_len)      # train.py:584, scene by scene
                    sb1 = np.asarray([[a - lo, b - lo] for a, b in batches[i:j]], dtype=np.int64)
                    sb = np.concatenate([sb1 + k * n for k in range(K)])
                    ph = self.G(obsv.repeat(K, 1, 1), noise.view(K * n, -1).to(dev), self.n_next, sb).view(K, n, self.n_next, 4)
                    preds_k = ph
                    errs = torch.pow((ph[:, :, :, :2] - pred.unsqueeze(0)) / ss, 2).sum(dim=3, keepdim=True).sqrt()
                if write_to_file or collect is not None:
                    sc = data.scale
                    for si, (a, b) in enumerate(batches[i:j]):
                        t = data.times[a] if data.times is not None else i + si
                        r = slice(a - lo, b - lo)
                        rec = dict(timestamp=t, obsvs=sc.denormalize(obsv[r, :, :2].cpu().numpy()),
                                   preds_our=sc.denormalize(preds_k[:, r, :, :2].cpu().numpy()),
                                   preds_gtt=sc.denormalize(pred[r, :, :2].cpu().numpy()),
                                   preds_lnr=sc.denormalize(linear_preds[r, :, :2].cpu().numpy()))
                        if collect is not None:
                            collect.append(rec)
                        if write_to_file:
                            os.makedirs(write_to_file, exist_ok=True)
                            np.savez(os.path.join(write_to_file, str(self.epoch) + '-' + str(t) + '.npz'), **rec)
                e = errs.double()
                sums += torch.stack([e.mean(2).mean(0).sum(), e[:, :, -1].mean(0).sum(),
                                     e.mean(2).min(0)[0].sum(), e[:, :, -1].min(0)[0].sum()])
            i = j
        n = data.n_test_samples
        ade_avg, fde_avg, ade_min, fde_min = (sums / n).tolist()
        return ade_avg, fde_avg, ade_min, fde_min

    # ------------------------------------------------------------------------------------------
    def checkpoint(self, epoch=None):
        """The reference's checkpoint dict (train.py:653-663)."""
        G = self.G
        return {'epoch': self.epoch if epoch is None else epoch,
                'attentioner_dict': G.attention.state_dict(),
                'feature_embedder_dict': G.feature_embedder.state_dict(),
                'encoder_dict': G.encoder.state_dict(),
                'decoder_dict': G.decoder.state_dict(),
                'pred_optimizer': self.predictor_optimizer.state_dict(),
                'D_dict': self.D.state_dict(),
                'D_optimizer': self.D_optimizer.state_dict()}

    def save(self, path, epoch=None):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(self.checkpoint(epoch), path)

    def load_checkpoint(self, ck):
        """Resume (train.py:622-634); optimizer entries are optional (weights-only dicts load too)."""
        if isinstance(ck, (str, os.PathLike)):
            ck = torch.load(ck, map_location=self.device)
        G = self.G
        G.attention.load_state_dict(ck['attentioner_dict'])
        G.feature_embedder.load_state_dict(ck['feature_embedder_dict'])
        G.encoder.load_state_dict(ck['encoder_dict'])
        G.decoder.load_state_dict(ck['decoder_dict'])
        self.D.load_state_dict(ck['D_dict'])
        for key, optim in (('pred_optimizer', self.predictor_optimizer), ('D_optimizer', self.D_optimizer)):
            if key in ck:
                if isinstance(optim, PackedAdam):
                    optim.load_state_dict(ck[key])
                    continue
                keep = {k: optim.param_groups[0].get(k) for k in ('fused', 'foreach', 'capturable')}
                optim.load_state_dict(copy.deepcopy(ck[key]))
                for k, v in keep.items():
                    optim.param_groups[0][k] = v
        self.epoch = int(ck.get('epoch', 0))
        if self.pg is not None and self.world > 1:
            self.sync_replicas()
        return self.epoch + 1
