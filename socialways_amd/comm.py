"""Two-hop gradient all-reduce over peer-mapped exchange buffers (csrc/sw_comm.hip, include/socialways_hip.h) - the
data-parallel step's alternative to RCCL's ring for its three small flat buckets.

`DirectAllReduce(process_group, device, max_floats)` is built once per trainer: every rank allocates an exchange buffer,
the 64-byte hipIpc handles travel through the process group (`all_gather_object`: any backend), every rank maps its peers'
buffers.  `ar(flat)` then all-reduces a flat fp32 device tensor in place on the current stream (a plain kernel launch:
capturable in the step's hipGraph).  Selected by `SW_ALLREDUCE=direct` (SocialWaysTrainer); the default stays RCCL until a
multi-GPU node has measured both.
"""
import ctypes
import socket

import torch

from . import _lib as L


class DirectAllReduce:
    def __init__(self, process_group, device, max_floats):
        """Collective over `process_group`, and FAILURE-SYMMETRIC: every rank takes part in the same sequence of group
        operations whatever fails locally (allocation, export, a peer mapping), the ranks agree on the outcome, and either
        all of them hold a working exchange or all of them raise (buffers freed, mappings closed) - a caller may catch the
        exception on every rank and fall back to the process group (comm.probe, bench.py's exchange report)."""
        dist = torch.distributed
        self.pg, self.device = process_group, L.indexed_device(device)
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self.max_floats = int(max_floats)
        self._own, self._opened, self._peers = None, [], []
        host, pid = socket.gethostname(), int(torch.multiprocessing.current_process().pid)

        def first_line(e):
            return "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        with torch.cuda.device(self.device):
            raw, err = None, None
            try:
                lib = L.load()
                nbytes = lib.sw_comm_bytes(self.world, self.max_floats)
                if nbytes <= 0:
                    raise L.SocialWaysHipError("sw_comm_bytes(%d, %d) = %d" % (self.world, self.max_floats, nbytes))
                own = ctypes.c_void_p()
                L.call("sw_comm_alloc", nbytes, ctypes.byref(own))
                self._own = own.value
                handle = ctypes.create_string_buffer(64)
                L.call("sw_comm_ipc_export", self._own, handle)
                raw = bytes(handle.raw)
            except Exception as e:      # noqa: BLE001 - reported to the peers below: they must not wait for this rank
                err = first_line(e)
            # (host name + pid: a handle must be opened by ANOTHER process of the same host); a rank that failed sends None
            everyone = [None] * self.world
            dist.all_gather_object(everyone, (host, pid, raw, err), group=process_group)
            if err is None:
                bad = [(r, h[3]) for r, h in enumerate(everyone) if h[2] is None]
                if bad:
                    err = "rank %d could not build its exchange buffer (%s)" % bad[0]
                elif any(h[0] != host for h in everyone):
                    err = "SW_ALLREDUCE=direct needs all ranks on one node (hipIpc)"
            if err is None:
                try:
                    for r, (_, _, peer_raw, _) in enumerate(everyone):
                        if r == self.rank:
                            self._peers.append(self._own)
                            continue
                        p = ctypes.c_void_p()
                        L.call("sw_comm_ipc_import", ctypes.create_string_buffer(peer_raw, 64), ctypes.byref(p))
                        self._peers.append(p.value)
                        self._opened.append(p.value)
                except Exception as e:      # noqa: BLE001
                    err = first_line(e)
            # nobody launches before every rank has mapped every buffer - and nobody keeps a half-built exchange
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, err, group=process_group)
            failed = [(r, v) for r, v in enumerate(verdicts) if v is not None]
            if failed:
                self._release()
                raise L.SocialWaysHipError("direct exchange not built: rank %d: %s" % failed[0])
            self._arr = (ctypes.c_void_p * self.world)(*self._peers)

    def _release(self):
        """Close the peer mappings and free the own buffer (local, no collective)."""
        for p in self._opened:
            try:
                L.call("sw_comm_ipc_close", p)
            except Exception:      # noqa: BLE001 - best effort on a failure path / at exit
                pass
        if self._own is not None:
            try:
                L.call("sw_comm_free", self._own)
            except Exception:      # noqa: BLE001
                pass
        self._own, self._opened, self._peers = None, [], []

    def __call__(self, flat):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() <= self.max_floats
        L.call("sw_allreduce_direct", ctypes.cast(self._arr, ctypes.c_void_p), self.rank, self.world, self.max_floats,
               L.ptr(flat), flat.numel(), L.stream())
        return flat

    def adam(self, flat_grad, opt, step_tensor=None):
        """All-reduce `flat_grad` AND apply `opt` (a PackedAdam over the packed weights that gradient belongs to) in the same
        launch; `step_tensor` = device scalar with the 1-based index of this update (None: the optimizer counts itself)."""
        m, v, st, lr, b1, b2, eps = opt.fused_args(step_tensor)
        L.call("sw_allreduce_direct_adam", ctypes.cast(self._arr, ctypes.c_void_p), self.rank, self.world, self.max_floats,
               L.ptr(flat_grad), flat_grad.numel(), L.ptr(opt.flat), L.ptr(m), L.ptr(v), L.ptr(st), float(lr), float(b1),
               float(b2), float(eps), int(opt.disc_tp), L.stream())

    def status(self):
        """0, or 1 after a wait on a peer has timed out (synchronises the whole device)."""
        st = ctypes.c_int(0)
        L.call("sw_comm_status", self._own, ctypes.byref(st))
        return st.value

    def status_all(self):
        """The status of the WHOLE group (collective: max over the ranks) - what a caller checks before it trusts the
        gradients of an epoch: a rank whose wait timed out publishes nothing, so its peers time out in turn."""
        dist = torch.distributed
        cpu_pg = dist.get_backend(self.pg) != "nccl"
        t = torch.tensor([float(self.status())], device="cpu" if cpu_pg else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return int(t.item())

    def check(self, bucket_floats, rounds=2):
        """Collective: does the exchange agree with the process group's own all-reduce on buffers of these sizes (to fp32
        summation order) on THIS node's links?  True / False, the same on every rank."""
        dist = torch.distributed
        cpu_pg = dist.get_backend(self.pg) != "nccl"
        gen = torch.Generator().manual_seed(1234 + self.rank)
        ok = True
        for _ in range(rounds):
            for n in bucket_floats:
                x = torch.randn(n, generator=gen).to(self.device)
                want = x.clone()
                dist.all_reduce(want, group=self.pg)
                got = self(x.clone())
                torch.cuda.synchronize(self.device)
                ok = ok and bool(torch.allclose(got, want, rtol=1e-5, atol=1e-6 * self.world))
        ok = ok and self.status() == 0
        t = torch.tensor([1.0 if ok else 0.0], device="cpu" if cpu_pg else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(t.item() > 0.5)

    def close(self):
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            torch.distributed.barrier(group=self.pg)     # peers have stopped storing into this buffer
        except Exception:      # noqa: BLE001 - the group may be gone at interpreter exit
            pass
        self._release()


def probe(process_group, device, bucket_floats, rounds=3, reps=30):
    """SW_ALLREDUCE=auto: build the direct exchange, check it against the process group's all-reduce on this node and time
    both on the step's bucket sizes; every rank runs this at the same point (it is collective).  Returns (DirectAllReduce or
    None, report): the direct form is chosen only if EVERY rank built it, every check agreed (to fp32 summation order: the
    group's reduction order is its own) and its summed time over the buckets is the smaller one (max over ranks)."""
    dist = torch.distributed
    dev = L.indexed_device(device)
    cpu_pg = dist.get_backend(process_group) != "nccl"
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)

    def agree(ok):          # logical AND over the ranks
        t = torch.tensor([1.0 if ok else 0.0], device="cpu" if cpu_pg else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
        return bool(t.item() > 0.5)

    def slowest(x):
        t = torch.tensor([float(x)], device="cpu" if cpu_pg else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=process_group)
        return float(t.item())
    rep = {"buckets_floats": [int(n) for n in bucket_floats], "chosen": "group"}
    try:        # the constructor is failure-symmetric: it raises on every rank or on none
        ar = DirectAllReduce(process_group, dev, max(bucket_floats))
    except Exception as e:      # noqa: BLE001 - any failure means "use the process group"
        rep["reason"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        return None, rep
    if not ar.check(bucket_floats, rounds):      # collective; the verdict is the same on every rank
        rep["reason"] = "the direct exchange disagreed with the group's all-reduce (or a wait timed out)"
        ar.close()
        return None, rep

    def time_calls(fn):
        out = []
        for n in bucket_floats:
            b = torch.zeros(n, device=dev)
            for _ in range(5):
                fn(b)
            dist.barrier(group=process_group)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(b)
            e1.record()
            torch.cuda.synchronize(dev)
            out.append(slowest(e0.elapsed_time(e1) * 1e3 / reps))
        return out
    rep["group_us"] = time_calls(lambda b: dist.all_reduce(b, group=process_group))
    rep["direct_us"] = time_calls(ar)
    if sum(rep["direct_us"]) < sum(rep["group_us"]):
        rep["chosen"] = "direct"
        return ar, rep
    rep["reason"] = "the process group's all-reduce is faster here"
    ar.close()
    return None, rep
