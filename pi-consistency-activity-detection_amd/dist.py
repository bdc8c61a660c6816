"""Data parallelism for the train step: one process per GPU, per-rank minibatches / BN statistics /
dropout draws (the reference has no multi-GPU semantics; SURVEY §8e defines DP(world x bs) as the
mean of `world` independent reference steps), gradients summed with bucketed RCCL all-reduce
(torch.distributed backend "nccl" == RCCL over xGMI) on a side stream while the rest of the
backward still runs, 1/world folded into the fused Adam (pc_adam_step gscale).

The bucket schedule comes from Plan.grad_buckets(): contiguous ranges of the flat gradient buffer
in the order backward finalises them.  The same class runs on CPU tensors with the gloo backend,
which is how the N>1 path is tested without GPUs (tests/test_dist_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world, local_rank).  No-op for world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def check_replicas_agree(flat_params, group=None, what="parameters", rtol=0.0):
    """Insurance for the first step of a data-parallel run: every rank must start from IDENTICAL parameters (the reference has no
    multi-GPU loop; here a rank's state comes from its own seed / checkpoint load, dropin/main_ucf101.py).  All-reduces a 3-number
    checksum of the flat buffer -- sum, sum of squares, and a position-weighted sum that sees permutations -- with MIN and MAX and
    raises on every rank if they differ.  Float64 accumulation of the same fp32 values in the same order is bit-identical across ranks,
    so the default tolerance is zero."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    flat = flat_params.detach().reshape(-1)
    n = flat.numel()
    chk = torch.zeros(3, dtype=torch.float64, device=flat.device)
    CH = 1 << 22                          # float64 temporaries of one chunk at a time (not 3 x 8 B per parameter)
    for a in range(0, n, CH):
        x = flat[a:a + CH].double()
        w = torch.arange(a + 1, a + x.numel() + 1, device=x.device, dtype=torch.float64) / n
        chk += torch.stack([x.sum(), (x * x).sum(), (x * w).sum()])
    dev = chk.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    lo, hi = chk.to(dev).clone(), chk.to(dev).clone()
    # a NaN / Inf checksum compares false against everything: carry "this rank's checksum is finite" through the MIN reduction explicitly
    finite = torch.tensor([1.0 if bool(torch.isfinite(chk).all()) else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(finite, op=dist.ReduceOp.MIN, group=group)
    spread = (hi - lo).abs()
    bad = finite.item() < 1.0 or not bool((spread <= rtol * hi.abs().clamp_min(1e-300)).all())
    if bad:
        raise RuntimeError("data-parallel ranks do not hold the same %s: checksum (sum, sum of squares, weighted sum) spans %s .. %s across the "
                           "%d ranks (this rank: %s) -- seed every rank alike or broadcast rank 0's state before training"
                           % (what, lo.tolist(), hi.tolist(), dist.get_world_size(group), chk.tolist()))
    return True


class GradReducer:
    """Sums ranges of one flat gradient tensor across ranks, bucket by bucket.

    launch(i) is called by the step engine right after the backward op that finalises bucket i was
    enqueued; on GPU the collective is issued on a dedicated stream behind an event, so it overlaps
    with the remaining backward kernels.  wait() joins before Adam."""

    def __init__(self, flat_grad, buckets, group=None, force=False):
        """force: run the bucket schedule and the collectives even in a group of ONE rank (an all-reduce over one rank is the
        identity) -- how the RCCL path is executed on a one-GPU box (tests/test_dp_gpu.py, bench.py PICONS_FORCE_REDUCER=1)."""
        self.g = flat_grad
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        # RCCL reduces device memory in place.  With any other backend (gloo: the single-GPU two-process test of the real
        # engine, tests/test_dp_gpu.py -- RCCL refuses two ranks on one device) device gradients are staged through pinned
        # host memory: D2H on the side stream behind the bucket's event, the host all-reduce and the H2D copy in wait().
        self.host_staged = bool(self.cuda and self.active and dist.get_backend(group) != "nccl")
        self.host = torch.empty(flat_grad.numel(), dtype=flat_grad.dtype).pin_memory() if self.host_staged else None
        self.handles = []
        self.launched = []
        self._work = {}                  # bucket index -> the collective's Work handle (RCCL path; wait_buckets_on)
        # per-step diagnostics (bench.py's `reducer` block): host time spent in wait(), and -- device side -- how long the main stream sat
        # between the last backward kernel and the last collective's completion (the EXPOSED communication time).  Event pairs are kept
        # pending and read by stats() after a synchronize, so the step itself gains no host wait.
        # OFF in a training loop (an event pair per step that nobody reads); bench.py and the tests that read stats() switch it on
        self.diagnostics = False
        self.comm_wait_ms = []
        self._exposed_pending = []
        self._exposed_ms = []
        self._ev_pool = []

    def launch(self, i, streams=()):
        """Start bucket i's all-reduce behind everything enqueued so far on torch's current stream and on `streams`
        (the engine's side lanes: Plan.grad_buckets only marks a bucket ready at joined points, waiting on the
        side lanes as well makes that independent of the plan)."""
        if not self.active:
            return
        _ready, a, b = self.buckets[i]
        view = self.g[a:b]
        self.launched.append(i)
        if self.cuda:
            # the comm stream goes behind torch's current stream and every lane through the library's persistent events
            # (pc_streams_fanin): torch's wait_stream() creates and drops an event per call, and dropping an event that a busy lane
            # has not reached yet stalled the host -- 0.7 ms per step with five buckets
            from . import ops
            ops.streams_fanin(self.comm_stream, streams)
            with torch.cuda.stream(self.comm_stream):
                if self.host_staged:
                    self.host[a:b].copy_(view, non_blocking=True)
                else:
                    self._work[i] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    self.handles.append(self._work[i])
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduced_from(self):
        """Lowest flat-buffer offset such that every bucket launched so far covers [offset, nparams) contiguously: gradients from there on
        are (or will be, once the launched collectives finish) the cross-rank sums.  None when nothing is launched or RCCL is not the backend."""
        if not self.active or not self.launched or self.host_staged or not self.cuda:
            return None
        spans = sorted((self.buckets[i][1], self.buckets[i][2]) for i in self.launched)
        end = self.g.numel()
        lo = end
        for a, b in reversed(spans):
            if b != lo:
                break
            lo = a
        return lo if lo < end else None

    def wait_buckets_on(self, stream):
        """Make `stream` wait (device side, no host block) for the all-reduce of every bucket launched so far -- what an optimiser op
        enqueued on that stream needs before it reads the summed gradients.  (A collective runs on the backend's own stream: Work.wait()
        on a CUDA stream enqueues the dependency on the CURRENT stream and returns.)"""
        with torch.cuda.stream(stream):
            for i in self.launched:
                w = self._work.get(i)
                if w is not None:
                    w.wait()

    def bucket_table(self):
        """[(bytes, ready-op index in the backward list, first float, last float)] per bucket, in launch order."""
        es = self.g.element_size()
        return [dict(bytes=int((b - a) * es), ready_op=int(ready), first=int(a), last=int(b)) for ready, a, b in self.buckets]

    def stats(self, reset=True):
        """Per-step diagnostics since the last reset: call after the device is idle (the pending event pairs are read here).
        comm_wait_ms: host time in wait(); exposed_ms: time the main stream spent waiting for collectives behind its last backward
        kernel (gloo / host-staged paths: the host time of the same span)."""
        import statistics
        for e0, e1 in self._exposed_pending:
            e1.synchronize()
            self._exposed_ms.append(e0.elapsed_time(e1))
            self._ev_pool += [e0, e1]
        self._exposed_pending = []
        med = lambda v: float(statistics.median(v)) if v else None
        out = dict(steps=len(self.comm_wait_ms), comm_wait_ms=med(self.comm_wait_ms), comm_wait_ms_max=max(self.comm_wait_ms) if self.comm_wait_ms else None,
                   exposed_ms=med(self._exposed_ms), exposed_ms_max=max(self._exposed_ms) if self._exposed_ms else None)
        if reset:
            self.comm_wait_ms, self._exposed_ms = [], []
        return out

    def wait(self):
        """Join every launched bucket before the optimiser reads the gradient."""
        import time
        if not self.active or not self.launched:
            self.handles, self.launched, self._work = [], [], {}
            return
        t0 = time.perf_counter()
        ev = None
        if self.cuda and self.diagnostics:
            ev = (self._ev_pool.pop() if self._ev_pool else torch.cuda.Event(enable_timing=True),
                  self._ev_pool.pop() if self._ev_pool else torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream())          # behind the backward's last kernel on the main stream
        if self.host_staged and self.launched:
            self.comm_stream.synchronize()
            for i in self.launched:
                _ready, a, b = self.buckets[i]
                dist.all_reduce(self.host[a:b], op=dist.ReduceOp.SUM, group=self.group)
            with torch.cuda.stream(self.comm_stream):
                for i in self.launched:
                    _ready, a, b = self.buckets[i]
                    self.g[a:b].copy_(self.host[a:b], non_blocking=True)
        for h in self.handles:
            h.wait()
        self.handles = []
        self.launched = []
        self._work = {}
        if self.cuda and self.active:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if ev is not None:
                ev[1].record(torch.cuda.current_stream())      # the main stream gets here once every collective has finished
                self._exposed_pending.append(ev)
                if len(self._exposed_pending) > 4096:           # stats() was never called: give the oldest pair back (both events have long completed)
                    old = self._exposed_pending.pop(0)
                    old[1].synchronize()
                    self._ev_pool += list(old)
        if self.diagnostics:
            dt = (time.perf_counter() - t0) * 1e3
            self.comm_wait_ms.append(dt)
            if not self.cuda:
                self._exposed_ms.append(dt)
            for lst in (self.comm_wait_ms, self._exposed_ms):   # bounded on every path (the gloo path's list used to grow without limit)
                if len(lst) > 65536:
                    del lst[:32768]

    @property
    def gscale(self):
        """Factor the optimiser applies to the summed gradient (mean over ranks)."""
        return 1.0 / self.world


def shard_indices(n_items, rank, world):
    """DistributedSampler-style contiguous shard [rank*n/world, (rank+1)*n/world)."""
    per = n_items // world
    return list(range(rank * per, (rank + 1) * per))


def barrier_max_ms(ms, device=None):
    """MAX over ranks of a local timing (bench.py contract)."""
    if not dist.is_initialized():
        return ms
    t = torch.tensor([ms], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mean_over_ranks(values, device=None):
    """Mean over ranks of a few host scalars (losses that drive the LR scheduler / checkpoint policy).  Identity for world 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                     device=device if (device and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) / dist.get_world_size() for v in t.tolist()]
