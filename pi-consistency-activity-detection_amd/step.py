"""The fused semi-supervised train step on one MI355X: owns the device arena and the flat
parameter / gradient / Adam buffers, stages a minibatch pair, and replays the op lists of plan.py
through libpicons.so.  Mirrors /root/reference/main_ucf101.py train_model_interface :50-150 plus the
zero_grad/backward/Adam loop :171-190, with both forward passes batched and every mask/loss on device.

PyTorch is used for device memory, streams and (in dist.py) the RCCL collective only.
"""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import capi, ops, spec, switches as sw, synthetic
from .plan import Plan


def default_args(**kw):
    """CLI defaults of main_ucf101.py:285-315 that reach the step."""
    a = dict(bv=False, gv=False, n_frames=3, predict_maps=False, lower_thresh=None, upper_thresh=None,
             bv_wt=0.5, gv_wt=0.5, wt_loc=1.0, wt_cls=1.0, wt_cons=1.0, thresh_epoch=11, lr=1e-3, epochs=1)
    a.update(kw)
    return SimpleNamespace(**a)


def exp_rampup(rampup_length):
    """utils/ramp_ups.py:15-24 (host scalar)."""
    def f(epoch):
        if epoch < rampup_length:
            e = float(np.clip(epoch, 0.0, rampup_length))
            ph = 1.0 - e / rampup_length
            return float(np.exp(-5.0 * ph * ph))
        return 1.0
    return f


class StepEngine:
    def __init__(self, args, bs=8, hw=224, num_classes=24, device="cuda:0", jhmdb=False, state=None, seed=47, lanes=None, exp=None):
        if not torch.cuda.is_available():
            raise RuntimeError("StepEngine needs a GPU: the hot path is HIP-only (no CPU fallback)")
        capi.lib()
        self.args = args
        self.exp = dict(exp or {})          # experiment switches passed explicitly (switches.py)
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        self.bs = bs
        self.C = num_classes
        self.hw = hw
        self.jhmdb = jhmdb
        if lanes is None:
            # measured best on MI355X: lane 1 for the second Inception branch, lane 2 for the decoder's skip convs,
            # lane 3 for the weight gradients
            lanes = int(sw.get("PICONS_LANES", "4"))
        p = Plan(num_classes, hw, n=bs, groups=2, training=True, jhmdb=jhmdb, lanes=lanes, early_adam=True, exp=self.exp)
        self.side = [torch.cuda.Stream(device=self.dev) for _ in range(lanes - 1)]   # lanes 1.. of the op lists
        # ROCm binds a stream to one of its (by default four) hardware queues when the stream is first used, and two lanes on one
        # hardware queue serialise: use the lanes, in order, before anything else in this process creates work on another stream
        # (the collectives of a process group, the loss read-back stream) -- GPU_MAX_HW_QUEUES != 4 is 3-25 % slower
        if sw.exp("PICONS_BIND_LANES", "1", self.exp) != "0":
            tick = torch.zeros(64, device=self.dev)
            order = [torch.cuda.current_stream(self.dev)] + self.side
            perm = sw.exp("PICONS_BIND_ORDER", "", self.exp)             # diagnostic: e.g. "0,3,2,1"
            if perm:
                order = [order[int(q)] for q in perm.split(",") if int(q) < len(order)]
            for st in order:
                with torch.cuda.stream(st):
                    tick.add_(1.0)
            torch.cuda.synchronize(self.dev)
        # PICONS_PRIO=1: lane 0 (the dependency chain) on a high-priority stream of its own, so its workgroups win the CU slots and
        # the side lanes fill what is left
        self.main = torch.cuda.Stream(device=self.dev, priority=-1) if (lanes > 1 and sw.exp("PICONS_PRIO", "0", self.exp) != "0") else None
        p.build_forward()
        p.build_loss(args)
        p.build_backward()
        p.build_adam()
        self.plan = p
        f32 = dict(device=self.dev, dtype=torch.float32)
        self.arena = torch.empty(p.arena_bytes + 256, device=self.dev, dtype=torch.uint8)
        self.P = torch.zeros(p.nparams, **f32)
        self.G = torch.zeros(p.nparams, **f32)
        self.M = torch.zeros(p.nparams, **f32)
        self.V = torch.zeros(p.nparams, **f32)
        self.R = torch.zeros(p.nrunning, **f32)
        base_a = (self.arena.data_ptr() + 255) // 256 * 256
        self._a0 = base_a - self.arena.data_ptr()
        p.upload_consts(self.aview)
        self.bases = dict(A=base_a, P=self.P.data_ptr(), G=self.G.data_ptr(), M=self.M.data_ptr(), V=self.V.data_ptr(), R=self.R.data_ptr())
        self.ops = p.resolve(self.bases)
        self.step_count = 0
        self.kind_ms, self.kind_count = 0.0, 0
        # the loss scalars are final before the backward starts: their D2H goes out on a stream of its own behind the loss list and
        # the step waits for THAT copy only, so the host is enqueueing the next step while the backward and Adam still run
        self.scal_host = torch.empty(20, dtype=torch.float32).pin_memory()
        self.scal_stream = torch.cuda.Stream(device=self.dev)
        # both events are created ONCE and re-recorded every step: dropping a torch event that a busy lane has not reached yet stalls the
        # host (what dist.GradReducer.launch ran into: 0.7 ms per step)
        self._loss_done = torch.cuda.Event()
        self._scal_event = torch.cuda.Event()
        self.scal_event = None                                # set (to _scal_event) once a step has enqueued the copy
        self.load_state(state if state is not None else synthetic.init_state(seed, num_classes))

    # ------------------------------------------------------------------ views
    def aview(self, ref, nfloats, dtype=torch.float32):
        o = self._a0 + ref[1]
        return self.arena[o:o + 4 * nfloats].view(dtype)

    def param(self, name):
        shp = self.plan.pshape[name]
        o = self.plan.poff[name]
        return self.P[o:o + int(np.prod(shp))].view(shp)

    def grad(self, name):
        shp = self.plan.pshape[name]
        o = self.plan.poff[name]
        return self.G[o:o + int(np.prod(shp))].view(shp)

    def load_state(self, state):
        """state: reference-layout state_dict (numpy or torch values, SURVEY §5 key names)."""
        T = lambda v: (v.detach() if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))).to(self.dev)     # host or device values
        for k in self.plan.pshape:
            self.param(k).copy_(T(state[k]))
        for k, o in self.plan.roff.items():
            v = T(state[k])
            self.R[o:o + v.numel()].copy_(v)
        self.nbt = {k: int(v.item() if torch.is_tensor(v) else np.asarray(v)) for k, v in state.items() if k.endswith("num_batches_tracked")}

    def state_dict(self):
        sd = {}
        for k in spec.state_dict_keys(self.C):
            if k in self.plan.pshape:
                sd[k] = self.param(k).detach().clone()
            elif k in self.plan.roff:
                o = self.plan.roff[k]
                n = self.plan.pshape[k.rsplit(".bn.", 1)[0] + ".bn.weight"][0]
                sd[k] = self.R[o:o + n].detach().clone()
            else:
                sd[k] = torch.tensor(self.nbt.get(k, 0), dtype=torch.long)
        return sd

    # ------------------------------------------------------------------ staging
    def stage(self, label_mb, unlabel_mb, perm, drops):
        """Concat labeled+unlabeled, apply the shuffle (main_ucf101.py:65-79) and copy to the arena.
        drops: four (bs,C) scale arrays [d832 pass0, d128 pass0, d832 pass1, d128 pass1]."""
        p = self.plan
        self._restore_input_ops()
        T = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
        cat = lambda k: torch.cat([T(label_mb[k]), T(unlabel_mb[k])], dim=0)
        perm = torch.as_tensor(np.asarray(perm)).long()
        n = self.bs
        per = 3 * spec.FRAMES * self.hw * self.hw
        from . import inputpipe
        perm_d = None

        def shuffled(k):          # device-resident samples (inputpipe.get_item) are gathered with a device index: a host index tensor is a blocking upload
            nonlocal perm_d
            c = cat(k)
            if not c.is_cuda:
                return c[perm]
            if perm_d is None:
                perm_d = inputpipe.pinned_upload(perm.numpy(), c.device)
            return c[perm_d]
        data = shuffled("data").to(self.dev, torch.float32, non_blocking=True)
        aug = shuffled("aug_data").to(self.dev, torch.float32, non_blocking=True)
        seg = shuffled("loc_msk").to(self.dev, torch.float32, non_blocking=True)
        act_h = cat("action")[perm].reshape(-1).to(torch.float32)
        if self.jhmdb:        # main_jhmdb.py:68-70
            lab = torch.cat([torch.ones(len(label_mb["action"])), torch.zeros(len(unlabel_mb["action"]))])
        else:
            lab = cat("label_vid")
        lab_h = lab[perm].to(torch.int32)
        self.labels_host, self.action_host = lab_h.cpu(), act_h.cpu()      # kept from the host inputs (no read-back)
        # host-made scalars go up through page-locked staging (inputpipe.pinned_upload): a pageable H2D copy, however small, blocks the host
        # until everything queued on the stream has run -- i.e. until the PREVIOUS step is over, and the host stops running ahead of the GPU
        up = lambda t: t.to(self.dev) if t.is_cuda else inputpipe.pinned_upload(t.numpy(), self.dev)
        act, lab = up(act_h), up(lab_h)
        self.aview(p.in_data, n * per).copy_(data.reshape(-1))
        self.aview(p.in_aug, n * per).copy_(aug.reshape(-1))
        self.aview(p.in_seg, n * per // 3).copy_(seg.reshape(-1))
        self.aview(p.in_cls, 2 * n).copy_(torch.cat([act, act]))
        self.aview(p.in_labeled, 2 * n, torch.int32).copy_(torch.cat([lab, lab]))
        d = [x.to(self.dev, torch.float32) if torch.is_tensor(x) and x.is_cuda else up(torch.as_tensor(np.asarray(x.cpu() if torch.is_tensor(x) else x), dtype=torch.float32))
             for x in drops]
        self.aview(p.in_drop832, 2 * n * spec.TRUNK_OUT_CH).copy_(torch.cat([d[0], d[2]]).reshape(-1))
        self.aview(p.in_drop128, 2 * n * 128).copy_(torch.cat([d[1], d[3]]).reshape(-1))

    # ------------------------------------------------------------------ the reference's minibatch contract, one step ahead
    def host_stager(self):
        """-> HostDictStager bound to this engine (created once)."""
        if getattr(self, "_stager", None) is None:
            self._stager = HostDictStager(self)
        return self._stager

    def sample_stager(self):
        """-> SampleStager bound to this engine (created once): device-made samples written straight into the next step's minibatch."""
        if getattr(self, "_sstager", None) is None:
            self._sstager = SampleStager(self)
        return self._sstager

    def readers_of(self, ref, nfloats, skip=()):
        """(base address, [(list, op index, pointer slot, byte offset)]) of every op pointer that points into the arena buffer `ref`: what a
        stager re-points at its own double buffer instead of copying into the arena."""
        base = int(self.bases[ref[0]] + ref[1])
        out = []
        # pointers that live in HOST job tables (OP_WGRAD_MULTI, OP_TRANSPOSE_MULTI, OP_SPLIT_PLANES_MULTI) are not re-pointable through the
        # op arrays: a reader grouped into such a table would keep reading the arena buffer a stager never writes -- refuse instead
        for tab in self.ops.get("_tjobs", []):
            for f in tab.dtype.names:
                if tab.dtype[f] == np.uint64 and bool(((tab[f] >= base) & (tab[f] < base + 4 * nfloats)).any()):
                    raise RuntimeError("an op's host job table points into the buffer at arena offset %d: re-pointing its readers would miss it" % ref[1])
        for name, arr in self.ops.items():
            if name.startswith("_"):
                continue
            for idx in range(len(arr)):
                if (name, idx) in skip:
                    continue
                pp = arr[idx]["p"]
                for q in range(len(pp)):
                    if base <= int(pp[q]) < base + 4 * nfloats:
                        out.append((name, idx, q, int(pp[q]) - base))
        return base, out

    def _restore_input_ops(self):
        """stage() feeds the fp32 arena staging buffers: undo a HostDictStager.commit() that pointed the clip conversion at its own."""
        for idx, (flag, ptr) in getattr(self, "_to_ndhwc_orig", {}).items():
            self.ops["fwd"][idx]["i"][0] = flag
            self.ops["fwd"][idx]["i"][1] = self.bs
            self.ops["fwd"][idx]["p"][0] = ptr
        for base, readers in getattr(self, "_patched", {}).values():
            for name, idx, q, offb in readers:
                self.ops[name][idx]["p"][q] = base + offb

    # ------------------------------------------------------------------ execution
    def reset_workspaces(self):
        """Zero the workspaces whose counters every launch must find (and leaves) at zero -- the tail-split arrival counters of the bf16-split
        conv launches (csrc/conv_x6.hip).  Done once when the arena is created; again here after a step that raised, because a launch that
        failed or never ran can leave a counter mid-count, after which that tile's epilogue would never run again."""
        torch.cuda.synchronize(self.dev)
        for ref, n in self.plan.zero_once:
            self.aview(ref, n).zero_()
        torch.cuda.synchronize(self.dev)

    def forward_backward(self, epoch, wt_ramp, reducer=None, timed_kind=None, timed_on_lanes=False):
        try:
            return self._forward_backward(epoch, wt_ramp, reducer, timed_kind, timed_on_lanes)
        except Exception:
            try:
                self.reset_workspaces()
            except Exception:          # the device itself is gone: the first error is the one to report
                pass
            raise

    def _forward_backward(self, epoch, wt_ramp, reducer=None, timed_kind=None, timed_on_lanes=False):
        """prep -> forward (both passes batched) -> losses -> backward.  With a dist.GradReducer the
        backward list is replayed in segments and each gradient bucket's all-reduce is launched as soon
        as the ops that finalise it are enqueued.  timed_kind: accumulate hipEvent time of that op kind
        (bench.py roofline leg) into self.kind_ms / self.kind_count."""
        p, o = self.plan, self.ops
        o["fwd"][p.op_cmask]["i"][3] = 0 if epoch < self.args.thresh_epoch else 1
        o["loss"][p.op_loss]["f"][6] = wt_ramp

        def run(arr):
            if timed_kind is None or len(arr) == 0:
                ops.run_ops(arr, side=self.side)
            else:
                # a timed step is replayed on ONE stream (FORK / JOIN are no-ops then): kernels that overlap on two lanes stretch
                # each other's event-timed duration, and the roofline leg wants every kernel's own duration.  timed_on_lanes: the event
                # pairs ride in the dispatches of the ordinary four-lane step instead (durations as stretched by the co-running lanes):
                # what bench.py's headline leg uses, so that its timed steps cost what every other step costs
                ops.run_ops_timed(arr, timed_kind, side=self.side if timed_on_lanes else None, defer=True)    # read after the step's own sync
        run(o["prep"])
        run(o["prep_late"])        # side lanes only; joined inside the forward list (plan.late_prep)
        run(o["fwd"])
        run(o["loss"])
        self._send_scalars()
        if reducer is None or not reducer.active:
            run(o["bwd"])
        else:
            k = p.op_adam_early if getattr(self, "_early_dp", False) else None
            done = 0

            def run_to(end):
                """bwd[done:end), stopping at the early Adam op: it may only read gradients whose all-reduce has been launched, and its lane
                must first wait for those collectives (device-side events of the comm stream)."""
                nonlocal done
                if k is not None and done <= k < end:
                    if k > done:
                        run(o["bwd"][done:k])
                    lo = reducer.reduced_from()
                    e = o["bwd"][k]
                    if lo is not None and lo < p.nparams:
                        lo = max(lo, p.adam_split)
                        for q, base in enumerate(("P", "G", "M", "V")):
                            e["p"][q] = self.bases[base] + 4 * lo
                        e["l"][0] = p.nparams - lo
                        e["f"][4] = reducer.gscale
                        self._early_lo = lo
                        lane = int(e["lane"])
                        reducer.wait_buckets_on(self.side[lane - 1] if lane > 0 else torch.cuda.current_stream(self.dev))
                    else:
                        e["l"][0] = 0
                        self._early_lo = None
                    done = k
                if end > done:
                    run(o["bwd"][done:end])
                    done = end
            for i, (ready, _a, _b) in enumerate(reducer.buckets):
                if ready > done:
                    run_to(ready)
                reducer.launch(i, self.side)
            run_to(len(o["bwd"]))

    def arm_early_adam(self, lr, on):
        """on: the backward list's Adam op (plan.early_adam) updates every parameter but the stem's beside the stem's weight gradient,
        with THIS step's count and lr; adam() then updates the stem alone.  Off (a reducer is active: the gradients are only final
        after the all-reduce; or a caller that runs the lists by hand): the op is a no-op and adam() covers everything."""
        k = self.plan.op_adam_early
        self._early_armed = bool(on) and k is not None
        self._early_dp, self._early_lo = False, None
        if k is None:
            return
        e = self.ops["bwd"][k]
        n0 = self.plan.adam_split
        for q, base in enumerate(("P", "G", "M", "V")):
            e["p"][q] = self.bases[base] + 4 * n0
        e["l"][0] = self.plan.nparams - n0 if self._early_armed else 0
        e["i"][0] = self.step_count + 1
        e["f"][0] = lr
        e["f"][4] = 1.0

    def arm_early_adam_dp(self, lr, reducer):
        """Data parallelism (RCCL in place): the early Adam op covers the parameters whose gradient buckets have been LAUNCHED by the time the
        backward reaches it (decoder, capsule head and most of the trunk: the buckets leave mid-backward), behind those collectives on its
        own lane, with 1/world folded in; adam() then covers the rest -- the trunk's last bucket, reduced at the end of the backward.
        forward_backward() fills in the range when it gets there."""
        self.arm_early_adam(lr, on=False)
        k = self.plan.op_adam_early
        ok = (k is not None and reducer is not None and reducer.active and reducer.cuda and not reducer.host_staged
              and sw.get("PICONS_EARLY_ADAM_DP", "1") != "0")
        self._early_dp = bool(ok)

    def adam(self, lr, gscale=1.0):
        self.step_count += 1
        a = self.ops["adam"][0]
        a["i"][0] = self.step_count
        a["f"][0] = lr
        a["f"][4] = gscale
        if getattr(self, "_early_armed", False):
            a["l"][0] = self.plan.adam_split
        elif getattr(self, "_early_dp", False) and getattr(self, "_early_lo", None) is not None:
            a["l"][0] = self._early_lo            # the early op took [_early_lo, nparams) behind the launched buckets
        else:
            a["l"][0] = self.plan.nparams
        ops.run_ops(self.ops["adam"])
        self.last_adam_split = int(a["l"][0])          # diagnostics / tests: the final Adam covered [0, last_adam_split), the early op the rest
        self._early_armed = False
        self._early_dp, self._early_lo = False, None
        if self.plan.op_adam_early is not None:
            self.ops["bwd"][self.plan.op_adam_early]["l"][0] = 0       # a backward replayed by hand must not step the optimiser
        for k in self.nbt:
            self.nbt[k] += 2           # two forward passes per step

    def _send_scalars(self):
        self._loss_done.record(torch.cuda.current_stream())
        self.scal_stream.wait_event(self._loss_done)
        with torch.cuda.stream(self.scal_stream):
            self.scal_host.copy_(self.aview(self.plan.scalars, 20), non_blocking=True)   # spread_out sits at scalars + 16 (plan.build_loss)
            self._scal_event.record(self.scal_stream)
        self.scal_event = self._scal_event

    def read_scalars(self):
        """One packed D2H for the step's loss scalars (replaces the reference's five .item() syncs); the copy was enqueued behind
        the loss list by forward_backward, this waits for it (not for the backward / Adam)."""
        if self.scal_event is None:                           # the loss list was replayed by hand (tools/bench_loss.py): blocking copy
            s = self.aview(self.plan.scalars, 20).cpu()
        else:
            self.scal_event.synchronize()
            s = self.scal_host.clone()
        loc, cons, cls = float(s[0]), float(s[1]), float(s[16])
        a = self.args
        return dict(total=a.wt_loc * loc + a.wt_cls * cls + a.wt_cons * cons, loc=loc, cls=cls, cons=cons,
                    bce=float(s[2]), dice=float(s[3]), l2=float(s[4]), lvar=float(s[5]), lgrad=float(s[6]))

    def synchronize(self):
        """Wait until everything this engine has enqueued (every lane, the loss read-back stream) has finished.  train_step /
        run_staged return once the LOSS scalars are on the host -- the backward and Adam may still be running; call this before
        reading G / P / the running statistics through raw pointers or from another stream, or before stopping a clock."""
        torch.cuda.current_stream(self.dev).synchronize()
        for st in self.side:
            st.synchronize()
        if self.main is not None:
            self.main.synchronize()
        self.scal_stream.synchronize()

    def outputs(self):
        """(output (bs,1,8,H,W), flip_op, predicted_action (bs,C)) of the last forward, shuffled order."""
        p = self.plan
        per = spec.FRAMES * self.hw * self.hw
        out = self.aview(p.out.ref, 2 * self.bs * per).view(2 * self.bs, 1, spec.FRAMES, self.hw, self.hw)
        pred = self.aview(p.pred, 2 * self.bs * self.C).view(2 * self.bs, self.C)
        return out[:self.bs], out[self.bs:], pred[:self.bs]

    def make_reducer(self, group=None, target_floats=3_000_000, force=False, check=True):
        """check: refuse to train if the ranks do not hold identical parameters (dist.check_replicas_agree: one 3-number all-reduce)."""
        from . import dist as pdist
        joined = sw.exp("PICONS_BUCKETS_JOINED", "0", self.exp) != "0"      # A/B switch: buckets only where lane 0 has joined their lane
        if check:
            pdist.check_replicas_agree(self.P, group)
        return pdist.GradReducer(self.G, self.plan.grad_buckets(target_floats, joined=joined), group, force=force)

    def collect_timing(self):
        """Read the event pairs the timed replays left pending (pc_run_ops_timed_collect) into kind_ms / kind_count."""
        ms, cnt = ops.timed_collect()
        self.kind_ms += ms
        self.kind_count += cnt

    def run_staged(self, epoch, wt_ramp, lr=None, reducer=None, timed_kind=None, collect=True, timed_on_lanes=False):
        if self.main is not None and reducer is None:
            self.main.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.main):
                out = self._run_staged(epoch, wt_ramp, lr, reducer, timed_kind, collect, timed_on_lanes)
            torch.cuda.current_stream().wait_stream(self.main)
            return out
        return self._run_staged(epoch, wt_ramp, lr, reducer, timed_kind, collect, timed_on_lanes)

    def _run_staged(self, epoch, wt_ramp, lr=None, reducer=None, timed_kind=None, collect=True, timed_on_lanes=False):
        """One full step on the minibatch already staged in HBM: fwd x2 + losses + bwd (+ all-reduce) +
        Adam + the packed loss read-back.  RETURNS WITH THE BACKWARD AND ADAM STILL IN FLIGHT: the only host wait is for the loss
        scalars' copy, which is final before the backward starts.  Stream-ordered consumers (the next step, torch ops on the current
        stream) need nothing; anything else calls synchronize() first.  collect=False leaves the timing events of a timed step pending (the caller
        reads them with collect_timing() later, e.g. after its timed region: reading 212 events costs ~0.4 ms of host time)."""
        product = timed_kind is None or timed_on_lanes          # the ordinary schedule (a single-stream timed replay is not)
        if reducer is not None and reducer.active and product:
            self.arm_early_adam_dp(self.args.lr if lr is None else lr, reducer)
        else:
            self.arm_early_adam(self.args.lr if lr is None else lr, on=(reducer is None or not reducer.active) and product)
        self.forward_backward(epoch, wt_ramp, reducer, timed_kind, timed_on_lanes)
        gscale = 1.0
        if reducer is not None:
            reducer.wait()
            gscale = reducer.gscale
        self.adam(self.args.lr if lr is None else lr, gscale)
        out = self.read_scalars()            # the step's one host wait: for the loss scalars' copy, not for the GPU to drain
        if timed_kind is not None and collect:
            self.collect_timing()
        return out

    def train_step(self, label_mb, unlabel_mb, epoch, wt_ramp, perm, drops, lr=None, reducer=None):
        self.stage(label_mb, unlabel_mb, perm, drops)
        return self.run_staged(epoch, wt_ramp, lr, reducer)


class HostDictStager:
    """The reference's own minibatch contract taken one step ahead.  main_ucf101.py:52-79 gets two dicts of float64 HOST tensors from its
    DataLoaders (datasets/ucf_dataloader.py:179-191: data / aug_data (n,3,8,H,W), loc_msk (n,1,8,H,W); 180 MB per bs-8 step), casts,
    concatenates, shuffles and uploads them inside the step.  Here, while step i runs on the GPU:
      prepare(i+1)  the host gathers the shuffled samples into a page-locked double buffer (one memcpy per sample: cat + randperm shuffle
                    are the order of those copies) and enqueues ONE async H2D per tensor on a copy stream;
      commit(i+1)   at the head of step i+1 the main stream waits for that upload and the step's first kernels read the float64 staging
                    directly: pc_ncdhw_to_ndhwc converts f64 -> f32 while it re-lays the clip out (its op is re-pointed), the mask is cast
                    into the arena, the per-sample scalars and Dropout3d draws follow in one small packed upload.
    The host never waits for the GPU here; its only wait stays StepEngine.read_scalars()."""

    def __init__(self, eng, dtype=torch.float64, host=True):
        self.eng = eng
        n, hw, T = eng.bs, eng.hw, spec.FRAMES
        self.dtype = dtype
        self.shapes = dict(data=(n, 3, T, hw, hw), aug_data=(n, 3, T, hw, hw), loc_msk=(n, 1, T, hw, hw))
        self.pin = [{k: torch.empty(shp, dtype=dtype).pin_memory() for k, shp in self.shapes.items()} for _ in range(2)] if host else None
        self.dev = [{k: torch.empty(shp, dtype=dtype, device=eng.dev) for k, shp in self.shapes.items()} for _ in range(2)]
        self.nsmall = 2 * n + 2 * n * spec.TRUNK_OUT_CH + 2 * n * 128           # action, labeled flag, the four Dropout3d draws
        self.pin_small = [torch.empty(self.nsmall, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.dev_small = [torch.empty(self.nsmall, dtype=torch.float32, device=eng.dev) for _ in range(2)]
        # A normal-priority stream.  (A high-priority one -- a hardware queue of its own instead of sharing a lane's, DESIGN.md 5 -- was measured
        # and is SLOWER: the staged step 19.92 - 20.04 ms against 19.33 - 19.43, tools/gpu/r04_o.sh: its kernels then pre-empt the lanes' blocks.
        # PICONS_STAGE_STREAM_PRIO=-1 selects it.)
        self.copy_stream = torch.cuda.Stream(device=eng.dev, priority=int(sw.exp("PICONS_STAGE_STREAM_PRIO", "0", eng.exp)))
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]
        self.used = [False, False]
        self.host = [None, None]
        # the 180 MB gather is 24 sample-sized memcpys: one thread moves ~5 GB/s (33 ms per step, more than the step itself); tensor.copy_
        # releases the GIL, so a small pool brings it to a few ms
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=int(sw.get("PICONS_STAGE_THREADS", "8"))) if host else None
        if not hasattr(eng, "_to_ndhwc_orig"):
            eng._to_ndhwc_orig = {idx: (int(eng.ops["fwd"][idx]["i"][0]), int(eng.ops["fwd"][idx]["p"][0])) for idx in eng.plan.op_to_ndhwc}

    def prepare(self, slot, label_mb, unlabel_mb, perm, drops):
        """Host side of one minibatch: gather into the pinned slot in shuffled order, enqueue the uploads.  Returns at once."""
        eng, n = self.eng, self.eng.bs
        nl = len(label_mb["action"])
        T_ = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))
        perm = np.asarray(perm)
        if self.used[slot]:
            self.consumed[slot].synchronize()      # the step that read this slot's device copy was enqueued two steps ago: long done
        jobs = []
        for k in self.shapes:
            lab, unl = T_(label_mb[k]), T_(unlabel_mb[k])
            dst = self.pin[slot][k]
            for j, src in enumerate(perm):         # torch.cat + the randperm shuffle of main_ucf101.py:65-79 as the ORDER of these copies
                jobs.append(self.pool.submit(dst[j].copy_, lab[src] if src < nl else unl[src - nl]))
        for f in jobs:
            f.result()
        act = torch.cat([T_(label_mb["action"]).reshape(-1), T_(unlabel_mb["action"]).reshape(-1)]).float()[perm]
        if eng.jhmdb:                              # main_jhmdb.py:68-70
            lab_flag = torch.cat([torch.ones(nl), torch.zeros(n - nl)])[perm]
        else:
            lab_flag = torch.cat([T_(label_mb["label_vid"]), T_(unlabel_mb["label_vid"])]).float()[perm]
        self._pack_small(slot, act, lab_flag, drops)
        with torch.cuda.stream(self.copy_stream):
            for k in self.shapes:
                self.dev[slot][k].copy_(self.pin[slot][k], non_blocking=True)
            self.dev_small[slot].copy_(self.pin_small[slot], non_blocking=True)
            self.ready[slot].record(self.copy_stream)
        self.used[slot] = True

    def _pack_small(self, slot, act, lab_flag, drops):
        """Per-sample scalars and the four Dropout3d draws of a step into the slot's page-locked vector (uploaded in one copy)."""
        n = self.eng.bs
        T_ = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))
        ps = self.pin_small[slot]
        ps[:n] = act
        ps[n:2 * n] = lab_flag
        o = 2 * n
        for d, c in zip((drops[0], drops[2], drops[1], drops[3]), (spec.TRUNK_OUT_CH, spec.TRUNK_OUT_CH, 128, 128)):
            ps[o:o + n * c] = T_(d).reshape(-1).float()
            o += n * c
        self.host[slot] = (lab_flag.to(torch.int32), act.clone())

    def commit(self, slot):
        """Head of the step (main stream): wait for the slot's upload, point the clip conversion at its float64 staging, cast the mask
        and scatter the small inputs into the arena."""
        eng, p, n = self.eng, self.eng.plan, self.eng.bs
        main = torch.cuda.current_stream(eng.dev)
        main.wait_event(self.ready[slot])
        eng._restore_input_ops()                      # whatever another stager left re-pointed
        fwd = eng.ops["fwd"]
        for g, idx in enumerate(p.op_to_ndhwc):
            fwd[idx]["i"][0] = 1 if self.dtype == torch.float64 else 0        # src_is_f64
            fwd[idx]["p"][0] = self.dev[slot]["data" if g == 0 else "aug_data"].data_ptr()
        self._commit_small(slot)

    def _commit_small(self, slot):
        eng, p, n = self.eng, self.eng.plan, self.eng.bs
        per = spec.FRAMES * eng.hw * eng.hw
        eng.aview(p.in_seg, n * per).copy_(self.dev[slot]["loc_msk"].reshape(-1))         # (f64 -> f32 on the device)
        ds = self.dev_small[slot]
        eng.aview(p.in_cls, 2 * n).copy_(torch.cat([ds[:n], ds[:n]]))
        eng.aview(p.in_labeled, 2 * n, torch.int32).copy_(torch.cat([ds[n:2 * n], ds[n:2 * n]]).to(torch.int32))
        o, c8 = 2 * n, n * spec.TRUNK_OUT_CH
        eng.aview(p.in_drop832, 2 * c8).copy_(ds[o:o + 2 * c8])
        eng.aview(p.in_drop128, 2 * n * 128).copy_(ds[o + 2 * c8:o + 2 * c8 + 2 * n * 128])
        eng.labels_host, eng.action_host = self.host[slot]

    def release(self, slot):
        """Behind the step that consumed the slot (its last reader is the loss list / the backward's dropout ops reading the arena -- the
        float64 staging itself is only read by the first two kernels)."""
        self.consumed[slot].record(torch.cuda.current_stream(self.eng.dev))


class SampleStager(HostDictStager):
    """The DEVICE input pipeline taken one step ahead: the per-sample work of the loaders' __getitem__ (inputpipe.get_item: frame choice, crop,
    /255, flip, box mask from the decoded uint8 frames) writes every sample straight into ITS PLACE of the next step's minibatch -- position j
    of a double buffer, where j is where torch.cat + the randperm shuffle of main_ucf101.py:65-79 put the sample -- on a side stream while the
    current step runs, and IN THE LAYOUT THE FIRST CONV READS ([2 bs][T][H][W][4]: clip and flipped clip, RGB padded to one 16-byte piece,
    pc_clip_from_u8_ndhwc4).  No stack, no cat, no gather, no copy into the arena and no NCDHW -> NDHWC conversion: commit() re-points the
    readers of the clip (the stem's conv and its weight gradient) at the buffer and switches the conversion ops off; the only per-step copies left
    are the mask (6 MB) and the small vectors."""

    def __init__(self, eng):
        super().__init__(eng, dtype=torch.float32, host=False)
        n, hw, T = eng.bs, eng.hw, spec.FRAMES
        p = eng.plan
        # the stream the samples are made on: lane 1's (prepare()); PICONS_STAGE_LANE=0: the copy stream of the base class (rounds 4 - 5)
        lane = int(sw.get("PICONS_STAGE_LANE", "1"))
        self.stage_stream = eng.side[lane - 1] if 1 <= lane <= len(eng.side) else self.copy_stream
        self.x = [torch.empty(2 * n, T, hw, hw, 4, dtype=torch.float32, device=eng.dev) for _ in range(2)]
        for d in self.dev:                             # the planar staging of the base class is not used here
            d.pop("data"); d.pop("aug_data")
        # the small inputs in the slot exactly as the ops read them -- class ids and labeled flags once per pass, the Dropout3d draws of both
        # passes -- so that they too are re-pointed instead of copied: [cls 2n | labeled 2n (int32) | drop832 2n x 832 | drop128 2n x 128]
        self.o_cls, self.o_lab, self.o_d832 = 0, 2 * n, 4 * n
        self.o_d128 = self.o_d832 + 2 * n * spec.TRUNK_OUT_CH
        self.nsmall = self.o_d128 + 2 * n * 128
        self.pin_small = [torch.empty(self.nsmall, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.dev_small = [torch.empty(self.nsmall, dtype=torch.float32, device=eng.dev) for _ in range(2)]
        if not hasattr(eng, "_patched"):               # every pointer of the op lists into the clip tensor / the arena's input buffers
            conv_ops = tuple(("fwd", idx) for idx in p.op_to_ndhwc)
            per = T * hw * hw
            eng._patched = dict(img=eng.readers_of(p.img.ref, 2 * n * per * 4, skip=conv_ops), seg=eng.readers_of(p.in_seg, n * per),
                                cls=eng.readers_of(p.in_cls, 2 * n), lab=eng.readers_of(p.in_labeled, 2 * n),
                                d832=eng.readers_of(p.in_drop832, 2 * n * spec.TRUNK_OUT_CH), d128=eng.readers_of(p.in_drop128, 2 * n * 128))
            if not eng._patched["img"][1] or not eng._patched["seg"][1]:
                raise RuntimeError("no op reads the clip tensor / the mask: the plan changed under SampleStager")
            # the readers this stager is built for: the stem's conv in the forward and the stem's weight gradient in the backward (training
            # plans); anything else reading the clip would be re-pointed too, but if one of THESE is missing it was folded into something
            # readers_of() cannot see
            kinds = {(name, int(eng.ops[name][idx]["kind"])) for name, idx, _q, _o in eng._patched["img"][1]}
            if ("fwd", capi.OP_CONV) not in kinds or (p.training and ("bwd", capi.OP_WGRAD) not in kinds):
                raise RuntimeError("SampleStager: the stem's conv / weight gradient are not among the clip's readers (%s)" % sorted(kinds))

    def _pack_small(self, slot, act, lab_flag, drops):
        n = self.eng.bs
        T_ = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))
        ps = self.pin_small[slot]
        ps[self.o_cls:self.o_cls + n] = act; ps[self.o_cls + n:self.o_cls + 2 * n] = act
        li = ps.view(torch.int32)
        li[self.o_lab:self.o_lab + n] = lab_flag.to(torch.int32); li[self.o_lab + n:self.o_lab + 2 * n] = lab_flag.to(torch.int32)
        o = self.o_d832
        for d, c in zip((drops[0], drops[2], drops[1], drops[3]), (spec.TRUNK_OUT_CH, spec.TRUNK_OUT_CH, 128, 128)):
            ps[o:o + n * c] = T_(d).reshape(-1).float()
            o += n * c
        self.host[slot] = (lab_flag.to(torch.int32), act.clone())

    def prepare(self, slot, make_sample, nl, perm, drops):
        """make_sample(i, out) -> the sample dict of dataset position i (0 .. nl-1 labeled, then unlabeled) written into
        out = (data, aug_data, loc_msk) views (inputpipe.get_item(..., out=out, ndhwc4=True)); called in dataset order, so the loader's own
        random draws come in the reference's order whatever the shuffle."""
        eng, n = self.eng, self.eng.bs
        perm = np.asarray(perm)
        where = np.empty(n, np.int64)
        where[perm] = np.arange(n)                     # sample i lands at position where[i]
        x, m = self.x[slot], self.dev[slot]["loc_msk"]
        # Round 6: the samples are made on LANE 1's stream, not on a stream of their own.  A fifth stream shares a hardware queue with one of the four
        # lanes -- it got the skip lane's, whose last op of a step is the early Adam right in front of the stem's backward -- so the
        # uploads and pc_clip_from_u8 launches enqueued behind the step sat in that queue until then and ran beside / behind the stem's weight
        # gradient, the LAST kernel of the step: the next step's stem conv waited 0.5 - 0.75 ms for them (rocprofv3 kernel trace,
        # profiles/r06_staging_lane.txt).  Lane 1 (the second Inception branch) has drained a millisecond earlier; its queue is the one to wait in.
        st_ = self.stage_stream
        with torch.cuda.stream(st_):
            if self.used[slot]:
                # the slot's last readers (the previous-but-one minibatch's step: stem conv / weight gradient, loss list): waited for on the device
                st_.wait_event(self.consumed[slot])
            samples = [make_sample(i, (x[where[i]], x[n + where[i]], m[where[i]])) for i in range(n)]
            act = torch.tensor([float(torch.as_tensor(smp["action"]).reshape(-1)[0]) for smp in samples])[perm]
            if eng.jhmdb:                              # main_jhmdb.py:68-70
                lab_flag = torch.cat([torch.ones(nl), torch.zeros(n - nl)])[perm]
            else:
                lab_flag = torch.tensor([float(smp["label_vid"]) for smp in samples])[perm]
            self._pack_small(slot, act, lab_flag, drops)
            self.dev_small[slot].copy_(self.pin_small[slot], non_blocking=True)
            self.ready[slot].record(st_)
        self.used[slot] = True

    def commit(self, slot):
        """Head of the step (main stream): wait for the slot and point every reader of the clip, the mask and the small inputs at it; the layout
        conversion is switched off.  No kernel, no copy."""
        eng, p = self.eng, self.eng.plan
        torch.cuda.current_stream(eng.dev).wait_event(self.ready[slot])
        for idx in p.op_to_ndhwc:
            eng.ops["fwd"][idx]["i"][1] = 0                                    # N = 0: nothing to convert
        ds = self.dev_small[slot].data_ptr()
        where = dict(img=self.x[slot].data_ptr(), seg=self.dev[slot]["loc_msk"].data_ptr(), cls=ds + 4 * self.o_cls, lab=ds + 4 * self.o_lab,
                     d832=ds + 4 * self.o_d832, d128=ds + 4 * self.o_d128)
        for key, (_base, readers) in eng._patched.items():
            for name, idx, q, offb in readers:
                eng.ops[name][idx]["p"][q] = where[key] + offb
        eng.labels_host, eng.action_host = self.host[slot]
