"""Host-side plan builder: turns the network of spec.py into flat lists of pc_op records
(include/picons.h) over a bump-allocated device arena.  Shape inference, memory planning and the
reverse (gradient) schedule are done ONCE here in Python; libpicons.so replays the lists.

Forward follows /root/reference/models/capsules_ucf101.py CapsNet.forward :413-512 and
models/pytorch_i3d.py :328-346; the two forward passes of a train step (main_ucf101.py:85-86)
run as ONE batch of 2n clips whose BatchNorm statistics are kept per half (`groups=2`).

Memory spaces (pointer = base[space] + byte offset, resolved by Plan.resolve):
  A arena (activations, gradients of activations, workspaces, kernel-layout weights)
  P flat parameters   G flat parameter gradients   M,V Adam moments   R BN running stats
"""
from dataclasses import dataclass, field

import os

import numpy as np

from . import capi, desc as D, spec, spectral, switches as sw, tail6

ALIGN = 256


@dataclass
class TR:
    """NDHWC tensor (or a channel slice of one): ref=(space, byte offset of the slice start)."""
    ref: tuple
    N: int
    thw: tuple
    C: int
    ld: int
    name: str = ""

    @property
    def rows(self):
        return self.N * self.thw[0] * self.thw[1] * self.thw[2]

    def slice(self, c0, c):
        return TR((self.ref[0], self.ref[1] + 4 * c0), self.N, self.thw, c, self.ld, self.name)


FOLD_AT = 128       # an ordered split-K weight gradient with more K slices than this is folded to groups before its re-layout


def off(ref, nfloats):
    return (ref[0], ref[1] + 4 * int(nfloats))


class Plan:
    def __init__(self, num_classes=24, hw=224, n=4, groups=2, training=True, jhmdb=False, accum_grads=False, lanes=1, spectral_pc=None,
                 early_adam=False, exp=None):
        """n = clips per forward pass; the batch the kernels see is N = groups*n.
        accum_grads: backward adds into the flat G buffer instead of overwriting it (drop-in nn.Module path,
        where the reference's two forward passes are two separate autograd graphs).
        lanes: HIP streams the runner may use; >1 tags the four branches of every Inception module (and their
        backward) onto separate lanes between FORK/JOIN ops, so the under-filled 14x14 launches overlap."""
        self.acc = 1 if accum_grads else 0
        self.exp = dict(exp or {})        # experiment switches passed explicitly (switches.py): the environment alone does not select them
        # early_adam: the backward list carries an (un-armed: length 0) Adam op over every parameter but the stem's, on a side lane in
        # front of the stem's backward -- the HBM-bound optimiser pass then runs beside the stem's MFMA-bound weight gradient, the last
        # kernel of the step, instead of behind it.  The caller arms it (StepEngine, single process); see build_backward.
        self.early_adam = bool(early_adam) and lanes >= 2 and training and sw.get("PICONS_EARLY_ADAM", "1") != "0"
        self.op_adam_early, self.adam_split = None, 0
        if hw % 8 or hw // 8 < spec.PRIMARY_K:
            # the reference fails the same way, inside nn.Conv2d (capsules_ucf101.py:43-49: a 9x9 valid conv on the hw/8 feature map)
            raise ValueError("frame size %d: must be a multiple of 8 and at least %d (the 9x9 PrimaryCaps conv needs a feature map of "
                             "9x9 or more)" % (hw, 8 * spec.PRIMARY_K))
        if not 1 <= lanes <= capi.MAX_LANES:
            raise ValueError("lanes must be 1..%d" % capi.MAX_LANES)
        self.lanes = lanes
        self.lane = 0
        # Lane roles (measured on MI355X): lanes 0 .. branch_lanes-1 carry the Inception branches; with >= 3 lanes one
        # lane (PICONS_SKIP_LANE=0 turns it off) carries the decoder's skip convs conv56 / conv112 -- big, chip-filling GEMMs that
        # depend on nothing but out56 / out112 -- beside the trunk: forward behind Conv3d_2c / Conv3d_1a, backward from the decoder's
        # gradient until the trunk's backward first touches d(out56); and every weight-gradient launch with its gradient re-layout
        # goes to a side lane too (PICONS_WGRAD_LANE=0 turns it off; a lane of its own with >= 4 lanes, else the skip lane):
        # nothing in the backward waits for a weight gradient except the optimiser, so the dgrad / BatchNorm chain and the wgrad
        # stream overlap, and their blocks fill the slots the under-filled 28x28 launches leave idle.
        want_skip = sw.get("PICONS_SKIP_LANE", "1") != "0" and lanes >= 3
        want_wg = sw.get("PICONS_WGRAD_LANE", "1") != "0" and lanes >= 3
        own_wg = want_wg and want_skip and lanes >= 4 and sw.get("PICONS_WGRAD_SEPARATE", "1") != "0"
        self.wg_lane = lanes - 1 if want_wg else 0
        self.skip_lane = (lanes - 2 if own_wg else lanes - 1) if want_skip else 0
        self.branch_lanes = lanes - len({ln for ln in (self.wg_lane, self.skip_lane) if ln})
        self.skip_bwd = {}
        self.final_lane = {}      # param name -> lane of the op that finalises its gradient
        self.deferred = None      # build_backward: side-lane closures held back until the EM backward is enqueued
        self.after_bn_bwd = None   # build_backward: emitted once, right behind the next BatchNorm backward op
        self.wgrad_collect = None  # inside a wgrad group: [(descriptor, pointer refs)] collected for one pc_conv_wgrad_multi op
        # Ordered split-K (round 6): the K slices of a weight gradient leave their partial sums as images in a workspace (plain stores) and
        # the gradient re-layout adds the images in slice order -- no fp32 atomics, no zero fill of the kernel-layout gradient, and every
        # gradient is bit-identical from run to run.  PICONS_WGRAD_ATOMIC=1 restores the atomic epilogue (the A/B switch).
        self.wg_ordered = sw.get("PICONS_WGRAD_ATOMIC", "0") == "0"
        # PrimaryCaps in its row-spectral form (spectral.py): a third of the direct form's FLOPs
        self.spectral_pc = (sw.get("PICONS_SPECTRAL", "1") != "0") if spectral_pc is None else bool(spectral_pc)
        # convolutions multiplied on the bf16 matrix cores (csrc/conv_x6.hip: fp32 operands as exact sums of three bf16 terms, six products,
        # fp32 accumulate); PICONS_SPLIT=0 keeps every launch on the fp32 MFMA kernels
        self.x6 = sw.get("PICONS_SPLIT", "1") != "0"
        self.wbufs = []           # kernel-layout weight buffers: dict(ref, n floats, lst / lane of the producing ops, planes ref once split)
        self.consts = []          # (arena ref, float32 ndarray): constant tables the owner uploads once (upload_consts)
        self.zero_once = []       # (arena ref, floats): buffers the owner zeroes once (upload_consts): workspaces whose counters every launch leaves zero
        # decoder tail as one five-tap transposed conv with a single output channel (csrc/tail6.hip) instead of the
        # 27-channel form + tap sum
        self.merged_tail = sw.get("PICONS_TAIL6", "1") != "0"
        # the decoder's forward sits on lane 0 with the side lanes idle: the position classes of a stride-2 transposed conv (8 independent
        # launches of 196 - 784 blocks) are dealt to all lanes, and conv28 (196 blocks, K = 7488) runs on the skip lane beside PrimaryCaps
        self.spread_classes = lanes >= 2 and sw.get("PICONS_SPREAD_CLASSES", "1") != "0"
        self.conv28_aside = sw.get("PICONS_CONV28_ASIDE", "1") != "0"
        self.C = num_classes
        self.hw = hw
        self.n = n
        self.groups = groups
        self.N = n * groups
        self.training = training
        self.jhmdb = jhmdb
        self.arena_bytes = 0
        self.lists = {"prep": [], "prep_late": [], "fwd": [], "loss": [], "bwd": [], "unprep": [], "adam": []}
        # Weight-layout prep of everything but the first trunk layers goes to `prep_late`: enqueued on the side lanes behind `prep`
        # and joined inside the forward list before Mixed_3b, so it runs beside the stem / Conv3d_2b / Conv3d_2c instead of in front
        # of them (0.65 ms of the step's critical path).  PICONS_LATE_PREP=0 or a plan without side lanes: everything in `prep`.
        self.late_prep = bool(self.wg_lane) and sw.get("PICONS_LATE_PREP", "1") != "0"
        self.prep_target = "prep_late" if self.late_prep else "prep"
        self.cur = "fwd"
        self.tape = []
        self.grads = {}           # buffer name -> [TR, initialised]
        self.kw = {}              # weight name -> dict(fwd=ref, tr=ref, ...)
        self.named = {}           # name -> TR / ref for host access
        # flat parameter layout (reference order / layout)
        self.pshape = spec.param_shapes(num_classes)
        # the three 1x1x1 Unit3Ds of an Inception module that read the module input (b1a, b2a, b0) run as ONE conv + BN
        # over their stacked output channels (unit3d with several prefixes): their BN parameters and running statistics
        # sit next to each other in the flat buffers, in that order.  Names and shapes are the reference's; only the
        # offsets inside the flat buffers differ from nn.Module.parameters() order.
        self.fuse_1x1 = sw.get("PICONS_FUSE1X1", "1") != "0"
        self.fused_groups = ([["conv1.%s.%s" % (ent[0], b) for b in ("b1a", "b2a", "b0")] for ent in spec.TRUNK if ent[1] == "mixed"]
                             if self.fuse_1x1 else [])
        head = {g[2] + ".conv3d.weight": g for g in self.fused_groups}      # b0 comes first in reference order
        grouped = {pre + sfx for g in self.fused_groups for pre in g for sfx in (".conv3d.weight", ".bn.weight", ".bn.bias")}
        order = []
        for k in self.pshape:
            if k in head:
                order += [pre + sfx for sfx in (".conv3d.weight", ".bn.weight", ".bn.bias") for pre in head[k]]
            elif k not in grouped:
                order.append(k)
        assert sorted(order) == sorted(self.pshape)
        self.poff = {}
        o = 0
        for k in order:
            self.poff[k] = o
            o += int(np.prod(self.pshape[k]))
            o = (o + 3) // 4 * 4       # keep every tensor 16-byte aligned
        self.nparams = o
        self.roff = {}
        o = 0
        units = {pre: co for pre, _ci, co, _k, _s in spec.trunk_units()}
        rgroup = {g[2]: g for g in self.fused_groups}
        rskip = {pre for g in self.fused_groups for pre in g}
        for pre in units:
            for grp in ([rgroup[pre]] if pre in rgroup else ([] if pre in rskip else [[pre]])):
                for stat in (".bn.running_mean", ".bn.running_var"):
                    for q in grp:
                        self.roff[q + stat] = o; o += units[q]
        self.nrunning = o
        # kernel-layout weight-gradient buffers (wgrad accumulates into them with atomics) live in ONE region so a
        # single fill zeroes them all each step; their total size is the parameter count plus channel padding
        self.kg_cap = 4 * (self.nparams + (1 << 20))
        self.kg_base = self.alloc(self.kg_cap // 4)
        self.kg_used = 0
        self.final_at = {}        # param name -> number of bwd ops after which its gradient in G is final
        self.alg_flops = {}       # list name -> conv FLOPs as the layer's own formulation counts them (all taps, padding included)
        self.alg_flops_wino = {}  # the same for the layers that run in Winograd form (their launches are not OP_CONV)
        self.issued = {}          # (list name, op kind) -> FLOPs the emitted (trimmed) descriptors multiply, real channel counts
        self.work = {}            # (list name, "mfma" | "executed" | "valid") -> conv / dgrad FLOPs as the kernels run them (pc_conv_work)
        self.op_work = {}         # id(op's int list) -> pc_conv_work / pc_wgrad_work of that launch

    # ------------------------------------------------------------------ memory
    def alloc(self, nfloats, name=""):
        o = self.arena_bytes
        self.arena_bytes = (o + 4 * int(nfloats) + ALIGN - 1) // ALIGN * ALIGN
        return ("A", o)

    def const(self, arr):
        """Constant table in the arena (uploaded once by the owner of the arena)."""
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        ref = self.alloc(arr.size)
        self.consts.append((ref, arr))
        return ref

    def upload_consts(self, view):
        """view(ref, nfloats) -> float32 device tensor over the arena."""
        import torch
        for ref, arr in self.consts:
            view(ref, arr.size).copy_(torch.from_numpy(arr.reshape(-1)))
        for ref, n in self.zero_once:
            view(ref, n).zero_()

    def alloc_kg(self, nfloats):
        o = self.kg_used
        self.kg_used = (o + 4 * int(nfloats) + ALIGN - 1) // ALIGN * ALIGN
        if self.kg_used > self.kg_cap:
            raise RuntimeError("kernel-layout gradient region overflow")
        return off(self.kg_base, o // 4)

    def tensor(self, N, thw, C, name):
        t = TR(self.alloc(N * thw[0] * thw[1] * thw[2] * C), N, tuple(thw), C, C, name)
        self.named[name] = t
        return t

    def P(self, name):
        return ("P", 4 * self.poff[name])

    def G(self, name):
        return ("G", 4 * self.poff[name])

    def R(self, name):
        return ("R", 4 * self.roff[name])

    # ------------------------------------------------------------------ op emission
    def emit(self, kind, i=(), f=(), p=(), l=(), lst=None, lane=None):
        if lane is None:
            lane = self.lane if lst in (None, self.cur) else 0
        il = list(i)
        self.lists[lst or self.cur].append((kind, il, list(f), list(p), list(l), lane))
        return il

    def fork(self, mask=None, src=0):
        """The lanes in `mask` (default: the branch lanes) wait for everything enqueued so far on lane `src` (no-op for a
        single-lane plan)."""
        mask = (1 << self.branch_lanes) - 2 if mask is None else mask
        mask &= ~(1 << src)
        if self.lanes > 1 and mask:
            self.lists[self.cur].append((capi.OP_FORK, [mask, src], [], [], [], 0))

    def join(self, mask=None):
        """Lane 0 waits for everything enqueued so far on the side lanes in `mask` (default: the branch lanes)."""
        mask = (1 << self.branch_lanes) - 2 if mask is None else mask
        if self.lanes > 1 and mask:
            self.lists[self.cur].append((capi.OP_JOIN, [mask], [], [], [], 0))

    def grad_for_write(self, x):
        """Gradient buffer of x's (whole) buffer for a consumer's backward: (TR, accumulate?)."""
        if x.name not in self.grads:
            base = self.named[x.name]
            g = TR(self.alloc(base.rows * base.ld), base.N, base.thw, base.ld, base.ld, "d_" + x.name)
            self.grads[x.name] = [g, {}]
        g, written = self.grads[x.name]
        delta = x.ref[1] - self.named[x.name].ref[1]
        # first writer of a channel slice overwrites, later ones accumulate (slices of one buffer that are written
        # separately - the module output and the b1a / b2a intermediates ahead of it - never overlap)
        init = written.get((delta, x.C), False)
        written[(delta, x.C)] = True
        return TR((g.ref[0], g.ref[1] + delta), x.N, x.thw, x.C, x.ld, g.name), init

    def grad_of(self, y):
        if y.name not in self.grads or not self.grads[y.name][1]:       # no slice written yet
            raise RuntimeError("gradient of %s requested before any consumer wrote it" % y.name)
        g = self.grads[y.name][0]
        delta = y.ref[1] - self.named[y.name].ref[1]
        return TR((g.ref[0], g.ref[1] + delta), y.N, y.thw, y.C, y.ld, g.name)

    # ------------------------------------------------------------------ weights
    def prep_conv_weight(self, names, O_list, I, k, need_tr, Ipad=None, layouts=True):
        """Reference OI(T)HW masters (one or several stacked along O) -> kernel layouts:
        fwd [O][taps][Ipad], tr [I][taps][O] (for dgrad).  Also the kernel-layout grad buffer.
        layouts=False (the layer's forward and input gradient run in Winograd form, from U / U^T built off the master weights): only the
        kernel-layout gradient buffer and its way back into G are made -- no fwd / tr copies, no transposes."""
        taps = k[0] * k[1] * k[2]
        Ipad = Ipad or I
        O = sum(O_list)
        key = names[0]
        saved_target = self.prep_target
        if self.late_prep and any(key.startswith("conv1." + u) for u in ("Conv3d_1a_7x7", "Conv3d_2b_1x1", "Conv3d_2c_3x3")):
            self.prep_target = "prep"          # needed before anything else runs
        w = dict(O=O, I=I, Ipad=Ipad, taps=taps, kg=self.alloc_kg(O * taps * Ipad), unprep=[])
        pl = self.next_prep_lane()
        if layouts:
            w["fwd"] = self.alloc(O * taps * Ipad)
            if Ipad != I:
                self.emit(capi.OP_FILL, p=[w["fwd"]], l=[O * taps * Ipad], f=[0.0], lst=self.prep_target, lane=pl)
            if need_tr:
                w["tr"] = self.alloc(I * taps * O)
        o0 = 0
        for nm, Oi in zip(names, O_list):
            src = self.P(nm)
            if layouts:
                self.emit(capi.OP_TRANSPOSE, i=[Oi, I, taps, taps, Ipad, 0], l=[I * taps, taps * Ipad],
                          p=[src, off(w["fwd"], o0 * taps * Ipad)], lst=self.prep_target, lane=pl)
                if need_tr:
                    self.emit(capi.OP_TRANSPOSE, i=[1, Oi, I * taps, I * taps, O, 0], l=[0, 0], p=[src, off(w["tr"], o0)], lst=self.prep_target, lane=pl)
            # grad back: kg [Oi][taps][Ipad] -> G [Oi][I][taps]  (flushed right after the wgrad, see flush_grad)
            w["unprep"].append((nm, (capi.OP_TRANSPOSE, [Oi, taps, I, Ipad, taps, self.acc], [], [off(w["kg"], o0 * taps * Ipad), self.G(nm)], [taps * Ipad, I * taps])))
            o0 += Oi
        if layouts:
            self.reg_weight(w["fwd"], O * taps * Ipad, self.prep_target, pl)
            if need_tr:
                self.reg_weight(w["tr"], I * taps * O, self.prep_target, pl)
        self.kw[key] = w
        self.prep_target = saved_target
        return w

    def prep_convT_weight(self, name, I, O, k):
        """Reference IO(T)HW master -> fwd-type layout [O][taps][I] (ConvTranspose forward) and
        [I][taps][O] (its dgrad, a strided conv); grads come back as [I][taps][O]."""
        taps = k[0] * k[1] * k[2]
        w = dict(O=O, I=I, taps=taps, fwd=self.alloc(O * taps * I), tr=self.alloc(I * taps * O), kg=self.alloc_kg(I * taps * O), unprep=[])
        src = self.P(name)
        pl = self.next_prep_lane()
        self.emit(capi.OP_TRANSPOSE, i=[1, I, O * taps, O * taps, I, 0], l=[0, 0], p=[src, w["fwd"]], lst=self.prep_target, lane=pl)
        self.emit(capi.OP_TRANSPOSE, i=[I, O, taps, taps, O, 0], l=[O * taps, taps * O], p=[src, w["tr"]], lst=self.prep_target, lane=pl)
        w["unprep"].append((name, (capi.OP_TRANSPOSE, [I, taps, O, O, taps, self.acc], [], [w["kg"], self.G(name)], [taps * O, O * taps])))
        self.reg_weight(w["fwd"], O * taps * I, self.prep_target, pl)
        self.reg_weight(w["tr"], I * taps * O, self.prep_target, pl)
        self.kw[name] = w
        return w

    def next_prep_lane(self):
        """Weight-layout prep (one or two tiny transposes per parameter, ~160 launches a step) is spread round-robin
        over the lanes; ops of one weight stay on one lane, in order."""
        self._prep_rr = (getattr(self, "_prep_rr", -1) + 1) % self.lanes
        return self._prep_rr

    def alg_dgrad(self, fwd_flops, wino=False):
        tab = self.alg_flops_wino if wino else self.alg_flops
        tab[self.cur] = tab.get(self.cur, 0) + fwd_flops

    def flush_grad(self, w):
        """Kernel-layout weight gradient -> reference layout in the flat G buffer, emitted right after the
        wgrad's layer (flush_unprep, after every layer / Inception module of the backward) so a gradient bucket is final
        (all-reduce can start) as early as possible.  The re-layout runs on the lane its wgrad ran on."""
        self._pending_unprep = getattr(self, "_pending_unprep", [])
        # (a wgrad on a branch lane is joined before the flush: its re-layout runs on whatever lane flushes)
        lane = self.lane if (self.wg_lane and self.lane == self.wg_lane) else None
        if self.wgrad_collect is not None and self.wg_lane:
            lane = self.wg_lane            # the group's launch runs on the weight-gradient lane: so does the re-layout
        self._pending_unprep.extend((nm, op, lane) for nm, op in w["unprep"])

    def flush_unprep(self):
        """Emit the pending kernel-layout -> reference-layout gradient transposes (several weights of an Inception module
        as ONE multi-job launch per lane) and mark those parameters final."""
        pend = getattr(self, "_pending_unprep", [])
        if not pend:
            return
        pend = [(nm, op, self.lane if ln is None else ln) for nm, op, ln in pend]
        for lane in sorted({ln for _nm, _op, ln in pend}):
            mine = [(nm, op) for nm, op, ln in pend if ln == lane]
            if len(mine) == 1:
                self.lists[self.cur].append(mine[0][1] + (lane,))
            else:
                self.multi_jobs = getattr(self, "multi_jobs", [])
                self.multi_jobs.append([op for _nm, op in mine])
                self.lists[self.cur].append((capi.OP_TRANSPOSE_MULTI, [len(mine)], [], [("JOBS", len(self.multi_jobs) - 1)], [], lane))
            saved, self.lane = self.lane, lane
            self.mark_final(*[nm for nm, _op in mine])
            self.lane = saved
        self._pending_unprep = []

    def on_wgrad_lane(self, fn):
        """Run the emissions of fn (a weight-gradient launch and what hangs off it) on the wgrad lane, behind the current lane."""
        if not self.wg_lane or self.lane == self.wg_lane or (self.wgrad_collect is not None and fn.__name__ != "emit_all"):
            return fn()
        if self.deferred is not None:
            self.deferred.append(lambda: self.on_wgrad_lane(fn))
            return
        self.fork(1 << self.wg_lane, src=self.lane)
        saved, self.lane = self.lane, self.wg_lane
        fn()
        self.lane = saved

    def mark_final(self, *names):
        for nm in names:
            self.final_at[nm] = len(self.lists["bwd"])
            self.final_lane[nm] = self.lane

    def bn_finalize_or_defer(self, nrows, cout, z, part, gamma, beta, pre, stat):
        """BatchNorm(train) statistics of a Unit3D from the conv's partial rows: either the finalize launch is emitted here (-> None), or --
        few partial rows per batch group AND PICONS_BN_FUSED=1 (off by default: no gain measured) -- the caller folds it into the apply launch (-> what that op needs)."""
        npg = nrows // self.groups
        if sw.exp("PICONS_BN_FUSED", "0", self.exp) != "0" and capi.lib().pc_bn_finalize_apply_ok(int(npg), int(cout)):
            return dict(npg=npg, part=part)
        self.emit(capi.OP_BN_FINALIZE, i=[npg, self.groups, cout], l=[z.rows // self.groups], f=[spec.BN_EPS, spec.BN_MOMENTUM],
                  p=[part, gamma, beta, self.R(pre + ".bn.running_mean"), self.R(pre + ".bn.running_var"), stat, self.bn_fin_ws(npg, cout)])
        return None

    def bn_fin_ws(self, npg, C):
        """Workspace of the two-stage BatchNorm finalize (pc_bn_finalize_ws; None below 512 partial rows per group)."""
        n = capi.lib().pc_bn_finalize_ws_floats(int(npg), int(self.groups), int(C))
        return self.alloc(n) if n > 0 else None

    # ------------------------------------------------------------------ Winograd F(2x2, 3x3) form of the stride-1 3x3x3 layers
    def wino_ok(self, x, cout, k, stride, pad=None):
        """The layers that run in Winograd form (csrc/wino.hip, 2.25x fewer multiply-accumulates, measured 1.5 - 2.2x faster than
        the gather-GEMM kernel on them): 3x3x3 convs with spatial stride 1 (any temporal stride: the temporal taps stay direct) and same
        padding, even H, W and enough channels to fill the kernel's 64 x 64 (tiles x channels) block.  The one-frame 28x28 layers
        (Mixed_4b..4f, 1x3x3 once the padding taps are dropped) are 0.97x alone -- 320 whole-CU blocks on 256 CUs -- but 2.25x fewer
        MFMAs in a step whose lanes keep the chip full: kept (PICONS_WINO_T1=0 restores the gather-GEMM form).  PICONS_WINO=0: off."""
        if sw.get("PICONS_WINO", "1") == "0":
            return False
        if tuple(k) != (3, 3, 3) or tuple(stride[1:]) != (1, 1) or stride[0] not in (1, 2) or (pad is not None and tuple(pad) != (1, 1, 1)):
            return False
        T, H, W = x.thw
        tmin = 1 if sw.get("PICONS_WINO_T1", "1") != "0" else 2     # one frame (Mixed_4b..4f: only the centre temporal tap is real): -0.13 ms on the step
        # the input gradient runs the same kernel with the roles swapped: its Ci is cout, read from a gradient tensor whose leading dimension is cout
        return (T >= tmin and H % 2 == 0 and W % 2 == 0 and x.C % 8 == 0 and x.C >= int(sw.get("PICONS_WINO_CMIN", "32")) and cout >= 64
                and cout % 8 == 0 and x.ld % 4 == 0)

    def wino_m(self, x):
        """Output tile edge of a Winograd layer: 4 -- F(4x4, 3x3), csrc/wino4.hip: 1.78x fewer MFMAs than F(2x2, 3x3), measured 1.42 - 1.68x
        faster on the 112 x 112 and 56 x 56 layers (tools/bench_wino4.py) -- where H, W are multiples of 4 and a frame has at least
        PICONS_WINO4_MIN_TILES (196 = 56 x 56) tiles.  The 28 x 28 layers stay on F(2x2, 3x3): most of them are slower alone (64 - 192
        whole-CU blocks), the step is 0.3 ms faster with them -- and five step-level parity tests fail (they are the trunk, in
        front of EM routing).  PICONS_WINO4=0: F(2x2, 3x3) everywhere."""
        _T, H, W = x.thw
        if sw.get("PICONS_WINO4", "1") == "0" or H % 4 or W % 4:
            return 2
        return 4 if (H // 4) * (W // 4) >= int(sw.exp("PICONS_WINO4_MIN_TILES", "196", self.exp)) else 2

    def wino_weights(self, wname, O, I, need_tr, m=2, m_tr=None):
        """Transform-domain weights of a layer, built per step straight from the master OIDHW parameter (and, for the input
        gradient, from its transpose with mirrored taps: strides + flip, no intermediate layout).  m / m_tr: the Winograd form (2 or 4) of the
        forward / of the input gradient."""
        m_tr = m if m_tr is None else m_tr
        u_floats = capi.lib().pc_wino4_u_floats if m == 4 else capi.lib().pc_wino_u_floats
        u_floats_tr = capi.lib().pc_wino4_u_floats if m_tr == 4 else capi.lib().pc_wino_u_floats
        nU = u_floats(O, I, 3)
        if nU <= 0 or (need_tr and u_floats_tr(I, O, 3) <= 0):
            raise ValueError("Winograd form of %s: %d -> %d channels is not a shape the kernel takes" % (wname, I, O))
        u = dict(fwd=self.alloc(nU))
        src = self.P(wname)
        saved_target = self.prep_target
        if self.late_prep and wname.startswith("conv1.Conv3d_2c_3x3"):
            self.prep_target = "prep"          # needed before the late-prep list has run
        pl = self.next_prep_lane()
        self.emit(capi.OP_WINO_WEIGHTS, i=[O, I, 3, 0, m], l=[I * 27, 1, 27], p=[src, u["fwd"]], lst=self.prep_target, lane=pl)
        if need_tr:
            u["tr"] = self.alloc(u_floats_tr(I, O, 3))
            self.emit(capi.OP_WINO_WEIGHTS, i=[I, O, 3, 1, m_tr], l=[27, 1, I * 27], p=[src, u["tr"]], lst=self.prep_target, lane=pl)
        self.prep_target = saved_target
        if wname in self.kw:               # finalize() lays a skip conv's weights out on the skip lane, in front of the conv
            self.kw[wname]["wino_fwd"], self.kw[wname]["wino_tr"] = u["fwd"], u.get("tr")
        u["m"], u["m_tr"] = m, m_tr
        return u

    @staticmethod
    def _wino_desc(N, othw, Ti, Ci, ldi, Co, ldo, tmap, act=0, flags=0, m=2):
        d = capi.WinoDesc()
        d.m = m
        if m != 4 and sw.get("PICONS_WINO_STRIPS", "0") != "0":
            flags |= capi.F_STRIPS          # F(2x2, 3x3) blocks of two 2 x 14-tile strips (bit-identical; measured neutral on the step, off by default)
        d.N, d.T, d.H, d.W, d.Ci, d.ldi, d.Co, d.ldo, d.KT, d.act, d.flags = N, othw[0], othw[1], othw[2], Ci, ldi, Co, ldo, 3, act, flags
        d.Ti, d.ta, d.tc, d.tden = Ti, tmap[0], tmap[1], tmap[2]
        return d

    def wino_op(self, x_ref, N, othw, Ti, Ci, ldi, Co, ldo, tmap, U, out_ref, bias=None, bnpart=None, act=0, flags=0, m=2):
        """One pc_wino_conv launch: output frames / positions othw, Ti input frames, tmap = (ta, tc, tden) of pc_wino_desc."""
        import ctypes as C
        d = self._wino_desc(N, othw, Ti, Ci, ldi, Co, ldo, tmap, act, flags, m)
        out = (C.c_double * 3)()
        capi.check(capi.lib().pc_wino_work(C.byref(d), out))
        for key, v in (("wino_mfma", out[0]), ("wino_executed", out[1])):
            self.work[(self.cur, key)] = self.work.get((self.cur, key), 0) + 2 * v
        il = self.emit(capi.OP_WINO_CONV, i=[getattr(d, f) for f, _t in capi.WinoDesc._fields_], p=[x_ref, U, bias, out_ref, bnpart])
        self.op_work[id(il)] = dict(issued=out[0], executed=out[1], valid=out[1], blocks=int(out[2]), wino=True)
        return d

    def wino_bnpart_rows(self, N, othw, Ti, Ci, ldi, Co, ldo, tmap, m=2):
        import ctypes as C
        return capi.lib().pc_wino_bnpart_rows(C.byref(self._wino_desc(N, othw, Ti, Ci, ldi, Co, ldo, tmap, m=m)))

    def wino_flops_executed(self):
        """Per list: FLOPs of the Winograd launches -- `mfma` issued to the matrix cores (whole 64 x 64 blocks), `executed` on real tiles
        and channels (transform-domain multiply-accumulates: 16 per 2x2 output tile, tap and channel pair, where the direct form does 36)."""
        return {name: {k: self.work.get((name, "wino_" + k), 0) for k in ("mfma", "executed")} for name in self.lists}

    # ------------------------------------------------------------------ weight planes for the bf16-split conv kernel
    def reg_weight(self, ref, nfloats, lst=None, lane=None):
        """A kernel-layout weight buffer [ref, ref + nfloats) whose producing ops sit in list `lst` on lane `lane` (None: wherever the first
        consumer runs).  conv_op splits it into bf16 planes the first time a launch that takes the bf16-split kernel reads it."""
        self.wbufs.append(dict(ref=ref, n=int(nfloats), lst=lst, lane=lane, planes=None))

    def _wbuf_of(self, w_ref):
        for b in self.wbufs:
            if b["ref"][0] == w_ref[0] and b["ref"][1] <= w_ref[1] < b["ref"][1] + 4 * b["n"]:
                return b
        return None

    def maybe_x6(self, d, w_ref):
        """The descriptor with PC_F_X6 if this launch takes the bf16-split kernel (the switch is on, the library says the shape fits, and the
        weights are a registered kernel-layout buffer that can be split into planes), else unchanged."""
        if not self.x6 or (d["flags"] & capi.F_X6) or self._wbuf_of(w_ref) is None:
            return d
        if sw.exp("PICONS_SPLIT_ONLY_SPECTRAL", "0", self.exp) != "0" and w_ref[0] != "V":     # A/B diagnostics (tools/probe_tensor_grad.py)
            return d
        if w_ref[0] == "V":           # weights that exist as planes only: no choice
            return dict(d, flags=d["flags"] | capi.F_X6) if capi.lib().pc_conv_x6_ok(_cdesc(dict(D.trim_conv(d), flags=d["flags"] | capi.F_X6))) else d
        if self.cur not in sw.exp("PICONS_SPLIT_LISTS", "fwd,bwd", self.exp).split(","):
            return d
        if not int(sw.exp("PICONS_SPLIT_CI_MIN", "0", self.exp)) <= d["Ci"] <= int(sw.exp("PICONS_SPLIT_CI_MAX", "1000000", self.exp)):
            return d
        if not int(sw.exp("PICONS_SPLIT_ROWS_MIN", "0", self.exp)) <= d["N"] * d["Tq"] * d["Hq"] * d["Wq"] <= int(sw.exp("PICONS_SPLIT_ROWS_MAX", "2000000000", self.exp)):
            return d
        t = dict(D.trim_conv(d), flags=d["flags"] | capi.F_X6)
        if not capi.lib().pc_conv_x6_ok(_cdesc(t)):
            return d
        return dict(d, flags=d["flags"] | capi.F_X6)

    def planes_of(self, w_ref):
        """-> (ref of the bf16 planes at w_ref's offset, plane stride in elements); the split op is emitted once per buffer, behind the
        buffer's producers (same list, same lane) or, for a buffer made inside the forward / backward list, in front of its first reader."""
        b = self._wbuf_of(w_ref)
        if b["planes"] is None:
            assert b["n"] % 8 == 0, "weight buffer of %d floats" % b["n"]
            b["planes"] = self.alloc((3 * b["n"] + 1) // 2)
            if b["lst"] in ("prep", "prep_late"):
                self.emit(capi.OP_SPLIT_PLANES, p=[b["ref"], b["planes"]], l=[b["n"], b["n"]], lst=b["lst"], lane=b["lane"])
            else:
                self.emit(capi.OP_SPLIT_PLANES, p=[b["ref"], b["planes"]], l=[b["n"], b["n"]])
        return (b["planes"][0], b["planes"][1] + (w_ref[1] - b["ref"][1]) // 2), b["n"]

    # ------------------------------------------------------------------ layers
    def conv_op(self, d, x_ref, w_ref, out_ref, bias=None, cscale=None, bnpart=None, alg=None, x6=True):
        """alg: algorithmic FLOPs to book for this launch (default: the descriptor's own 2*M*N*K with all taps).
        Dgrad launches book the layer's FORWARD FLOPs once (alg_dgrad) and pass alg=0, so gather-form overheads
        (padding taps, the 28x28 gather of the 20x20 PrimaryCaps dgrad) never inflate the roofline numerator."""
        self.alg_flops[self.cur] = self.alg_flops.get(self.cur, 0) + (_conv_flops(d) if alg is None else alg)
        if x6:
            d = self.maybe_x6(d, w_ref)
        t = D.trim_conv(d)
        self.issued[(self.cur, capi.OP_CONV)] = self.issued.get((self.cur, capi.OP_CONV), 0) + _conv_flops(t)
        w = conv_work(t)
        fam = "x6_" if t["flags"] & capi.F_X6 else ""
        for key, v in (("mfma", w["issued"]), ("executed", w["executed"]), ("valid", w["valid"])):
            self.work[(self.cur, fam + key)] = self.work.get((self.cur, fam + key), 0) + 2 * v
        if w_ref[0] == "V" and not (t["flags"] & capi.F_X6):
            raise RuntimeError("weights that exist as bf16 planes only met a launch that does not take the bf16-split kernel")
        if t["flags"] & capi.F_X6:
            wp, pstride = self.planes_of(w_ref)
            # workspace for the launch's tail split (csrc/conv_x6.hip: the tiles of the last, partly filled round of resident blocks run as K
            # slices); private to the op, its counters zeroed once by the arena's owner (upload_consts) and left zero by every launch
            n_ws = int(capi.lib().pc_conv_x6_ws_floats(_cdesc(t))) if sw.get("PICONS_X6_TAIL_SPLIT", "1") != "0" else 0
            ws = None
            if n_ws > 0:
                ws = self.alloc(n_ws)
                self.zero_once.append((ws, n_ws))
            il = self.emit(capi.OP_CONV_X6, i=D.flatten(t, D.CONV_FIELDS), p=[x_ref, wp, bias, cscale, out_ref, bnpart, ws], l=[pstride, n_ws])
        else:
            il = self.emit(capi.OP_CONV, i=D.flatten(t, D.CONV_FIELDS), p=[x_ref, w_ref, bias, cscale, out_ref, bnpart])
        self.op_work[id(il)] = w          # keyed by the op's own int list (it survives the re-laning of finalize()): tools/launch_table.py

    def wgrad_op(self, d, p, w=None):
        """Emit a weight-gradient launch and book the FLOPs its (already trimmed) descriptor multiplies.  Inside a wgrad_group the
        launch is only collected: the group leaves as ONE pc_conv_wgrad_multi op.
        w: the weight record (prep_conv_weight / prep_convT_weight) whose re-layout consumes the result -- with ordered split-K the launch
        writes K-slice images into a workspace of its own (p[2] is replaced) and w's re-layout jobs are pointed at them."""
        if self.x6 and sw.get("PICONS_SPLIT_WGRAD", "1") != "0":
            d = dict(d, flags=int(d.get("flags", 0)) | capi.WG_X6)       # row-segment and generic split-K routes (the stem and the 9-tap spectral planes stay fp32)
        if w is not None and self.wg_ordered and int(d.get("splitk", 0)) != -1:
            ns = capi.lib().pc_wgrad_slices(_wdesc(d))
            if ns < 1:
                raise RuntimeError("pc_wgrad_slices: %s" % capi.lib().pc_last_error().decode())
            image = d["Cd"] * d["KT"] * d["KH"] * d["KW"] * d["Cs"]
            ws = self.alloc(ns * image)
            self.zero_once.append((ws, ns * image))          # trimmed taps / padding channels / empty slices are never written
            d = dict(d, ws_slices=ns)
            p = list(p)
            kg, p[2] = p[2], ws
            nim, stride = ns, image
            if ns > FOLD_AT:        # many slices: folded in place to groups first (pc_wgrad_fold), the re-layout then adds the group sums
                G = capi.lib().pc_wgrad_fold_group()
                d["_post"] = [(capi.OP_WGRAD_FOLD, [ns], [ws], [image])]
                nim, stride = -(-ns // G), G * image
            # taps the descriptor trimmed away (Mixed_4b..4f: T = 1, only the centre temporal tap is ever real) are never written: their rows
            # of G come from ONE (zero) image instead of being "summed" over every slice
            taps, khw = d["KT"] * d["KH"] * d["KW"], d["KH"] * d["KW"]
            r0, r1 = d["wk0"][0] * khw, (d["wk0"][0] + d["ntap"][0]) * khw
            jobs = []
            for nm, (kind, i, f, pp, l) in w["unprep"]:
                src = off(ws, (pp[0][1] - kg[1]) // 4)
                assert kind == capi.OP_TRANSPOSE and i[1] == taps, (nm, i)
                for a, b, n_ in ((0, r0, 1), (r0, r1, nim), (r1, taps, 1)):
                    if b > a:
                        jobs.append((nm, (kind, [i[0], b - a, i[2], i[3], i[4], i[5], n_], f, [off(src, a * i[3]), off(pp[1], a)], list(l) + [stride])))
            w["unprep"] = jobs
            w["ordered"] = True
        self.issued[(self.cur, capi.OP_WGRAD)] = self.issued.get((self.cur, capi.OP_WGRAD), 0) + _wgrad_flops(d)
        wk = wgrad_work(d)
        fam = "wgx6_" if capi.lib().pc_wgrad_uses_x6(_wdesc(d)) else "wgf32_"       # the 9-tap spectral route ignores PC_WG_X6
        for key, v in (("mfma", wk["issued"]), ("executed", wk["executed"]), ("valid", wk["valid"])):
            for pre in ("wg_", fam):
                self.work[(self.cur, pre + key)] = self.work.get((self.cur, pre + key), 0) + 2 * v
        if self.wgrad_collect is not None:
            self.wgrad_collect.append((d, list(p)))
            return
        self._emit_wgrad(d, p, wk)

    def _emit_wgrad(self, d, p, wk=None):
        self.op_work[id(self.emit(capi.OP_WGRAD, i=D.flatten(d, D.WGRAD_FIELDS), p=p))] = wk or wgrad_work(d)
        for kind, i, pp, l in d.get("_post", ()):
            self.emit(kind, i=i, p=pp, l=l)

    def wgrad_group_begin(self):
        """Weight gradients emitted until wgrad_group_end() are held back and leave together at the group's end, on the weight-gradient
        lane: as individual launches by default, as ONE multi-problem launch with PICONS_WGRAD_MULTI=1 (0.3 ms less kernel time per step,
        but measured 0.23 ms slower on the step: a chip-filling grouped launch competes with lane 0 where the small ones slip into gaps)."""
        self.wgrad_collect = []

    def wgrad_group_end(self):
        jobs, self.wgrad_collect = self.wgrad_collect, None
        if not jobs:
            return

        def emit_all():
            if len(jobs) == 1 or self.wg_ordered or sw.exp("PICONS_WGRAD_MULTI", "0", self.exp) == "0":
                for d, p in jobs:
                    self._emit_wgrad(d, p)
            else:
                self.wjobs = getattr(self, "wjobs", [])
                self.wjobs.append(jobs)
                self.emit(capi.OP_WGRAD_MULTI, i=[len(jobs)], p=[("WJOBS", len(self.wjobs) - 1)])
        self.on_wgrad_lane(emit_all)

    def unit3d(self, pre, x, cout, k, stride, out=None, need_dx=True):
        """Unit3D (pytorch_i3d.py:89-120): SAME conv (no bias) -> BN(train) -> ReLU.
        pre / cout may be lists: several Unit3Ds with the same kernel that read the same x run as ONE conv + BN + ReLU over
        their stacked output channels (BN is per channel, so stacking is exact); their BN parameters and running
        statistics must be adjacent in the flat buffers in that order (Plan.__init__ lays the fused groups out so)."""
        pres, couts = ([pre], [cout]) if isinstance(pre, str) else (list(pre), list(cout))
        pre, cout = pres[0], sum(couts)
        for tab, sfxs in ((self.poff, (".bn.weight", ".bn.bias")), (self.roff, (".bn.running_mean", ".bn.running_var"))):
            for sfx in sfxs:
                o = tab[pre + sfx]
                for q, c in zip(pres, couts):
                    if tab[q + sfx] != o:
                        raise RuntimeError("stacked Unit3Ds need adjacent %s (%s)" % (sfx, q))
                    o += c
        Ci = x.C
        Ci_real = self.pshape[pre + ".conv3d.weight"][1]          # 3 for the RGB stem, whose clip is padded to 4 channels
        othw = tuple(spec.same_out(x.thw[i], k[i], stride[i]) for i in range(3))
        pf = [spec.same_pad(x.thw[i], k[i], stride[i])[0] for i in range(3)]
        wino = len(pres) == 1 and Ci == Ci_real and self.wino_ok(x, cout, k, stride)
        w = self.prep_conv_weight([q + ".conv3d.weight" for q in pres], couts, Ci_real, k, need_dx, Ipad=Ci, layouts=not wino)
        z = self.tensor(x.N, othw, cout, pre + ".z")
        y = out if out is not None else self.tensor(x.N, othw, cout, pre + ".y")
        stat = self.alloc(self.groups * 4 * cout)
        gamma, beta = self.P(pre + ".bn.weight"), self.P(pre + ".bn.bias")
        F_fwd = _conv_flops(dict(D.conv_fwd(x.N, x.thw, Ci, x.ld, cout, z.ld, k, stride, pf, othw), Ci_real=Ci_real))
        ci3 = Ci == 4 and Ci_real == 3             # the RGB clip: the padding channel's MFMAs are not issued (PC_F_CI3 / PC_WG_CS3)
        # F(4x4, 3x3) for the input gradient; the trunk's FORWARD stays in F(2x2, 3x3) (PICONS_WINO4_TRUNK_FWD=1: F(4x4, 3x3) too): EM routing
        # amplifies any perturbation of the trunk's forward arithmetic -- with Conv3d_2c's forward in F(4x4, 3x3) (2.9x the rms rounding error) the
        # stem's BatchNorm gradients sit at 2.0 - 2.4x the fp32 reference's own distance from its fp64 run instead of 1.0 - 1.2x
        wm_b = self.wino_m(x) if wino else 2
        wm = wm_b if sw.exp("PICONS_WINO4_TRUNK_FWD", "0", self.exp) == "1" else 2
        wu = self.wino_weights(pre + ".conv3d.weight", cout, Ci, need_dx and self.training, wm, wm_b) if wino else None
        fin = None
        tmap_f = (stride[0], -pf[0], 1)                    # forward: tap kt of output frame t reads input frame t * s - pad_front + kt
        tmap_b = (1, pf[0] - 2, stride[0])                 # input gradient (mirrored taps): frame (t + pad_front - 2 + kt) / s
        if self.training and wino:
            nrows = self.wino_bnpart_rows(x.N, othw, x.thw[0], Ci, x.ld, cout, z.ld, tmap_f, wm)
            part = self.alloc(nrows * 2 * cout)
            self.alg_flops_wino[self.cur] = self.alg_flops_wino.get(self.cur, 0) + F_fwd
            self.wino_op(x.ref, x.N, othw, x.thw[0], Ci, x.ld, cout, z.ld, tmap_f, wu["fwd"], z.ref, bnpart=part, flags=capi.F_BNPART, m=wm)
            fin = self.bn_finalize_or_defer(nrows, cout, z, part, gamma, beta, pre, stat)
            g_apply = self.groups
        elif wino:
            self.alg_flops_wino[self.cur] = self.alg_flops_wino.get(self.cur, 0) + F_fwd
            self.wino_op(x.ref, x.N, othw, x.thw[0], Ci, x.ld, cout, z.ld, tmap_f, wu["fwd"], z.ref, m=wm)
            self.emit(capi.OP_BN_EVAL_STAT, i=[cout], f=[spec.BN_EPS],
                      p=[gamma, beta, self.R(pre + ".bn.running_mean"), self.R(pre + ".bn.running_var"), stat])
            g_apply = 1
        elif self.training:
            d = D.conv_fwd(x.N, x.thw, Ci, x.ld, cout, z.ld, k, stride, pf, othw, flags=capi.F_BNPART | (capi.F_CI3 if ci3 else 0), groups=self.groups)
            d["Ci_real"] = Ci_real
            # The trunk's forward convs stay on the fp32-MFMA kernel (PICONS_SPLIT_TRUNK_FWD=1 moves them to the bf16-split kernel): the numerics
            # gate said no.  Per kernel the split is the more accurate of the two, but EM routing amplifies ANY perturbation of the trunk's
            # activations into the trunk's gradients, and with these ~20 launches split one per-tensor bar of the full-size suite
            # (Mixed_4f.b2a.bn.bias, bs-8 JHMDB case) went from 0.80 to 1.06 of its bar; they were worth 0.05 ms.
            trunk_x6 = sw.exp("PICONS_SPLIT_TRUNK_FWD", "0", self.exp) != "0"
            if trunk_x6:
                d = self.maybe_x6(d, w["fwd"])     # the tile (hence the partial rows) is the bf16-split kernel's where that kernel runs
            nrows = _bnpart_rows(d)
            part = self.alloc(nrows * 2 * cout)
            self.conv_op(d, x.ref, w["fwd"], z.ref, bnpart=part, x6=trunk_x6)
            fin = self.bn_finalize_or_defer(nrows, cout, z, part, gamma, beta, pre, stat)
            g_apply = self.groups
        else:
            d = D.conv_fwd(x.N, x.thw, Ci, x.ld, cout, z.ld, k, stride, pf, othw, flags=capi.F_CI3 if ci3 else 0)
            d["Ci_real"] = Ci_real
            self.conv_op(d, x.ref, w["fwd"], z.ref)
            self.emit(capi.OP_BN_EVAL_STAT, i=[cout], f=[spec.BN_EPS],
                      p=[gamma, beta, self.R(pre + ".bn.running_mean"), self.R(pre + ".bn.running_var"), stat])
            g_apply = 1
        if fin is not None:           # finalize folded into the apply kernel (few partial rows: one dispatch fewer on the chain)
            self.emit(capi.OP_BN_FIN_APPLY, i=[fin["npg"], self.groups, cout, z.ld, y.ld, 1], l=[z.rows // self.groups, z.rows], f=[spec.BN_EPS, spec.BN_MOMENTUM],
                      p=[fin["part"], gamma, beta, self.R(pre + ".bn.running_mean"), self.R(pre + ".bn.running_var"), stat, z.ref, y.ref])
        else:
            self.emit(capi.OP_BN_APPLY, i=[z.ld, cout, g_apply, y.ld, 1], l=[z.rows], p=[z.ref, stat, y.ref])

        st = {}

        def bwd(part="all"):
            """part "A": BN backward + wgrad (touches only this unit's buffers); part "B": dgrad into the
            input's gradient (shared by every consumer of x, so Inception serialises it on lane 0)."""
            if part in ("all", "A"):
                dy = self.grad_of(y)
                dz = st["dz"] = self.tensor(x.N, othw, cout, pre + ".dz")
                ws = self.alloc(_bn_bwd_ws(z.rows, cout, self.groups))
                fused_bwd = 2 if sw.exp("PICONS_BN_FUSED", "0", self.exp) != "0" else 0       # the planner's decision travels in the op (bit 1 of `relu`)
                self.emit(capi.OP_BN_BWD, i=[dy.ld, z.ld, cout, self.groups, 1 | fused_bwd, dz.ld, self.acc], l=[z.rows],
                          p=[dy.ref, z.ref, stat, dz.ref, self.G(pre + ".bn.weight"), self.G(pre + ".bn.bias"), ws])
                if self.after_bn_bwd is not None:          # build_backward: the early Adam op leaves behind the stem's BatchNorm backward
                    hook, self.after_bn_bwd = self.after_bn_bwd, None
                    hook()
                wd = D.trim_wgrad(D.wgrad(x.N, othw, cout, dz.ld, x.thw, Ci, x.ld, k, stride, pf))
                wd["Cs_real"] = Ci_real
                wd["flags"] = capi.WG_CS3 if ci3 else 0
                self.on_wgrad_lane(lambda: (self.wgrad_op(wd, [dz.ref, x.ref, w["kg"]], w), self.flush_grad(w)))
                self.mark_final(*[q + sfx for q in pres for sfx in (".bn.weight", ".bn.bias")])
            if need_dx and part in ("all", "B"):
                dz = st["dz"]
                dx, acc = self.grad_for_write(x)
                self.alg_dgrad(F_fwd, wino)
                if wino:      # the input gradient of a stride-1 same-padded conv is the same correlation with mirrored, transposed weights
                    self.wino_op(dz.ref, x.N, x.thw, othw[0], cout, dz.ld, Ci, dx.ld, tmap_b, wu["tr"], dx.ref, flags=capi.F_ACCUM if acc else 0, m=wm_b)
                else:
                    for dd in D.transposed_classes(x.N, othw, cout, dz.ld, x.thw, Ci, dx.ld, k, stride, pf, flags=capi.F_ACCUM if acc else 0, ldw=cout):
                        self.conv_op(dd, dz.ref, w["tr"], dx.ref, alg=0)
        self.tape.append(bwd)
        return y

    def maxpool(self, x, k, s, name):
        othw = tuple(spec.same_out(x.thw[i], k[i], s[i]) for i in range(3))
        padf = [spec.same_pad(x.thw[i], k[i], s[i])[0] for i in range(3)]
        y = self.tensor(x.N, othw, x.C, name)
        am = self.alloc((y.rows * x.C + 3) // 4)
        pd = D.flatten(D.pool(x.N, x.thw, x.C, x.ld, othw, y.ld, k, s, padf), D.POOL_FIELDS)
        self.emit(capi.OP_POOL_FWD, i=pd, p=[x.ref, y.ref, am])

        def bwd():
            dy = self.grad_of(y)
            dx, acc = self.grad_for_write(x)
            pdb = D.flatten(D.pool(x.N, x.thw, x.C, dx.ld, othw, dy.ld, k, s, padf), D.POOL_FIELDS)
            self.emit(capi.OP_POOL_BWD, i=pdb + [int(acc)], p=[dy.ref, am, dx.ref])
        self.tape.append(bwd)
        return y

    def inception(self, pre, x, oc):
        """InceptionModule (pytorch_i3d.py:123-154): four branches off x, channel-concatenated (free here:
        each branch's last BN-apply writes its slice of `out`).  The branches share nothing but x, so a
        multi-lane plan puts each on its own stream, forward and backward."""
        c1, c2, c3 = oc[0], oc[0] + oc[2], oc[0] + oc[2] + oc[4]
        L = lambda j: j % self.branch_lanes
        one = (1, 1, 1)
        if self.fuse_1x1:
            # b1a, b2a and b0 (the 1x1x1 Unit3Ds reading x) as ONE conv + BN + ReLU.  Its output [b1a | b2a | b0] is
            # one contiguous channel range of a buffer that is `e` channels wider than the module output:
            # [b1a b2a | b0 b1b b2b b3b]; the module output is the slice behind the two intermediates (ld = wide)
            e = oc[1] + oc[3]
            wide = self.tensor(x.N, x.thw, e + c3 + oc[5], pre + ".out")
            out = wide.slice(e, c3 + oc[5])
            outer, self.tape = self.tape, []
            # the big 3x3x3 branch on lane 0, pool branch + the small 3x3x3 branch on lane 1 (measured best of four assignments);
            # in the forward the weight-gradient lane is idle and takes the small 3x3x3 branch (PICONS_FWD_BRANCH3=0: off)
            third = self.wg_lane if (self.wg_lane and sw.get("PICONS_FWD_BRANCH3", "1") != "0") else 0
            fmask = ((1 << self.branch_lanes) - 2) | (1 << third if third else 0)
            # the pool branch needs nothing but x: it leaves at the module's head, beside the fused 1x1x1 unit (round 6: forked behind that
            # unit it ended 40 - 60 us after lane 0 in every 14 x 14 module, the step's forward idling on the join)
            early = sw.get("PICONS_POOL_BRANCH_EARLY", "1") != "0" and self.branch_lanes > 1
            if early:
                self.fork(1 << L(1))
                self.lane = L(1)
                t3 = self.maxpool(x, (3, 3, 3), one, pre + ".pool")
                self.unit3d(pre + ".b3b", t3, oc[5], one, one, out=out.slice(c3, oc[5]))
                self.lane = 0
            self.unit3d([pre + ".b1a", pre + ".b2a", pre + ".b0"], x, [oc[1], oc[3], oc[0]], one, one, out=wide.slice(0, e + oc[0]))
            self.fork((1 << third) if (early and third) else fmask)
            self.lane = L(0)
            self.unit3d(pre + ".b1b", wide.slice(0, oc[1]), oc[2], (3, 3, 3), one, out=out.slice(c1, oc[2]))
            self.lane = L(1)
            if not early:
                t3 = self.maxpool(x, (3, 3, 3), one, pre + ".pool")
                self.unit3d(pre + ".b3b", t3, oc[5], one, one, out=out.slice(c3, oc[5]))
            if third:
                self.lane = third
            self.unit3d(pre + ".b2b", wide.slice(oc[1], oc[3]), oc[4], (3, 3, 3), one, out=out.slice(c2, oc[4]))
            self.lane = 0
            self.join(fmask)
            if early:
                (pool, b3b, fused, b1b, b2b), self.tape = self.tape, outer
            else:
                (fused, b1b, pool, b3b, b2b), self.tape = self.tape, outer

            def bwd_fused():
                self.wgrad_group_begin()  # the module's four weight gradients leave as one multi-problem launch
                self.fork()
                self.lane = L(0); b1b()
                self.lane = L(1); b2b(); b3b()
                self.lane = 0
                self.join()
                fused(); pool()          # the two writers of d(x), in order on lane 0
                self.wgrad_group_end()
            self.tape.append(bwd_fused)
            return out
        out = self.tensor(x.N, x.thw, oc[0] + oc[2] + oc[4] + oc[5], pre + ".out")
        outer, self.tape = self.tape, []
        self.fork()
        self.lane = L(0)
        t1 = self.unit3d(pre + ".b1a", x, oc[1], one, one)
        self.unit3d(pre + ".b1b", t1, oc[2], (3, 3, 3), one, out=out.slice(c1, oc[2]))
        self.lane = L(1)
        t2 = self.unit3d(pre + ".b2a", x, oc[3], one, one)
        self.unit3d(pre + ".b2b", t2, oc[4], (3, 3, 3), one, out=out.slice(c2, oc[4]))
        self.lane = L(2)
        t3 = self.maxpool(x, (3, 3, 3), one, pre + ".pool")
        self.unit3d(pre + ".b3b", t3, oc[5], one, one, out=out.slice(c3, oc[5]))
        self.lane = L(3)
        self.unit3d(pre + ".b0", x, oc[0], one, one, out=out.slice(0, oc[0]))
        self.lane = 0
        self.join()
        (b1a, b1b, b2a, b2b, pool, b3b, b0), self.tape = self.tape, outer

        def bwd():
            self.fork()
            self.lane = L(0); b1b(); b1a("A")
            self.lane = L(1); b2b(); b2a("A")
            self.lane = L(2); b3b()
            self.lane = L(3); b0("A")
            self.lane = 0
            self.join()
            # the four writers of d(x): first overwrites, the rest accumulate -> one lane, in order
            b0("B"); b1a("B"); b2a("B"); pool()
        self.tape.append(bwd)
        return out

    def conv_bias_act(self, wname, x, cout, k, pad, act, out, act_c0=0, bias_ref=None, wkey=None):
        """nn.Conv2d/Conv3d with bias (+activation) written into `out` (a channel slice)."""
        Ci = x.C
        othw = tuple(x.thw[i] + 2 * pad[i] - k[i] + 1 for i in range(3))
        w = self.kw[wkey or wname]
        d = D.conv_fwd(x.N, x.thw, Ci, x.ld, cout, out.ld, k, (1, 1, 1), pad, othw, act=act, flags=capi.F_BIAS)
        d["act_c0"] = act_c0
        self.conv_op(d, x.ref, w["fwd"], out.ref, bias=bias_ref)
        return othw

    def conv_layer(self, name, x, cout, k, pad, act, out, need_dx=True):
        """Decoder skip convs conv28/conv56/conv112 (capsules_ucf101.py:380-384,490,497,501)."""
        wino = act in (capi.ACT_NONE, capi.ACT_RELU) and self.wino_ok(x, cout, k, (1, 1, 1), pad)
        w = self.prep_conv_weight([name + ".weight"], [cout], x.C, k, need_dx, layouts=not wino)
        if wino:
            othw = tuple(x.thw)
            wm = self.wino_m(x)
            wu = self.wino_weights(name + ".weight", cout, x.C, need_dx and self.training, wm)
            F_fwd = _conv_flops(D.conv_fwd(x.N, x.thw, x.C, x.ld, cout, out.ld, k, (1, 1, 1), pad, othw))
            self.alg_flops_wino[self.cur] = self.alg_flops_wino.get(self.cur, 0) + F_fwd
            self.wino_op(x.ref, x.N, othw, x.thw[0], x.C, x.ld, cout, out.ld, (1, -1, 1), wu["fwd"], out.ref, bias=self.P(name + ".bias"), act=act, flags=capi.F_BIAS, m=wm)
        else:
            othw = self.conv_bias_act(name + ".weight", x, cout, k, pad, act, out, bias_ref=self.P(name + ".bias"))
            F_fwd = _conv_flops(D.conv_fwd(x.N, x.thw, x.C, x.ld, cout, out.ld, k, (1, 1, 1), pad, othw))

        def bwd():
            dy = self.grad_of(out)
            dz = self.tensor(x.N, othw, cout, name + ".dz")
            ws = self.alloc(_act_bwd_ws(out.rows, cout))
            self.emit(capi.OP_ACT_BWD, i=[dy.ld, out.ld, act, cout, dz.ld, self.acc], l=[out.rows],
                      p=[dy.ref, out.ref, dz.ref, self.G(name + ".bias"), ws])
            self.on_wgrad_lane(lambda: (self.wgrad_op(D.trim_wgrad(D.wgrad(x.N, othw, cout, dz.ld, x.thw, x.C, x.ld, k, (1, 1, 1), pad)), [dz.ref, x.ref, w["kg"]], w),
                                        self.flush_grad(w)))
            self.mark_final(name + ".bias")
            if need_dx:
                dx, acc = self.grad_for_write(x)
                self.alg_dgrad(F_fwd, wino)
                if wino:
                    self.wino_op(dz.ref, x.N, x.thw, othw[0], cout, dz.ld, x.C, dx.ld, (1, -1, 1), wu["tr"], dx.ref, flags=capi.F_ACCUM if acc else 0, m=wm)
                else:
                    for dd in D.transposed_classes(x.N, othw, cout, dz.ld, x.thw, x.C, dx.ld, k, (1, 1, 1), pad, flags=capi.F_ACCUM if acc else 0, ldw=cout):
                        self.conv_op(dd, dz.ref, w["tr"], dx.ref, alg=0)
        self.tape.append(bwd)

    def _skip_conv(self, name, x, cat):
        """Forward of a decoder skip conv on the skip lane, right behind its input; its backward closure is kept aside and
        re-enters the tape at the decoder position it always had (build_forward)."""
        self.fork(1 << self.skip_lane)
        self.lane = self.skip_lane
        self.conv_layer(name, x, 64, (3, 3, 3), (1, 1, 1), capi.ACT_RELU, cat.slice(64, 64))
        self.lane = 0
        self.skip_bwd[name] = self.tape.pop()

    def _on_skip_lane(self, fn):
        def g():
            if self.deferred is not None:         # held back until the EM backward has been enqueued (release_deferred)
                self.deferred.append(g)
                return
            self.fork(1 << self.skip_lane)        # behind the decoder gradient it consumes
            self.lane = self.skip_lane
            fn()
            self.flush_unprep()
            self.lane = 0
        return g

    def release_deferred(self):
        """The side-lane work held back during the decoder's backward (skip-conv backward, weight gradients) is enqueued now,
        behind everything lane 0 has enqueued so far.  Called right after the EM backward: that kernel needs whole CUs (154 KB of
        LDS per block), so side-lane kernels in flight when it starts stretch it from 0.83 to 1.5 ms and stall themselves; and the
        launches that leave CU slots idle are the trunk's, which come after it -- that is where the side lanes' work belongs."""
        held, self.deferred = self.deferred, None
        for fn in held or ():
            fn()

    def convT_layer(self, name, x, cout, k, stride, pad, opad, act, out, cscale=None):
        """nn.ConvTranspose2d/3d + bias (+ReLU) (+Dropout3d scale) into `out` (channel slice)."""
        Ci = x.C
        othw = tuple((x.thw[i] - 1) * stride[i] - 2 * pad[i] + k[i] + opad[i] for i in range(3))
        assert othw == tuple(out.thw), (name, othw, out.thw)
        w = self.prep_convT_weight(name + ".weight", Ci, cout, k)
        if (self.spectral_pc and cscale is None and x.thw[0] == 1 and k[0] == 1 and k[2] >= 7 and tuple(stride) == (1, 1, 1)
                and tuple(pad) == (0, 0, 0) and tuple(opad) == (0, 0, 0)):
            return self._convT_spectral(name, x, cout, k, act, out, w)
        flags = capi.F_BIAS | (capi.F_CSCALE if cscale is not None else 0)
        if cscale is None and max(k) >= 7 and tuple(stride) == (1, 1, 1) and self.groups * self.n == x.N:
            flags |= capi.F_NFAST          # 9x9 'full' conv (upsample1): most taps of a border patch are padding
        F_fwd = 0
        classes = list(D.transposed_classes(x.N, x.thw, Ci, x.ld, othw, cout, out.ld, k, stride, pad, act=act, flags=flags))
        side = list(range(1, self.lanes)) if (self.spread_classes and self.lane == 0 and self.cur == "fwd" and len(classes) >= 4) else []
        if side:
            # position classes write disjoint sub-lattices of `out`: heaviest first onto the least loaded lane
            mask = sum(1 << q for q in side)
            self.fork(mask, src=0)
            load = {q: 0 for q in [0] + side}
            classes.sort(key=lambda dd: -_conv_flops(dd))
        for dd in classes:
            F_fwd += _conv_flops(dd)
            if side:
                self.lane = min(load, key=lambda q: (load[q], q))
                load[self.lane] += _conv_flops(dd)
            self.conv_op(dd, x.ref, w["fwd"], out.ref, bias=self.P(name + ".bias"), cscale=cscale)
        if side:
            self.lane = 0
            self.join(mask)

        def bwd():
            dy = self.grad_of(out)
            if act != capi.ACT_NONE or cscale is not None:
                dz = self.tensor(x.N, othw, cout, name + ".dz")
            else:
                dz = dy
            ws = self.alloc(_act_bwd_ws(out.rows, cout))
            if cscale is not None:        # d(acc+bias) = dy * scale  (out itself is already scaled)
                assert act == capi.ACT_NONE
                self.emit(capi.OP_CHSCALE, i=[dy.ld, x.N, cout, dz.ld, 0], l=[out.rows // x.N], p=[dy.ref, cscale, dz.ref])
                self.emit(capi.OP_ACT_BWD, i=[dz.ld, dz.ld, capi.ACT_NONE, cout, dz.ld, self.acc], l=[out.rows],
                          p=[dz.ref, dz.ref, None, self.G(name + ".bias"), ws])
            else:
                self.emit(capi.OP_ACT_BWD, i=[dy.ld, out.ld, act, cout, dz.ld, self.acc], l=[out.rows],
                          p=[dy.ref, out.ref, dz.ref if act != capi.ACT_NONE else None, self.G(name + ".bias"), ws])
            self.on_wgrad_lane(lambda: (self.wgrad_op(D.trim_wgrad(D.wgrad(x.N, x.thw, Ci, x.ld, othw, cout, dz.ld, k, stride, pad)), [x.ref, dz.ref, w["kg"]], w),
                                        self.flush_grad(w)))
            self.mark_final(name + ".bias")
            dx, acc = self.grad_for_write(x)
            dd = D.conv_fwd(x.N, othw, cout, dz.ld, Ci, dx.ld, k, stride, pad, x.thw, flags=capi.F_ACCUM if acc else 0, ldw=cout)
            self.conv_op(dd, dz.ref, w["tr"], dx.ref, alg=F_fwd)
        self.tape.append(bwd)

    def _convT_spectral(self, name, x, cout, k, act, out, w):
        """Stride-1 9x9 ConvTranspose2d (upsample1) in row-spectral form (spectral.LayoutT): a full convolution along x is
        a product per frequency, what is left is one grouped transposed conv along y (a quarter of the direct FLOPs)."""
        Ci, KY, KX = x.C, k[1], k[2]
        SL = spectral.LayoutT(x.N, x.thw[1], x.thw[2], Ci, x.ld, cout, out.ld, KY, KX)
        sm = {kk: self.const(v) for kk, v in SL.matrices().items()}
        w["wv"] = self.alloc(SL.G * SL.w_g)                  # [g][Co][ky][Ci]  forward (transposed-form) GEMM weight planes
        w["wvt"] = self.alloc(SL.G * SL.w_g)                 # [g][Ci][ky][Co]  dgrad GEMM weight planes
        self.emit(capi.OP_WSPEC_FWD, i=[cout, Ci, KY, KX, SL.nu, SL.Ur], p=[w["fwd"], sm["tw"], w["wv"]], lst=self.prep_target)
        self.emit(capi.OP_WSPEC_FWD, i=[Ci, cout, KY, KX, SL.nu, SL.Ur], p=[w["tr"], sm["tw"], w["wvt"]], lst=self.prep_target)
        self.reg_weight(w["wv"], SL.G * SL.w_g, self.prep_target, 0)
        self.reg_weight(w["wvt"], SL.G * SL.w_g, self.prep_target, 0)
        xpl = self.alloc(SL.G * SL.x_g)
        tpl = self.alloc(SL.G * SL.t_g)
        self.emit(capi.OP_AXIS, i=D.flatten(SL.x_to_planes(), capi.AXIS_FIELDS), p=[x.ref, sm["F"], None, xpl])
        for dd in SL.convT():
            self.conv_op(dd, xpl, w["wv"], tpl)
        self.emit(capi.OP_AXIS, i=D.flatten(SL.planes_to_y(act, 0), capi.AXIS_FIELDS), p=[tpl, sm["G"], self.P(name + ".bias"), out.ref])
        othw = tuple(out.thw)

        def bwd():
            dy = self.grad_of(out)
            dz = self.tensor(x.N, othw, cout, name + ".dz") if act != capi.ACT_NONE else dy
            ws = self.alloc(_act_bwd_ws(out.rows, cout))
            self.emit(capi.OP_ACT_BWD, i=[dy.ld, out.ld, act, cout, dz.ld, self.acc], l=[out.rows],
                      p=[dy.ref, out.ref, dz.ref if act != capi.ACT_NONE else None, self.G(name + ".bias"), ws])
            dtpl = self.alloc(SL.G * SL.t_g)
            dwv = self.alloc(SL.G * SL.w_g)
            self.emit(capi.OP_AXIS, i=D.flatten(SL.dy_to_planes(dz.ld), capi.AXIS_FIELDS), p=[dz.ref, sm["Gt"], None, dtpl])
            self.on_wgrad_lane(lambda: (self.wgrad_op(SL.wgrad(), [xpl, dtpl, dwv]),
                                        self.emit(capi.OP_WSPEC_BWD, i=[Ci, cout, KY, KX, SL.nu, SL.Ur], p=[dwv, sm["tw"], w["kg"]]),
                                        self.flush_grad(w)))
            self.mark_final(name + ".bias")
            dx, acc = self.grad_for_write(x)
            dxpl = self.alloc(SL.G * SL.x_g)
            self.alg_dgrad(SL.flops())
            self.conv_op(SL.dgrad(), dtpl, w["wvt"], dxpl, alg=0)
            self.emit(capi.OP_AXIS, i=D.flatten(SL.planes_to_dx(dx.ld, acc), capi.AXIS_FIELDS), p=[dxpl, sm["Ft"], None, dx.ref])
        self.tape.append(bwd)

    # ------------------------------------------------------------------ whole model
    def build_forward(self):
        N, hw, C = self.N, self.hw, self.C
        self.cur = "fwd"
        T = spec.FRAMES
        # staged inputs (host copies into these before a run)
        self.in_data = self.alloc(self.n * 3 * T * hw * hw)
        self.in_aug = self.alloc(self.n * 3 * T * hw * hw)
        self.in_cls = self.alloc(N)
        self.in_labeled = self.alloc(N)
        self.in_drop832 = self.alloc(N * spec.TRUNK_OUT_CH)
        self.in_drop128 = self.alloc(N * 128)
        x = self.tensor(N, (T, hw, hw), 4, "img")
        self.img = x              # the clip as the first conv reads it ([2 bs][T][hw][hw][4]); StepEngine.sample_stager re-points its readers
        self.op_to_ndhwc = []
        for g in range(self.groups):
            self.op_to_ndhwc.append(len(self.lists["fwd"]))
            src = self.in_data if g == 0 else self.in_aug
            self.emit(capi.OP_TO_NDHWC, i=[0, self.n, 3, hw, 4, 0], l=[T * hw * hw], p=[src, off(x.ref, g * self.n * T * hw * hw * 4)])
        out56 = out112 = None
        first = True
        s28e = hw // 8
        if self.skip_lane:       # the decoder's concat buffers exist before the trunk so the skip convs can write into them early
            cat56 = self.tensor(N, (2, 2 * s28e, 2 * s28e), 128, "cat56")
            cat112 = self.tensor(N, (4, 4 * s28e, 4 * s28e), 128, "cat112")
        pending_skip, skip_after = [], sw.exp("PICONS_SKIP_FWD_AFTER", "", self.exp)
        for ent in spec.TRUNK:
            name = "conv1." + ent[0]
            if ent[0] == "Mixed_3b" and self.late_prep:
                self.join(1 << self.wg_lane)          # the weight layouts of everything from here on (list prep_late)
            if ent[1] == "conv":
                x = self.unit3d(name, x, ent[3], ent[4], ent[5], need_dx=not first)
                first = False
            elif ent[1] == "pool":
                x = self.maxpool(x, ent[2], ent[3], name)
            else:
                x = self.inception(name, x, ent[3])
            if ent[0] == "Conv3d_2c_3x3":
                out56 = x
                if self.skip_lane:
                    pending_skip.append(("conv56", out56, cat56))
            if ent[0] == "Conv3d_1a_7x7":
                out112 = x
                if self.skip_lane:
                    pending_skip.append(("conv112", out112, cat112))
            # the skip convs' forward: right behind their input (default), or held until the trunk entry PICONS_SKIP_FWD_AFTER has been emitted
            # (A/B switch: whole-CU Winograd blocks of conv112 / conv56 beside the chain's own big layers, or beside its small 28 x 28 ones)
            if pending_skip and (not skip_after or ent[0] == skip_after or ent is spec.TRUNK[-1]):
                for nm_, t_, c_ in pending_skip:
                    self._skip_conv(nm_, t_, c_)
                pending_skip = []
            if ent[0] == "MaxPool3d_3a_3x3" and self.skip_lane:
                # first trunk op whose backward touches a gradient the skip lane writes (d out56; d out112 comes later still)
                inner = self.tape.pop()
                self.tape.append(lambda inner=inner: (self.join(1 << self.skip_lane), inner()))
        self.named["trunk_out"] = x
        s28 = x.thw[1]
        # Dropout3d #1 (capsules_ucf101.py:428)
        if self.training:
            xd = self.tensor(N, x.thw, x.C, "x832d")
            self.emit(capi.OP_CHSCALE, i=[x.ld, N, x.C, xd.ld, 0], l=[x.rows // N], p=[x.ref, self.in_drop832, xd.ref])
            xin = x

            def bwd_drop():
                dxd = self.grad_of(xd)
                dx, acc = self.grad_for_write(xin)
                self.emit(capi.OP_CHSCALE, i=[dxd.ld, N, xin.C, dx.ld, int(acc)], l=[xin.rows // N], p=[dxd.ref, self.in_drop832, dx.ref])
            self.tape.append(bwd_drop)
        else:
            xd = x
        cat28 = self.tensor(N, (1, s28, s28), 128, "cat28")
        conv28_early = bool(self.conv28_aside and self.skip_lane)
        if conv28_early:          # conv28 reads nothing but xd: on the skip lane (idle since conv112), beside PrimaryCaps; joined in front of upsample2
            self.fork(1 << self.skip_lane)
            self.lane = self.skip_lane
            self.conv_layer("conv28", xd, 64, (1, 3, 3), (0, 1, 1), capi.ACT_RELU, cat28.slice(64, 64))
            self.lane = 0
            self.skip_bwd["conv28"] = self.tape.pop()
        # PrimaryCaps: pose (512) and activation (32, sigmoid) convs as one GEMM (capsules_ucf101.py:43-49)
        KP = spec.PRIMARY_K
        s20 = s28 - KP + 1
        npose = spec.IN_CAPS * spec.POSE
        caps_in = self.tensor(N, (1, s20, s20), npose + spec.IN_CAPS, "caps_in")
        spectral_pc = self.spectral_pc and xd.thw[0] == 1
        Cpc = npose + spec.IN_CAPS
        pc_names = [("primary_caps.pose.weight", 0, npose), ("primary_caps.a.weight", npose, spec.IN_CAPS)]
        if spectral_pc:
            # row-spectral form: DFT along x, one grouped 9x1 conv (3 real groups per complex frequency), inverse DFT
            # (spectral.py).  The weight planes come straight from the master OIHW tensors (no kernel-layout copies of the
            # 36.7 M-element weight), the gradient goes straight back.
            SL = spectral.Layout(N, xd.thw[1], xd.thw[2], xd.C, xd.ld, Cpc, caps_in.ld, KP, KP)
            sm = {k: self.const(v) for k, v in spectral.matrices(xd.thw[2], KP).items()}
            pl = self.next_prep_lane()
            nW = SL.G * SL.w_g
            x6_pc = self.x6 and sw.get("PICONS_SPLIT_SPECTRAL", "1") != "0" and all(capi.lib().pc_conv_x6_ok(_cdesc(dict(D.trim_conv(dd), flags=dd["flags"] | capi.F_X6))) for dd in [SL.conv()] + SL.dgrad())
            if x6_pc:
                # the 2 x 167 M-element weight planes leave the producer as bf16 terms (6 B per element): no fp32 copy exists, the
                # "fp32" references below are virtual addresses that only locate a group inside the planes
                wpc = dict(wv=("V", 0), wvt=("V", 4 * nW))
                pf, pt = self.alloc((3 * nW + 1) // 2), self.alloc((3 * nW + 1) // 2)
                self.wbufs.append(dict(ref=wpc["wv"], n=nW, lst=None, lane=None, planes=pf))
                self.wbufs.append(dict(ref=wpc["wvt"], n=nW, lst=None, lane=None, planes=pt))
                for nm, a0, cnt in pc_names:
                    self.emit(capi.OP_WSPEC_MASTER_PLANES, i=[cnt, a0, Cpc, xd.C, KP, KP, SL.nu, SL.Ur], l=[nW], p=[self.P(nm), sm["tw"], pf, pt],
                              lst=self.prep_target, lane=pl)
            else:
                wpc = dict(wv=self.alloc(nW),             # [g][Co][ky][Ci]  forward GEMM weight planes
                           wvt=self.alloc(nW))            # [g][Ci][ky][Co]  dgrad GEMM weight planes
                for nm, a0, cnt in pc_names:      # pose rows, then activation rows, of the same plane buffers: one lane
                    self.emit(capi.OP_WSPEC_MASTER_FWD, i=[cnt, a0, Cpc, xd.C, KP, KP, SL.nu, SL.Ur], p=[self.P(nm), sm["tw"], wpc["wv"], wpc["wvt"]],
                              lst=self.prep_target, lane=pl)
        else:
            wpc = self.prep_conv_weight([nm for nm, _a, _c in pc_names], [cnt for _n, _a, cnt in pc_names], xd.C, (1, KP, KP), True)
            wpc["tio"] = self.alloc(KP * KP * xd.C * Cpc)       # [tap][ci][co]: GEMM weights of the col2im dgrad
            self.emit(capi.OP_TRANSPOSE, i=[1, Cpc, KP * KP * xd.C, KP * KP * xd.C, Cpc, 0], l=[0, 0],
                      p=[wpc["fwd"], wpc["tio"]], lst=self.prep_target)
        pc_bias = self.alloc(npose + spec.IN_CAPS)
        self.emit(capi.OP_TRANSPOSE, i=[1, 1, npose, npose, 1, 0], l=[0, 0], p=[self.P("primary_caps.pose.bias"), pc_bias], lst=self.prep_target)
        self.emit(capi.OP_TRANSPOSE, i=[1, 1, spec.IN_CAPS, spec.IN_CAPS, 1, 0], l=[0, 0], p=[self.P("primary_caps.a.bias"), off(pc_bias, npose)], lst=self.prep_target)
        pc_dbias = self.alloc(npose + spec.IN_CAPS)
        if spectral_pc:
            xpl = self.alloc(SL.G * SL.x_g)
            tpl = self.alloc(SL.G * SL.t_g)
            self.emit(capi.OP_AXIS, i=D.flatten(SL.x_to_planes(), capi.AXIS_FIELDS), p=[xd.ref, sm["F"], None, xpl])
            self.conv_op(SL.conv(), xpl, wpc["wv"], tpl)
            self.emit(capi.OP_AXIS, i=D.flatten(SL.planes_to_y(capi.ACT_SIGMOID, npose), capi.AXIS_FIELDS), p=[tpl, sm["G"], pc_bias, caps_in.ref])
        else:
            self.conv_bias_act("", xd, Cpc, (1, KP, KP), (0, 0, 0), capi.ACT_SIGMOID, caps_in, act_c0=npose,
                               bias_ref=pc_bias, wkey="primary_caps.pose.weight")
        # ConvCaps EM routing (capsules_ucf101.py:290-331)
        npos = N * s20 * s20
        comb = self.tensor(N, (1, s20, s20), C * 17, "comb")
        Wc, bu, ba = self.P("conv_caps.weights"), self.P("conv_caps.beta_u"), self.P("conv_caps.beta_a")
        # training: the forward leaves its routing state (21 KB per position) for the backward, which then skips the recompute
        em_state = self.alloc(capi.lib().pc_em_state_floats(int(npos))) if self.training else None
        self.emit(capi.OP_EM_FWD, i=[npos, spec.IN_CAPS, C], p=[caps_in.ref, Wc, bu, ba, comb.ref, em_state])
        # class-capsule masking (capsules_ucf101.py:438-484)
        self.pred = self.alloc(N * C)
        cmask = self.alloc(N * C)
        masked = self.tensor(N, (1, s20, s20), C * 16, "masked")
        self.op_cmask = len(self.lists["fwd"])
        self.emit(capi.OP_CMASK_FWD, i=[N, s20 * s20, C, 0 if self.training else 2], p=[comb.ref, self.in_cls, self.in_labeled, self.pred, cmask, masked.ref])

        def bwd_caps():
            dmasked = self.grad_of(masked)
            dcomb = self.tensor(N, (1, s20, s20), C * 17, "d_comb")
            self.emit(capi.OP_CMASK_BWD, i=[N, s20 * s20, C], p=[dmasked.ref, self.dpred, cmask, dcomb.ref])
            dcaps = self.tensor(N, (1, s20, s20), npose + spec.IN_CAPS, "d_caps_in")
            ws = self.alloc(_em_ws(npos, spec.IN_CAPS, C))
            for nm in ("conv_caps.weights", "conv_caps.beta_u", "conv_caps.beta_a"):
                if not self.acc:
                    self.emit(capi.OP_FILL, p=[self.G(nm)], l=[int(np.prod(self.pshape[nm]))], f=[0.0])
            self.emit(capi.OP_EM_BWD, i=[npos, spec.IN_CAPS, C],
                      p=[caps_in.ref, Wc, bu, ba, dcomb.ref, dcaps.ref, self.G("conv_caps.weights"), self.G("conv_caps.beta_u"), self.G("conv_caps.beta_a"), ws, em_state])
            self.release_deferred()
            # primary caps backward: sigmoid on the activation channels, bias grads, wgrad, dgrad
            ws2 = self.alloc(_act_bwd_ws(caps_in.rows, npose))
            a_sl, da_sl = caps_in.slice(npose, spec.IN_CAPS), dcaps.slice(npose, spec.IN_CAPS)
            self.emit(capi.OP_ACT_BWD, i=[dcaps.ld, caps_in.ld, capi.ACT_SIGMOID, spec.IN_CAPS, dcaps.ld, self.acc], l=[caps_in.rows],
                      p=[da_sl.ref, a_sl.ref, da_sl.ref, self.G("primary_caps.a.bias"), ws2])
            self.emit(capi.OP_ACT_BWD, i=[dcaps.ld, caps_in.ld, capi.ACT_NONE, npose, dcaps.ld, self.acc], l=[caps_in.rows],
                      p=[dcaps.ref, caps_in.ref, None, self.G("primary_caps.pose.bias"), ws2])
            if spectral_pc:
                dtpl = self.alloc(SL.G * SL.t_g)
                dwv = self.alloc(SL.G * SL.w_g)
                self.emit(capi.OP_AXIS, i=D.flatten(SL.dy_to_planes(dcaps.ld), capi.AXIS_FIELDS), p=[dcaps.ref, sm["Gt"], None, dtpl])
                def pc_wgrad():
                    self.wgrad_op(SL.wgrad(), [dtpl, xpl, dwv])
                    for nm, a0, cnt in pc_names:
                        self.emit(capi.OP_WSPEC_MASTER_BWD, i=[cnt, a0, Cpc, xd.C, KP, KP, SL.nu, SL.Ur, self.acc], p=[dwv, sm["tw"], self.G(nm)])
                    self.mark_final(*[nm for nm, _a, _c in pc_names])
                self.on_wgrad_lane(pc_wgrad)
            else:
                self.wgrad_op(D.trim_wgrad(D.wgrad(N, caps_in.thw, caps_in.C, dcaps.ld, xd.thw, xd.C, xd.ld, (1, KP, KP), (1, 1, 1), (0, 0, 0))),
                              [dcaps.ref, xd.ref, wpc["kg"]], wpc)
            if not spectral_pc:
                self.flush_grad(wpc)
            self.mark_final("conv_caps.weights", "conv_caps.beta_u", "conv_caps.beta_a", "primary_caps.pose.bias", "primary_caps.a.bias")
            dx, acc = self.grad_for_write(xd)
            F_pc = 2 * caps_in.rows * caps_in.C * xd.C * KP * KP
            if spectral_pc:
                dxpl = self.alloc(SL.G * SL.x_g)
                self.alg_dgrad(SL.flops())
                for dd in SL.dgrad():
                    self.conv_op(dd, dtpl, wpc["wvt"], dxpl, alg=0)
                self.emit(capi.OP_AXIS, i=D.flatten(SL.planes_to_dx(dx.ld, acc), capi.AXIS_FIELDS), p=[dxpl, sm["Ft"], None, dx.ref])
            elif xd.thw[0] == 1 and caps_in.thw[1] + KP - 1 == xd.thw[1]:
                # exact 'full' correlation as GEMM + col2im: cols[o][(tap, ci)] = dcaps[o][:] . W[:, tap, ci] over the
                # 20x20 real output positions only (the gather form multiplies 28x28 positions x 81 taps: 2x the FLOPs)
                cols = self.alloc(caps_in.rows * KP * KP * xd.C)
                dg = D.conv_fwd(N, caps_in.thw, caps_in.C, dcaps.ld, KP * KP * xd.C, KP * KP * xd.C, (1, 1, 1), (1, 1, 1), (0, 0, 0),
                                caps_in.thw, ldw=caps_in.C)
                self.conv_op(dg, dcaps.ref, wpc["tio"], cols, alg=F_pc)
                self.emit(capi.OP_COL2IM, i=[N, caps_in.thw[1], caps_in.thw[2], KP, KP, xd.C, dx.ld, int(acc)], p=[cols, dx.ref])
            else:
                self.alg_dgrad(F_pc)
                for dd in D.transposed_classes(N, caps_in.thw, caps_in.C, dcaps.ld, xd.thw, xd.C, dx.ld, (1, KP, KP), (1, 1, 1), (0, 0, 0),
                                               flags=(capi.F_ACCUM if acc else 0) | capi.F_NFAST, ldw=caps_in.C):
                    self.conv_op(dd, dcaps.ref, wpc["tr"], dx.ref, alg=0)
        self.tape.append(bwd_caps)
        # decoder (capsules_ucf101.py:486-510)
        self.convT_layer("upsample1", masked, 64, (1, KP, KP), (1, 1, 1), (0, 0, 0), (0, 0, 0), capi.ACT_RELU, cat28.slice(0, 64))
        if conv28_early:
            self.tape.append(self.skip_bwd["conv28"])         # its backward stays where it was: lane 0, behind upsample2's
            self.join(1 << self.skip_lane)
        else:
            self.conv_layer("conv28", xd, 64, (1, 3, 3), (0, 1, 1), capi.ACT_RELU, cat28.slice(64, 64))
        if not self.skip_lane:
            cat56 = self.tensor(N, (2, 2 * s28, 2 * s28), 128, "cat56")
        self.convT_layer("upsample2", cat28, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), capi.ACT_RELU, cat56.slice(0, 64))
        if self.skip_lane:
            self.tape.append(self._on_skip_lane(self.skip_bwd["conv56"]))
            self.join(1 << self.skip_lane)        # conv56 / conv112 forward (enqueued behind the stem) before the decoder reads them
        else:
            self.conv_layer("conv56", out56, 64, (3, 3, 3), (1, 1, 1), capi.ACT_RELU, cat56.slice(64, 64))
            cat112 = self.tensor(N, (4, 4 * s28, 4 * s28), 128, "cat112")
        self.convT_layer("upsample3", cat56, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), capi.ACT_RELU, cat112.slice(0, 64))
        if self.skip_lane:
            self.tape.append(self._on_skip_lane(self.skip_bwd["conv112"]))
        else:
            self.conv_layer("conv112", out112, 64, (3, 3, 3), (1, 1, 1), capi.ACT_RELU, cat112.slice(64, 64))
        # upsample4 -> Dropout3d -> smooth are linear: collapsed into ONE 128->27 transposed conv with per-sample combined
        # weights + the 27-tap shifted sum (csrc/tail.hip; capsules_ucf101.py:504-509).  u4 (205 MB/clip) never exists.
        othw = (8, 8 * s28, 8 * s28)
        k3, s2, p1 = (3, 3, 3), (2, 2, 2), (1, 1, 1)
        J = 27
        wt = self.alloc(N * 128 * 27 * 32)          # [n][ci][tap][32]  (dgrad layout)
        wf = self.alloc(N * 32 * 27 * 128)          # [n][32][tap][ci]  (forward layout)
        bc = self.alloc(N * 32)
        cs_ref = self.in_drop128 if self.training else None
        W4, b4, Wp = self.P("upsample4.weight"), self.P("upsample4.bias"), self.P("smooth.weight")
        self.emit(capi.OP_TAIL_COMBINE, i=[N, 128, 128, 27, J], p=[W4, b4, cs_ref, Wp, wt, wf, bc])
        if self.merged_tail:
            return self._merged_tail(cat112, othw, wf, bc, cs_ref, W4, b4, Wp, J)
        proj = self.tensor(N, othw, 32, "proj")
        F_tail = 0
        for dd in D.transposed_classes(N, cat112.thw, 128, cat112.ld, othw, 32, 32, k3, s2, p1, flags=capi.F_BIAS, groups=N):
            dd["wgstride"] = 32 * 27 * 128
            dd["bgstride"] = 32
            dd["Co_real"] = J
            F_tail += _conv_flops(dd)
            self.conv_op(dd, cat112.ref, wf, proj.ref, bias=bc)
        out = self.tensor(N, othw, 1, "out")
        self.emit(capi.OP_TAPSUM_FWD, i=[N, othw[0], othw[1], othw[2]], p=[proj.ref, self.P("smooth.bias"), out.ref])
        self.out = out

        def bwd_smooth():
            dproj = self.tensor(N, othw, 32, "d_proj")
            self.emit(capi.OP_TAPSUM_BWD, i=[N, othw[0], othw[1], othw[2]], p=[self.dout, dproj.ref])
            sums = self.alloc(N * 32)
            per_n = othw[0] * othw[1] * othw[2]
            self.emit(capi.OP_TAIL_COLSUM, i=[N], l=[per_n], p=[dproj.ref, sums])
            Gc = self.alloc(N * 128 * 27 * 32)
            self.emit(capi.OP_FILL, p=[Gc], l=[N * 128 * 27 * 32], f=[0.0])
            xin_per = cat112.thw[0] * cat112.thw[1] * cat112.thw[2] * cat112.ld
            # per-sample dWc[n][ci][tap][j] = sum_i x[n,i,ci] * dproj[n, 2i-1+k, j]: N problems in one launch
            wd = D.trim_wgrad(D.wgrad(1, cat112.thw, 128, cat112.ld, othw, 32, 32, k3, s2, p1))
            wd.update(nbatch=N, dbstride=xin_per, sbstride=per_n * 32, gbstride=128 * 27 * 32, Cs_real=J)
            self.wgrad_op(wd, [cat112.ref, dproj.ref, Gc])
            self.emit(capi.OP_TAIL_GRADS, i=[N, 128, 128, 27, J, 13, self.acc],
                      p=[Gc, sums, W4, b4, cs_ref, Wp, self.G("upsample4.weight"), self.G("upsample4.bias"), self.G("smooth.weight"), self.G("smooth.bias")])
            self.mark_final("upsample4.weight", "upsample4.bias", "smooth.weight", "smooth.bias")
            dx, acc = self.grad_for_write(cat112)
            dd = D.conv_fwd(N, othw, 32, 32, 128, dx.ld, k3, s2, p1, cat112.thw, flags=capi.F_ACCUM if acc else 0, groups=N, ldw=32)
            dd["wgstride"] = 128 * 27 * 32
            dd["Ci_real"] = J
            self.conv_op(dd, dproj.ref, wt, dx.ref, alg=F_tail)
        self.tape.append(bwd_smooth)
        return out

    def _merged_tail(self, cat112, othw, wf, bc, cs_ref, W4, b4, Wp, J):
        """upsample4 -> Dropout3d -> smooth as ONE five-tap stride-2 transposed conv with a single output channel
        (csrc/tail6.hip): per clip-pass 128 -> 125 column GEMMs over the 8 classes of the 4x112x112 input positions and a
        gather of the <= 27 column entries that land on each of the 8x224x224 outputs; 26 GFLOP per pass instead of 177."""
        N = cat112.N
        It, Ih, Iw = cat112.thw
        SP = tail6.SP
        w5f = self.alloc(N * 8 * SP * 128)              # [n][z][slot][ci]   forward GEMM weights
        w5t = self.alloc(N * 8 * 128 * SP)              # [n][z][ci][slot]   dgrad GEMM weights
        self.emit(capi.OP_TAIL6_WEIGHTS, i=[N, 128], p=[wf, w5f, w5t])
        self.reg_weight(w5f, N * 8 * SP * 128)
        self.reg_weight(w5t, N * 8 * 128 * SP)
        if self.x6:               # the planes are read on two lanes (interior class / border classes): made here, in front of the FORK
            self.planes_of(w5f)
            self.planes_of(w5t)
        cols = self.alloc(cat112.rows * SP)
        F_t6 = 2 * cat112.rows * 125 * 128
        # the class of the interior positions (z = 0) is 74 % of the work; the seven thin border classes are latency-bound
        # launches and run beside it on lane 1 (each class writes its own sub-lattice of cols)
        self.fork()
        for q, (z, d) in enumerate(tail6.conv_descs(N, cat112.thw, 128, cat112.ld)):
            self.lane = 0 if z == 0 else 1 % self.branch_lanes
            self.conv_op(d, cat112.ref, off(w5f, z * SP * 128), cols, alg=F_t6 if q == 0 else 0)
        self.lane = 0
        self.join()
        out = self.tensor(N, othw, 1, "out")
        self.emit(capi.OP_TAIL6_GATHER, i=[N, It, Ih, Iw], p=[cols, bc, self.P("smooth.bias"), out.ref])
        self.out = out

        def bwd_smooth():
            dcols = self.alloc(cat112.rows * SP)
            self.emit(capi.OP_TAIL6_SCATTER, i=[N, It, Ih, Iw], p=[self.dout, dcols])
            sums = self.alloc(N * 32)
            # ordered: per-block partial rows added in block order, K-slice images per class added in slice order -- no fp32 atomics
            # anywhere in the tail's gradients (round 6; with upsample4 / smooth's own reduction in pc_tail_grads)
            sums_ws = self.alloc(capi.lib().pc_tail6_bias_sums_ws_floats(N, It, Ih, Iw)) if self.wg_ordered else 0
            self.emit(capi.OP_TAIL6_BIAS_SUMS, i=[N, It, Ih, Iw], p=[self.dout, sums, sums_ws])
            wds = tail6.wgrad_descs(N, cat112.thw, 128, cat112.ld, compact=self.wg_ordered)
            ns8, zoff = [0] * 8, {}
            if self.wg_ordered:
                at = 0
                for z, wd in wds:
                    if self.x6 and sw.get("PICONS_SPLIT_WGRAD", "1") != "0":
                        wd["flags"] = int(wd.get("flags", 0)) | capi.WG_X6           # as wgrad_op routes it: the slice count is the routed kernel's
                    ns8[z] = capi.lib().pc_wgrad_slices(_wdesc(wd))
                    if ns8[z] < 1:
                        raise RuntimeError("pc_wgrad_slices: %s" % capi.lib().pc_last_error().decode())
                    wd["ws_slices"] = ns8[z]
                    zoff[z] = at
                    at += ns8[z] * N * 128 * SP
                dW5 = self.alloc(at)
                self.zero_once.append((dW5, at))            # padding slots and empty slices are never written
            else:
                dW5 = self.alloc(N * 8 * 128 * SP)
            Gc = self.alloc(N * 128 * 27 * 32)
            dx, acc = self.grad_for_write(cat112)

            def weight_grads(lane_of):
                """The tail's weight gradients: per-class wgrads, their map onto the 27x27 combined weights, upsample4 / smooth."""
                if not self.wg_ordered:
                    self.emit(capi.OP_FILL, p=[dW5], l=[N * 8 * 128 * SP], f=[0.0])
                grouped = bool(self.wg_lane) and self.wgrad_collect is None
                if grouped:
                    self.wgrad_collect = []
                for z, wd in wds:
                    self.lane = lane_of(z)
                    self.wgrad_op(wd, [cat112.ref, dcols, off(dW5, zoff[z] if self.wg_ordered else z * 128 * SP)])
                if grouped:              # the eight position classes (seven of them a few hundred positions) in one launch
                    jobs, self.wgrad_collect = self.wgrad_collect, None
                    self.wjobs = getattr(self, "wjobs", [])
                    self.wjobs.append(jobs)
                    if self.wg_ordered or sw.exp("PICONS_WGRAD_MULTI_TAIL", "0", self.exp) == "0":
                        for d_, p_ in jobs:
                            self._emit_wgrad(d_, p_)
                    else:
                        self.emit(capi.OP_WGRAD_MULTI, i=[len(jobs)], p=[("WJOBS", len(self.wjobs) - 1)])

            def weight_grads_tail():
                self.emit(capi.OP_TAIL6_WGRAD_MAP, i=[N, 128, int(self.wg_ordered)] + ns8, p=[dW5, Gc])
                self.emit(capi.OP_TAIL_GRADS, i=[N, 128, 128, 27, J, 13, self.acc],
                          p=[Gc, sums, W4, b4, cs_ref, Wp, self.G("upsample4.weight"), self.G("upsample4.bias"), self.G("smooth.weight"), self.G("smooth.bias"), grads_ws])
                self.mark_final("upsample4.weight", "upsample4.bias", "smooth.weight", "smooth.bias")
            grads_ws = self.alloc(capi.lib().pc_tail_grads_ws_floats(N, 128, 128)) if self.wg_ordered else 0
            if self.wg_lane:
                # nothing but the optimiser waits for them: the whole group goes to the weight-gradient lane, and lane 0's join
                # below only waits for the seven thin border classes' dgrads
                wl = self.wg_lane
                self.on_wgrad_lane(lambda: (weight_grads(lambda z: wl), weight_grads_tail()))
            self.fork()
            if not self.wg_lane:
                weight_grads(lambda z: 0 if z == 0 else 1 % self.branch_lanes)
            for q, (z, dd) in enumerate(tail6.dgrad_descs(N, cat112.thw, 128, dx.ld, acc)):
                self.lane = 0 if z == 0 else 1 % self.branch_lanes
                self.conv_op(dd, dcols, off(w5t, z * 128 * SP), dx.ref, alg=F_t6 if q == 0 else 0)
            self.lane = 0
            self.join()
            if not self.wg_lane:
                weight_grads_tail()
        self.tape.append(bwd_smooth)
        return out

    def build_loss(self, args):
        """Spread loss on pass-0 labeled rows + fused consistency/supervised loss; seeds dout/dpred."""
        self.cur = "loss"
        N, n, hw, C = self.N, self.n, self.hw, self.C
        T = spec.FRAMES
        per = T * hw * hw
        self.in_seg = self.alloc(n * per)
        self.seg32 = self.in_seg
        self.dout = self.alloc(N * per)
        self.dpred = self.alloc(N * C)
        self.scalars = self.alloc(32)                 # [0,16) consistency / supervised scalars, [16,20) spread loss: one D2H reads both
        self.spread_out = off(self.scalars, 16)
        self.emit(capi.OP_FILL, p=[self.dpred], l=[N * C], f=[0.0])
        self.op_spread = len(self.lists["loss"])
        self.emit(capi.OP_SPREAD, i=[n, C], f=[0.2, float(args.wt_cls)], p=[self.pred, self.in_cls, self.in_labeled, self.spread_out, self.dpred])
        ld = dict(B=n, T=T, H=hw, W=hw)
        ws = self.alloc(_loss_ws(n, hw))
        self.op_loss = len(self.lists["loss"])
        lower = -1.0 if args.lower_thresh is None else float(args.lower_thresh)
        upper = -1.0 if args.upper_thresh is None else float(args.upper_thresh)
        self.emit(capi.OP_LOSS, i=[n, T, hw, hw, int(args.bv), int(args.gv), int(args.n_frames), int(args.predict_maps), int(self.jhmdb)],
                  f=[lower, upper, float(args.bv_wt), float(args.gv_wt), float(args.wt_loc), float(args.wt_cons), 0.0],
                  p=[self.out.ref, off(self.out.ref, n * per), self.seg32, self.in_labeled, self.scalars, self.dout, off(self.dout, n * per), None, None, ws])

    def build_seeds(self):
        """Gradient seeds for the nn.Module path (no fused loss): autograd hands d(out_1), d(actor_prediction)."""
        per = spec.FRAMES * self.hw * self.hw
        self.dout = self.alloc(self.N * per)
        self.dpred = self.alloc(self.N * self.C)

    def build_backward(self):
        self.cur = "bwd"
        if not self.wg_ordered:       # the atomic split-K form adds into the kernel-layout gradients; the ordered form writes K-slice images (plain stores)
            self.emit(capi.OP_FILL, p=[self.kg_base], l=[self.kg_used // 4], f=[0.0])
        if (self.wg_lane or self.skip_lane) and sw.exp("PICONS_DEFER_SIDE", "0", self.exp) != "0":
            self.deferred = []
        for idx, fn in enumerate(reversed(self.tape)):
            if self.early_adam and idx == len(self.tape) - 1:
                # The early Adam op (HBM-bound, 0.1 - 0.2 ms alone) used to leave IN FRONT of the stem's backward and ran beside its BatchNorm
                # backward, HBM-bound too: 0.39 ms for the pair, on the step's serial tail.  Behind the BatchNorm backward it runs beside the stem's
                # weight gradient instead, which is bound by the matrix cores (PICONS_EARLY_ADAM_BEHIND_BN=0: in front, as rounds 3 - 5).
                if sw.get("PICONS_EARLY_ADAM_BEHIND_BN", "1") != "0":
                    self.after_bn_bwd = self._emit_early_adam
                else:
                    self._emit_early_adam()
            fn()
            assert self.after_bn_bwd is None, "the last backward closure has no BatchNorm backward to hang the early Adam op behind"
            self.flush_unprep()
        self.release_deferred()
        self.flush_unprep()
        if self.wg_lane or self.skip_lane:
            self.join((1 << self.wg_lane | 1 << self.skip_lane) & ~1)

    def _emit_adam_op(self, n_lo):
        lane = self.skip_lane or 1
        for src in range(self.lanes):
            if src != lane:
                self.fork(1 << lane, src=src)
        at = len(self.lists["bwd"])
        self.emit(capi.OP_ADAM, i=[1], f=[1e-4, 0.9, 0.999, 1e-6, 1.0], l=[0], p=[("P", 4 * n_lo), ("G", 4 * n_lo), ("M", 4 * n_lo), ("V", 4 * n_lo)], lane=lane)
        if lane != (self.skip_lane or self.wg_lane):
            self.join(1 << lane)
        return at

    def _emit_early_adam(self):
        """In front of the LAST backward closure (the stem unit: the first three tensors of the flat parameter buffer): every other
        gradient is final once what has been enqueued so far, on any lane, has run.  One Adam op over [adam_split, nparams) on the skip
        lane (idle by now), ordered behind every lane; length 0 until the caller arms it."""
        names = sorted(self.pshape, key=self.poff.get)
        stem = [nm for nm in names if nm.startswith(spec.trunk_units()[0][0] + ".")]
        assert names[:len(stem)] == stem and len(stem) == 3, "the stem's parameters lead the flat buffer"
        assert all(nm in self.final_at for nm in names[len(stem):]), "a gradient outside the stem is finalised by the stem's backward"
        n0 = self.poff[names[len(stem)]]
        self.adam_split = n0
        self.op_adam_early = self._emit_adam_op(n0)

    def grad_buckets(self, target_floats=12_000_000, joined=False):
        """Gradient all-reduce schedule for data parallelism: contiguous ranges of the flat G buffer in the
        order their gradients become final during backward, each with the number of bwd ops that must have been ENQUEUED
        before it may be reduced.  -> [(ready_after_bwd_ops, start_float, end_float)] sorted by readiness.

        Default: a bucket is ready once the last op that writes one of its gradients has been enqueued, on whatever lane -- the
        caller must put the collective behind EVERY lane's stream (GradReducer.launch(i, streams) does).  With the weight
        gradients on a lane of their own that is joined only at the end of the list, this is what lets a bucket leave while
        the backward is still running.  joined=True: ready only where lane 0 has JOINed the lane that finalised it (a caller
        that orders the collective behind lane 0 alone); with a weight-gradient lane that is the end of the list."""
        missing = [k for k in self.pshape if k not in self.final_at]
        if missing:
            raise RuntimeError("no backward op finalises %s" % missing[:4])
        names = sorted(self.pshape, key=self.poff.get)       # flat-buffer order
        # a gradient finalised by an op on a side lane is ready once lane 0 has JOINed that lane; one finalised on lane 0 inside a
        # branch region is snapped to the region's JOIN as well (its neighbours in the bucket live on the side lanes)
        bwd = self.lists["bwd"]
        next_join = {}                                   # lane -> index + 1 of the next JOIN covering it, per op index (filled backwards)
        cover = [dict() for _ in range(len(bwd) + 1)]
        cur = {}
        for idx in range(len(bwd) - 1, -1, -1):
            if bwd[idx][0] == capi.OP_JOIN:
                for q in range(1, self.lanes):
                    if (bwd[idx][1][0] >> q) & 1:
                        cur[q] = idx + 1
            cover[idx] = dict(cur)
        branch_mask = (1 << self.branch_lanes) - 2
        snap, open_at = list(range(len(bwd) + 1)), None
        for idx, op in enumerate(bwd):
            if op[0] == capi.OP_FORK and op[1][0] == branch_mask:
                open_at = idx
            elif op[0] == capi.OP_JOIN and op[1][0] == branch_mask and open_at is not None:
                for r in range(open_at + 1, idx + 1):
                    snap[r] = idx + 1
                open_at = None
        if open_at is not None:
            raise RuntimeError("backward list ends with an open FORK")

        def ready_of(nm):
            k, lane = self.final_at[nm], self.final_lane.get(nm, 0)
            if not joined:
                return max(1, min(k, len(bwd)))
            if lane == 0:
                return snap[k]
            j = cover[min(k, len(bwd))].get(lane) if k < len(bwd) else None
            if j is None:
                raise RuntimeError("gradient of %s is finalised on lane %d and never joined" % (nm, lane))
            return j
        buckets, cur_end, cur_ready, cur_size = [], self.nparams, 0, 0
        for nm in reversed(names):              # backward finalises parameters roughly in reverse flat order
            cur_ready = max(cur_ready, ready_of(nm))
            cur_size = cur_end - self.poff[nm]
            if cur_size >= target_floats:
                buckets.append((cur_ready, self.poff[nm], cur_end))
                cur_end, cur_ready = self.poff[nm], 0
        if cur_end > 0:
            buckets.append((max(cur_ready, max(ready_of(nm) for nm in names) if cur_ready == 0 else cur_ready), 0, cur_end))
        buckets.sort(key=lambda b: b[0])
        return buckets

    def build_adam(self):
        self.op_adam = 0
        self.emit(capi.OP_ADAM, i=[1], f=[1e-4, 0.9, 0.999, 1e-6, 1.0], l=[self.nparams], p=[("P", 0), ("G", 0), ("M", 0), ("V", 0)], lst="adam")

    # ------------------------------------------------------------------ finalisation
    def finalize(self):
        """Multi-lane plans: wrap the weight-layout prep in one FORK..JOIN (idempotent).  Ops that read master
        parameters are independent across weights; second-level layouts (made from a prepared buffer) run on
        lane 0 after the join."""
        lst = self.lists["prep"]
        if self.lanes > 1 and lst and lst[0][0] != capi.OP_FORK:
            first = lambda op: op[0] == capi.OP_FILL or op[3][0][0] == "P"
            lvl0 = [op for op in lst if first(op)]
            lvl1 = [op[:5] + (0,) for op in lst if not first(op)]
            mask = [(1 << self.lanes) - 2]
            lst[:] = [(capi.OP_FORK, mask, [], [], [], 0)] + lvl0 + [(capi.OP_JOIN, mask, [], [], [], 0)] + lvl1
        late = self.lists["prep_late"]
        if late and late[0][0] != capi.OP_FORK:
            # the skip convs' own weights are laid out on the skip lane, in front of the convs that read them; everything else on the
            # weight-gradient lane (idle during the forward), joined by the forward list before Mixed_3b
            mine = set()
            for nm in ("conv56.weight", "conv112.weight"):
                if nm in self.kw:
                    mine |= {self.kw[nm].get("fwd"), self.kw[nm].get("tr"), self.kw[nm].get("wino_fwd"), self.kw[nm].get("wino_tr")}
            mine.discard(None)
            lane_of = lambda op: self.skip_lane if (self.skip_lane and any(r in mine for r in op[3] if r is not None)) else self.wg_lane
            late[:] = [(capi.OP_FORK, [(1 << self.wg_lane | 1 << self.skip_lane) & ~1, 0], [], [], [], 0)] + [op[:5] + (lane_of(op),) for op in late]

    def resolve(self, bases):
        """-> dict list-name -> numpy array of capi.OP_DTYPE with absolute device pointers."""
        self.finalize()
        out = {}
        keep = out["_tjobs"] = []          # host job tables of the TRANSPOSE_MULTI ops (the list's owner keeps `out` alive)
        for name, lst in self.lists.items():
            if name in ("prep", "prep_late"):
                lst = self._merge_prep_transposes(lst, bases, keep)
            arr = np.zeros(len(lst), dtype=capi.OP_DTYPE)
            for j, (kind, i, f, p, l, lane) in enumerate(lst):
                arr[j]["kind"] = kind
                arr[j]["lane"] = lane
                arr[j]["i"][:len(i)] = i
                arr[j]["f"][:len(f)] = f
                arr[j]["l"][:len(l)] = l
                for q, r in enumerate(p):
                    if r is not None and r[0] == "JOBS":
                        tab = self._job_table(self.multi_jobs[r[1]], bases)
                        keep.append(tab)
                        r = ("HOST", tab.ctypes.data)
                    if r is not None and r[0] == "SJOBS":
                        tab = np.zeros(len(r[1]), dtype=capi.SJOB_DTYPE)
                        for w, (src, dst, n_, ps) in enumerate(r[1]):
                            tab[w] = (bases[src[0]] + src[1], bases[dst[0]] + dst[1], n_, ps)
                        keep.append(tab)
                        r = ("HOST", tab.ctypes.data)
                    if r is not None and r[0] == "WJOBS":
                        tab = np.zeros(len(self.wjobs[r[1]]), dtype=capi.WJOB_DTYPE)
                        for w, (wd, wp) in enumerate(self.wjobs[r[1]]):
                            tab[w]["d"][:] = D.flatten(wd, D.WGRAD_FIELDS)
                            tab[w]["D"], tab[w]["S"], tab[w]["g"] = [bases[x[0]] + x[1] for x in wp]
                        keep.append(tab)
                        r = ("HOST", tab.ctypes.data)
                    arr[j]["p"][q] = 0 if r is None else (r[1] if r[0] == "HOST" else bases[r[0]] + r[1])
            out[name] = arr
        return out

    @staticmethod
    def _job_table(jobs, bases):
        """OP_TRANSPOSE tuples -> host array of pc_transpose_job with absolute pointers."""
        tab = np.zeros(len(jobs), dtype=capi.TJOB_DTYPE)
        for q, op in enumerate(jobs):
            _kind, i, _f, p, l = op[:5]
            ns, sst = (i[6], l[2]) if len(i) > 6 else (0, 0)          # K-slice images of an ordered split-K weight gradient
            tab[q] = (bases[p[0][0]] + p[0][1], bases[p[1][0]] + p[1][1], l[0], l[1], i[0], i[1], i[2], i[3], i[4], i[5], ns, 0, sst)
        return tab

    def _merge_prep_transposes(self, lst, bases, keep):
        """The weight re-layouts that read master parameters are independent of each other: all of them on one lane
        become ONE launch (pc_transpose_multi, jobs passed by value) instead of ~80 seven-microsecond launches."""
        res, pending, splits = [], {}, {}

        def flush():
            for lane in sorted(pending):
                jobs = pending[lane]
                tab = self._job_table(jobs, bases)
                keep.append(tab)
                res.append((capi.OP_TRANSPOSE_MULTI, [len(jobs)], [], [("HOST", tab.ctypes.data)], [], lane))
            pending.clear()
            for lane in sorted(splits):          # the bf16 planes of the buffers those transposes filled: one launch per lane, behind them
                res.append((capi.OP_SPLIT_PLANES_MULTI, [len(splits[lane])], [], [("SJOBS", splits[lane])], [], lane))
            splits.clear()
        for op in lst:
            if op[0] == capi.OP_SPLIT_PLANES:
                splits.setdefault(op[5], []).append((op[3][0], op[3][1], op[4][0], op[4][1]))
            elif op[0] == capi.OP_TRANSPOSE and op[3][0][0] == "P":
                pending.setdefault(op[5], []).append(op)
            elif op[0] in (capi.OP_FILL, capi.OP_FORK, capi.OP_WSPEC_MASTER_FWD, capi.OP_WSPEC_MASTER_PLANES, capi.OP_WINO_WEIGHTS):
                res.append(op)           # fills precede the transposes into their buffer; the fork opens the region; the
                                         # master-layout weight planes touch nothing the transposes do
            else:
                flush()
                res.append(op)
        flush()
        return res

    def conv_flops_executed(self):
        """Per list, FLOPs of the conv / dgrad launches as the kernels run them (pc_conv_work: the host walks every launch's tiles
        in the kernel's row order): `mfma` = issued to the matrix cores (whole tiles, what an MFMA instruction counter sees),
        `executed` = on real output rows x columns over the K each tile's block walks -- taps that are padding for a whole tile are
        skipped by the kernel and are NOT counted, padding taps inside a tile's tap box are --, `valid` = non-padding MACs only."""
        return {name: {k: self.work.get((name, k), 0) for k in ("mfma", "executed", "valid")} for name in self.lists}

    def x6_flops_executed(self):
        """The same three counts for the conv / dgrad launches that run on the bf16-split kernel (fp32-equivalent multiply-accumulates x 2)."""
        return {name: {k: self.work.get((name, "x6_" + k), 0) for k in ("mfma", "executed", "valid")} for name in self.lists}

    def wgrad_flops_executed(self):
        """The same three counts for the weight-gradient launches (pc_wgrad_work)."""
        return {name: {k: self.work.get((name, "wg_" + k), 0) for k in ("mfma", "executed", "valid")} for name in self.lists}

    def wgrad_flops_by_family(self):
        """(bf16-split launches, fp32-MFMA launches): the counts of wgrad_flops_executed summed over the lists, per kernel family."""
        return tuple({k: sum(self.work.get((name, pre + k), 0) for name in self.lists) for k in ("mfma", "executed", "valid")} for pre in ("wgx6_", "wgf32_"))

    def flops(self, only_kind=None):
        """FLOPs ISSUED per list (2*M*N*K of the emitted, tap-trimmed descriptors with the real channel counts -- zero-padding
        taps dropped at descriptor level and padded channels are not work).  only_kind: OP_CONV / OP_WGRAD / None (both)."""
        kinds = (capi.OP_CONV, capi.OP_WGRAD) if only_kind is None else (only_kind,)
        return {name: sum(self.issued.get((name, k), 0) for k in kinds) for name in self.lists}

    def flops_reference_counted(self):
        """Conv / dgrad FLOPs per list as the layers' own formulation counts them (all taps incl. padding; every dgrad at its
        layer's forward FLOPs).  Kept beside flops() so the two roofline fractions can be told apart."""
        return {name: self.alg_flops.get(name, 0) for name in self.lists}

    def flops_reference_counted_wino(self):
        """The same count for the layers that run in Winograd form: what their direct 3x3x3 formulation multiplies (2.25x the transform-domain work)."""
        return {name: self.alg_flops_wino.get(name, 0) for name in self.lists}


def _conv_flops(d):
    """2*M*N*K of a conv descriptor with the real channel counts (Ci_real / Co_real where the kernel sees padded ones)."""
    return 2 * d["N"] * d["Tq"] * d["Hq"] * d["Wq"] * d.get("Co_real", d["Co"]) * d.get("Ci_real", d["Ci"]) * d["ntap"][0] * d["ntap"][1] * d["ntap"][2]


def conv_work(d):
    """pc_conv_work of a (trimmed) conv descriptor dict -> dict(issued, executed, valid [MACs], blocks, bm, bn, glds)."""
    import ctypes as C
    out = (C.c_double * 7)()
    capi.check(capi.lib().pc_conv_work(_cdesc(d), int(d.get("Ci_real", 0)), int(d.get("Co_real", 0)), out))
    return dict(issued=out[0], executed=out[1], valid=out[2], blocks=int(out[3]), bm=int(out[4]), bn=int(out[5]), glds=bool(out[6]))


def _wdesc(d):
    import ctypes as C
    st = capi.WgradDesc()
    for name, ctype in st._fields_:
        v = d.get(name, 0)
        setattr(st, name, (C.c_int32 * 3)(*[int(x) for x in v]) if isinstance(v, (list, tuple)) else ctype(v))
    return C.byref(st)


def wgrad_work(d):
    """pc_wgrad_work of a (trimmed) wgrad descriptor dict -> dict(issued, executed, valid [MACs], route)."""
    import ctypes as C
    out = (C.c_double * 5)()
    capi.check(capi.lib().pc_wgrad_work(_wdesc(d), int(d.get("Cd_real", 0)), int(d.get("Cs_real", 0)), out))
    return dict(issued=out[0], executed=out[1], valid=out[2], route=int(out[3]), launches=int(out[4]))


def _wgrad_flops(d):
    return (2 * d["N"] * d["Tq"] * d["Hq"] * d["Wq"] * d.get("Cd_real", d["Cd"]) * d.get("Cs_real", d["Cs"]) * d["ntap"][0] * d["ntap"][1] * d["ntap"][2]
            * max(1, d.get("nbatch", 0)))


# --- workspace sizes come from the C side itself (host-only entry points of libpicons.so; tests/test_capi_cpu.py checks the symbol set)
def _bnpart_rows(d):
    return capi.lib().pc_conv_bnpart_rows(_cdesc(d))


def _cdesc(d):
    import ctypes as C
    st = capi.ConvDesc()
    for name, ctype in st._fields_:
        v = d[name]
        setattr(st, name, (C.c_int32 * 3)(*[int(x) for x in v]) if isinstance(v, (list, tuple)) else ctype(v))
    return C.byref(st)


def _bn_bwd_ws(rows, C_, groups):
    return capi.lib().pc_bn_bwd_ws_floats(int(rows), int(C_), int(groups))


def _act_bwd_ws(rows, C_):
    return capi.lib().pc_act_bwd_ws_floats(int(rows), int(C_))


def _em_ws(npos, B, C_):
    return capi.lib().pc_em_ws_floats(int(npos), int(B), int(C_))


def _loss_ws(B, hw):
    import ctypes as C
    d = capi.LossDesc()
    d.B, d.T, d.H, d.W = B, spec.FRAMES, hw, hw
    return capi.lib().pc_loss_ws_floats(C.byref(d))
