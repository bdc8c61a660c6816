"""nn.Module mirror of the reference's CapsNet / InceptionI3d whose forward and backward run the
HIP op plans (drop-in for /root/reference/models/capsules_ucf101.py:334-512 and
models/pytorch_i3d.py:152-346).

Parameters are reference-shaped nn.Parameters that alias one flat device buffer (so state_dict keys,
shapes and layouts are the reference's: SURVEY §5 "293 entries"); the kernels see kernel-layout
copies made per forward.  A forward call returns tensors attached to a single autograd node whose
backward replays the plan's reverse list and accumulates into the parameters' .grad (views of one
flat gradient buffer) - so `loss.backward(); optimizer.step()` of main_ucf101.py:183-184 works
unchanged, including the reference's two forward passes per step (each call takes its own
activation arena from a small pool).  GPU only: there is no CPU path.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import capi, ops, spec, synthetic
from .plan import Plan


class _Slot:
    """One activation arena + resolved op lists for a given (batch, mode)."""

    def __init__(self, owner, n, training):
        self.plan = p = Plan(owner.num_classes, owner.hw, n=n, groups=1, training=training, accum_grads=True)
        p.build_forward()
        if training:
            p.build_seeds()
            p.build_backward()
        self.arena = torch.empty(p.arena_bytes + 256, device=owner.dev, dtype=torch.uint8)
        base = (self.arena.data_ptr() + 255) // 256 * 256
        self.a0 = base - self.arena.data_ptr()
        self.ops = p.resolve(dict(A=base, P=owner._P.data_ptr(), G=owner._G.data_ptr(), M=0, V=0, R=owner._R.data_ptr()))
        self.busy = False
        self.gen = 0             # bumped each time the slot is taken by a training forward (see _SlotGuard)
        p.upload_consts(self.view)

    def view(self, ref, nfloats, dtype=torch.float32):
        o = self.a0 + ref[1]
        return self.arena[o:o + 4 * nfloats].view(dtype)


class _SlotGuard:
    """Lives on the autograd ctx of one training forward.  If that graph is dropped without a backward (loss skipped, an
    exception between forward and backward), the ctx dies and the arena slot is released instead of staying busy forever;
    the generation check keeps a late finaliser from freeing a slot that a newer forward has taken since."""

    def __init__(self, slot):
        self.slot, self.gen = slot, slot.gen

    def __del__(self):
        if self.slot.gen == self.gen:
            self.slot.busy = False


class _CapsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, slot, n, *params):
        ctx.owner, ctx.slot, ctx.n = owner, slot, n
        ctx.guard = _SlotGuard(slot)
        ops.run_ops(slot.ops["prep"]); ops.run_ops(slot.ops["prep_late"])
        ops.run_ops(slot.ops["fwd"])
        p = slot.plan
        per = spec.FRAMES * owner.hw * owner.hw
        out = slot.view(p.out.ref, n * per).view(n, 1, spec.FRAMES, owner.hw, owner.hw)
        pred = slot.view(p.pred, n * owner.num_classes).view(n, owner.num_classes)
        s20 = p.named["comb"].thw[1]
        comb = slot.view(p.named["comb"].ref, n * s20 * s20 * owner.num_classes * 17).view(n, s20 * s20, owner.num_classes * 17)
        feat = comb[:, :, owner.num_classes * 16:]
        # outputs are copies: the arena slot is recycled after backward
        return out.clone(), pred.clone(), feat.clone()

    @staticmethod
    def backward(ctx, d_out, d_pred, d_feat):
        owner, slot, n = ctx.owner, ctx.slot, ctx.n
        p = slot.plan
        per = spec.FRAMES * owner.hw * owner.hw
        owner._attach_grads()
        dst = slot.view(p.dout, n * per)
        dst.zero_() if d_out is None else dst.copy_(d_out.reshape(-1))
        dp = slot.view(p.dpred, n * owner.num_classes)
        dp.zero_() if d_pred is None else dp.copy_(d_pred.reshape(-1))
        if d_feat is not None and bool((d_feat != 0).any()):
            raise RuntimeError("gradient through feat_shape is not supported (unused by the reference's callers)")
        ops.run_ops(slot.ops["bwd"])
        slot.busy = False
        return (None, None, None) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))


class CapsNet(nn.Module):
    """capsules_ucf101.py:334-512.  Initial state: init="reference" (default) draws what constructing the reference's
    CapsNet leaves in the parameters (synthetic.init_state_module: PrimaryCaps N(0, 0.1), ConvCaps randn, BN 1/0, PyTorch's
    default conv init); init="conditioned" is the parity-test initialiser (SURVEY finding 4).  `pt_path` supplies the trunk
    (keys copied like :343-352); a missing file raises like the reference's torch.load does, unless PICONS_SYNTHETIC is set
    (no network for the real rgb_charades.pt) or pt_path is None.  hw / num_classes generalise the fixed 224 / 24."""

    def __init__(self, pt_path='../weights/rgb_charades.pt', P=4, pretrained_load='i3d', num_classes=24, hw=224,
                 device=None, seed=47, init="reference"):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("CapsNet (HIP) needs a GPU: there is no CPU fallback")
        capi.lib()
        assert P == 4
        self.P = P
        self.num_classes = num_classes
        self.hw = hw
        self.dev = torch.device(device or "cuda:%d" % torch.cuda.current_device())
        lay = Plan(num_classes, hw, n=1, groups=1)
        self._pshape, self._poff, self._roff = lay.pshape, lay.poff, lay.roff
        self._P = torch.zeros(lay.nparams, device=self.dev)
        self._G = torch.zeros(lay.nparams, device=self.dev)
        self._R = torch.zeros(lay.nrunning, device=self.dev)
        self._pnames = list(self._pshape)
        self._params = nn.ParameterList()
        for k, shp in self._pshape.items():
            o = self._poff[k]
            self._params.append(nn.Parameter(self._P[o:o + int(np.prod(shp))].view(shp)))
        self._nbt = {k: 0 for k in spec.buffer_shapes() if k.endswith("num_batches_tracked")}
        self._slots = {}
        if init not in ("reference", "conditioned"):
            raise ValueError("init must be 'reference' or 'conditioned'")
        self.load_state_dict(synthetic.init_state_module(seed, num_classes) if init == "reference" else synthetic.init_state(seed, num_classes))
        import os
        import sys
        if pt_path and not os.path.exists(pt_path):
            if os.environ.get("PICONS_SYNTHETIC", "") in ("", "0"):
                raise FileNotFoundError("pretrained trunk %r not found (the reference's torch.load raises here too: "
                                        "capsules_ucf101.py:344); pass pt_path=None or set PICONS_SYNTHETIC=1 to train the trunk from its "
                                        "random initialisation" % pt_path)
            print("WARNING: pretrained trunk %r not found; PICONS_SYNTHETIC is set, the I3D trunk keeps its random initialisation" % pt_path,
                  file=sys.stderr)
        if pt_path and os.path.exists(pt_path):
            pre = torch.load(pt_path, map_location="cpu")
            sd = {"conv1." + k: v for k, v in pre.items() if "conv1." + k in self._keyset()}
            self.load_state_dict(sd, strict=False)
            print("Loaded I3D pretrained weights from ", pt_path, " for layers: ", len(sd))

    # ---- nn.Module protocol used by the reference's callers
    def _keyset(self):
        return set(spec.state_dict_keys(self.num_classes))

    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        for k, p in zip(self._pnames, self._params):
            yield (prefix + ("." if prefix else "") + k, p)

    def parameters(self, recurse=True):
        for _k, p in self.named_parameters():
            yield p

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def state_dict(self, *a, **k):
        sd = OrderedDict()
        pm = dict(zip(self._pnames, self._params))
        for key in spec.state_dict_keys(self.num_classes):
            if key in pm:
                sd[key] = pm[key].detach().clone()
            elif key in self._roff:
                co = self._pshape[key.rsplit(".bn.", 1)[0] + ".bn.weight"][0]
                sd[key] = self._R[self._roff[key]:self._roff[key] + co].clone()
            else:
                sd[key] = torch.tensor(self._nbt[key], dtype=torch.long)
        return sd

    def load_state_dict(self, state, strict=True):
        pm = dict(zip(self._pnames, self._params))
        missing = [k for k in self._keyset() if k not in state]
        if strict and missing:
            raise KeyError("missing keys: %s" % missing[:5])
        with torch.no_grad():
            for k, v in state.items():
                v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
                if k in pm:
                    pm[k].copy_(v.to(self.dev, torch.float32).reshape(pm[k].shape))
                elif k in self._roff:
                    self._R[self._roff[k]:self._roff[k] + v.numel()].copy_(v.to(self.dev, torch.float32))
                elif k in self._nbt:
                    self._nbt[k] = int(v)
                elif strict:
                    raise KeyError("unexpected key " + k)

    def load_previous_weights(self, weightfile):
        self.load_state_dict(torch.load(weightfile, map_location="cpu"), strict=False)
        print('loaded weights from previous run: ', weightfile)

    def zero_grad(self, set_to_none=True):
        self._G.zero_()
        for p in self._params:
            p.grad = None

    def _attach_grads(self):
        """(Re)attach .grad views of the flat gradient buffer; a None grad means the optimiser zeroed it."""
        if any(p.grad is None for p in self._params):
            self._G.zero_()
            for k, p in zip(self._pnames, self._params):
                o = self._poff[k]
                p.grad = self._G[o:o + p.numel()].view(p.shape)

    # ---- forward
    def _slot(self, n, training):
        pool = self._slots.setdefault((n, training), [])
        for s in pool:
            if not s.busy:
                return s
        s = _Slot(self, n, training)
        pool.append(s)
        return s

    def forward(self, img, classification, concat_labels, epoch, thresh_ep):
        """(B,3,8,H,W) clips -> (out_1 (B,1,8,H,W) logits, actor_prediction (B,C), feat_shape (B,h*w,C))."""
        n = img.shape[0]
        training = bool(self.training)
        s = self._slot(n, training)
        p = s.plan
        per = 3 * spec.FRAMES * self.hw * self.hw
        s.view(p.in_data, n * per).copy_(img.to(self.dev, torch.float32).reshape(-1))
        cls = torch.as_tensor(classification).to(self.dev, torch.float32).reshape(-1)
        lab = torch.as_tensor(concat_labels).to(self.dev).reshape(-1).to(torch.int32)
        s.view(p.in_cls, n).copy_(cls[:n])
        s.view(p.in_labeled, n, torch.int32).copy_(lab[:n])
        if training:
            # nn.Dropout3d(0.5) draws, capsules_ucf101.py:428,507 (per sample, per channel; scale 2)
            s.view(p.in_drop832, n * spec.TRUNK_OUT_CH).copy_((torch.rand(n * spec.TRUNK_OUT_CH, device=self.dev) < 0.5).float() * 2)
            s.view(p.in_drop128, n * 128).copy_((torch.rand(n * 128, device=self.dev) < 0.5).float() * 2)
            s.ops["fwd"][p.op_cmask]["i"][3] = 0 if epoch < thresh_ep else 1
            for k in self._nbt:
                self._nbt[k] += 1
            if torch.is_grad_enabled():
                s.busy = True            # released by _CapsFn.backward, or by _SlotGuard if the graph is dropped
                s.gen += 1
                return _CapsFn.apply(self, s, n, *self._params)
            # train-mode forward under no_grad (batch statistics, dropout): no backward will come, the slot stays free
        with torch.no_grad():
            ops.run_ops(s.ops["prep"]); ops.run_ops(s.ops["prep_late"])
            ops.run_ops(s.ops["fwd"])
            perm = spec.FRAMES * self.hw * self.hw
            out = s.view(p.out.ref, n * perm).view(n, 1, spec.FRAMES, self.hw, self.hw).clone()
            pred = s.view(p.pred, n * self.num_classes).view(n, self.num_classes).clone()
            s20 = p.named["comb"].thw[1]
            comb = s.view(p.named["comb"].ref, n * s20 * s20 * self.num_classes * 17).view(n, s20 * s20, -1)
            return out, pred, comb[:, :, self.num_classes * 16:].clone()


class InceptionI3d(nn.Module):
    """pytorch_i3d.py:152-346 (trunk to Mixed_4f), forward only: returns (Mixed_4f, out56, out112) as
    NCDHW-shaped tensors.  Training of the trunk goes through CapsNet, which owns its gradients."""

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint='Mixed_4f', name='inception_i3d',
                 in_channels=3, dropout_keep_prob=0.5, hw=224):
        super().__init__()
        if final_endpoint != 'Mixed_4f' or in_channels != 3:
            raise ValueError("the HIP trunk is built to final_endpoint='Mixed_4f', in_channels=3 (what CapsNet uses)")
        self._caps = CapsNet(pt_path=None, hw=hw)

    def state_dict(self, *a, **k):
        return OrderedDict((key[len("conv1."):], v) for key, v in self._caps.state_dict().items() if key.startswith("conv1."))

    def load_state_dict(self, state, strict=True):
        self._caps.load_state_dict({"conv1." + k: v for k, v in state.items()}, strict=False)

    def forward(self, x):
        c = self._caps
        n = x.shape[0]
        s = c._slot(n, bool(self.training))
        p = s.plan
        s.view(p.in_data, n * 3 * spec.FRAMES * c.hw * c.hw).copy_(x.to(c.dev, torch.float32).reshape(-1))
        if self.training:
            s.view(p.in_drop832, n * spec.TRUNK_OUT_CH).fill_(1.0)
            s.view(p.in_drop128, n * 128).fill_(1.0)
            s.view(p.in_cls, n).zero_()
            s.view(p.in_labeled, n, torch.int32).zero_()
        with torch.no_grad():
            ops.run_ops(s.ops["prep"]); ops.run_ops(s.ops["prep_late"])
            ops.run_ops(s.ops["fwd"])
            outs = []
            for nm in ("trunk_out", "conv1.Conv3d_2c_3x3.y", "conv1.Conv3d_1a_7x7.y"):
                t = p.named[nm]
                v = s.view(t.ref, (t.rows - 1) * t.ld + t.C).as_strided((t.rows, t.C), (t.ld, 1)).view(t.N, *t.thw, t.C)   # a channel slice
                outs.append(v.permute(0, 4, 1, 2, 3).clone())
        return tuple(outs)
