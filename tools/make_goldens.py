#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF on CPU (authoring container only).

    python tools/make_goldens.py [--only stages|masks|steps|steps_f64|traj] [--steps tag,tag]

Fixtures are data only (inputs/expected outputs); nothing of the reference's source travels.
Each file records torch version and thread count (results differ across thread counts,
SURVEY finding 4).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import picons_amd  # noqa: E402
from picons_amd import spec, synthetic  # noqa: E402
from tools import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
META = dict(torch_version=torch.__version__, threads=torch.get_num_threads())


def rng(seed):
    return np.random.default_rng(seed)


def t32(a):
    return torch.from_numpy(np.asarray(a, np.float32))


def save(name, d):
    d = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
    d["_torch_version"] = np.array(META["torch_version"])
    d["_threads"] = np.array(META["threads"])
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **d)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- stages
def gen_stages():
    from models.pytorch_i3d import Unit3D, MaxPool3dSamePadding, InceptionModule
    from models.capsules_ucf101 import PrimaryCaps, ConvCaps
    from utils.losses import SpreadLoss, DiceLoss, weighted_mse_loss
    from utils import ramp_ups
    d = {}
    g = rng(11)

    def run_unit(tag, cin, cout, k, s, shape):
        u = Unit3D(cin, cout, kernel_shape=list(k), stride=s)
        w = g.normal(0, np.sqrt(2.0 / (cin * np.prod(k))), (cout, cin) + tuple(k)).astype(np.float32)
        ga = g.uniform(0.5, 1.5, cout).astype(np.float32); be = g.normal(0, 0.1, cout).astype(np.float32)
        with torch.no_grad():
            u.conv3d.weight.copy_(t32(w)); u.bn.weight.copy_(t32(ga)); u.bn.bias.copy_(t32(be))
        x = t32(g.normal(0, 1, shape)).requires_grad_(True)
        u.train()
        y = u(x)
        dy = t32(g.normal(0, 1, tuple(y.shape)))
        y.backward(dy)
        d.update({tag + "_x": x, tag + "_w": w, tag + "_gamma": ga, tag + "_beta": be, tag + "_y": y,
                  tag + "_dy": dy, tag + "_dx": x.grad, tag + "_dw": u.conv3d.weight.grad,
                  tag + "_dgamma": u.bn.weight.grad, tag + "_dbeta": u.bn.bias.grad,
                  tag + "_rm": u.bn.running_mean, tag + "_rv": u.bn.running_var,
                  tag + "_k": np.array(k), tag + "_s": np.array(s)})
        u.eval()
        d[tag + "_y_eval"] = u(x.detach())

    run_unit("u333", 16, 24, (3, 3, 3), (2, 1, 1), (2, 16, 4, 12, 12))
    run_unit("u777", 3, 8, (7, 7, 7), (2, 2, 2), (1, 3, 8, 18, 18))
    run_unit("u111", 20, 12, (1, 1, 1), (1, 1, 1), (2, 20, 2, 7, 7))

    for tag, k, s, shape in [("p133", (1, 3, 3), (1, 2, 2), (2, 8, 3, 10, 10)),
                             ("p333s2", (3, 3, 3), (2, 1, 1), (2, 8, 2, 7, 7)),
                             ("p333s1", (3, 3, 3), (1, 1, 1), (1, 8, 2, 6, 6)),
                             ("p133odd", (1, 3, 3), (1, 2, 2), (1, 4, 1, 9, 9))]:
        mp = MaxPool3dSamePadding(kernel_size=list(k), stride=s, padding=0)
        x = torch.relu(t32(g.normal(0, 1, shape))).requires_grad_(True)
        y = mp(x)
        dy = t32(g.normal(0, 1, tuple(y.shape)))
        y.backward(dy)
        d.update({tag + "_x": x, tag + "_y": y, tag + "_dy": dy, tag + "_dx": x.grad,
                  tag + "_k": np.array(k), tag + "_s": np.array(s)})

    oc = [8, 8, 12, 4, 8, 8]
    inc = InceptionModule(16, oc, "t")
    inc.train()
    x = torch.relu(t32(g.normal(0, 1, (2, 16, 2, 6, 6))))
    names = []
    with torch.no_grad():
        for n, p in inc.named_parameters():
            p.copy_(t32(g.normal(0, 0.3, tuple(p.shape)))) if "conv3d" in n else \
                p.copy_(t32(g.uniform(0.5, 1.5, tuple(p.shape))))
            d["inc_p_" + n] = p.clone()
            names.append(n)
    d["inc_x"] = x; d["inc_y"] = inc(x); d["inc_oc"] = np.array(oc)

    # capsule head, small and full-size
    for tag, A, B, C, K, hw in [("capS", 12, 4, 5, 3, 6), ("capF", 8, 32, 24, 2, 4)]:
        pc = PrimaryCaps(A, B, K, 4, 1)
        cc = ConvCaps(B, C, (1, 1), 4, (1, 1), 3)
        with torch.no_grad():
            pc.pose.weight.copy_(t32(g.normal(0, 0.15, tuple(pc.pose.weight.shape))))
            pc.a.weight.copy_(t32(g.normal(0, 0.15, tuple(pc.a.weight.shape))))
            pc.pose.bias.copy_(t32(g.normal(0, 0.1, tuple(pc.pose.bias.shape))))
            pc.a.bias.copy_(t32(g.normal(0, 0.1, tuple(pc.a.bias.shape))))
            cc.weights.copy_(t32(g.normal(0, 0.5, tuple(cc.weights.shape))))
            cc.beta_u.copy_(t32(g.normal(0, 1, tuple(cc.beta_u.shape))))
            cc.beta_a.copy_(t32(g.normal(0, 1, tuple(cc.beta_a.shape))))
        x = torch.relu(t32(g.normal(0, 1, (2, A, hw, hw)))).requires_grad_(True)
        pcout = pc(x)
        pcout.retain_grad()
        out = cc(pcout)
        dout = t32(g.normal(0, 1, tuple(out.shape)))
        out.backward(dout)
        d.update({tag + "_x": x, tag + "_pose_w": pc.pose.weight, tag + "_pose_b": pc.pose.bias,
                  tag + "_a_w": pc.a.weight, tag + "_a_b": pc.a.bias, tag + "_W": cc.weights,
                  tag + "_beta_u": cc.beta_u, tag + "_beta_a": cc.beta_a, tag + "_pc": pcout,
                  tag + "_out": out, tag + "_dout": dout, tag + "_dpc": pcout.grad, tag + "_dx": x.grad,
                  tag + "_dW": cc.weights.grad, tag + "_dbeta_u": cc.beta_u.grad,
                  tag + "_dbeta_a": cc.beta_a.grad})

    # losses
    x = torch.sigmoid(t32(g.normal(0, 1, (5, 24)))).requires_grad_(True)
    tgt = t32(g.integers(0, 24, (5, 1)))
    l, al = SpreadLoss(num_class=24, m_min=0.2, m_max=0.9)(x, tgt)
    l.backward()
    d.update(dict(spread_x=x, spread_t=tgt, spread_loss=l, spread_abs=al, spread_dx=x.grad))
    lg = t32(g.normal(0, 2, (3, 1, 8, 10, 10))).requires_grad_(True)
    tg = t32(g.integers(0, 2, (3, 1, 8, 10, 10)))
    dl = DiceLoss()(lg, tg)
    bl = torch.nn.BCEWithLogitsLoss(size_average=True)(lg, tg)
    (dl + bl).backward()
    d.update(dict(seg_logits=lg, seg_t=tg, dice=dl, bce=bl, seg_dlogits=lg.grad))
    a = t32(g.normal(0, 1, (3, 1, 8, 6, 6))); b = t32(g.normal(0, 1, (3, 1, 8, 6, 6)))
    w5 = t32(g.random((3, 1, 8, 6, 6))); w4 = t32(g.random((3, 8, 6, 6)))
    d.update(dict(wm_a=a, wm_b=b, wm_w5=w5, wm_w4=w4, wm_l5=weighted_mse_loss(a, b, w5),
                  wm_l4=weighted_mse_loss(a, b, w4)))          # w4: the (B,B,...) gv broadcast
    d["ramp_100"] = np.array([ramp_ups.exp_rampup(100)(e) for e in (0, 1, 11, 50, 99, 100, 150)])
    save("stages.npz", d)


# ----------------------------------------------------------------------------- masks
def gen_masks():
    from utils.helpers import measure_pixelwise_var_v2, measure_pixelwise_gradient
    g = rng(23)
    d = {}
    pred = g.normal(0, 2, (2, 1, 8, 224, 224)).astype(np.float32)
    flip = (pred[:, :, ::-1] * 0.7 + g.normal(0, 1, pred.shape)).astype(np.float32)
    d["seed_note"] = np.array("pred=default_rng(23).normal(0,2,(2,1,8,224,224)).f32; "
                              "flip=(pred[:,:,::-1]*0.7+normal(0,1)).f32")

    def summar(tag, m):
        m = m.numpy()
        d[tag + "_sample"] = m[..., ::7, ::7].copy()
        d[tag + "_sum"] = m.sum(axis=(-1, -2))
        d[tag + "_sumsq"] = (m * m).sum(axis=(-1, -2))
        d[tag + "_shape"] = np.array(m.shape)
        d[tag + "_dtype"] = np.array(str(m.dtype))

    for nf in (3, 5):
        for sig in (False, True):
            summar("var%d%s" % (nf, "s" if sig else ""),
                   measure_pixelwise_var_v2(t32(pred), t32(flip), frames_cnt=nf, use_sig_output=sig))
    summar("grad", measure_pixelwise_gradient(t32(pred)))
    summar("grad_thr", measure_pixelwise_gradient(t32(pred), 0.2, 0.85))
    save("masks.npz", d)


# ----------------------------------------------------------------------------- full steps
def gen_steps(only_tags=None, add_f64=False):
    import importlib
    from models.capsules_ucf101 import CapsNet, ConvCaps
    import torch.nn as nn
    from utils.losses import SpreadLoss, DiceLoss
    from oracle.step import default_args
    from oracle.losses import exp_rampup

    # (tag, main module, classes, flags, epoch, synthetic step id, bs, conditioned init, also run the reference in fp64)
    cases = [
        ("step_bv5", "main_ucf101", 24, dict(bv=True, n_frames=5, wt_cons=0.1), 1, 0, 2, True, False),
        ("step_gv_pseudo", "main_ucf101", 24, dict(gv=True, lower_thresh=0.2, upper_thresh=0.9, wt_cons=0.1), 12, 1, 2, True, False),
        ("step_bvgv3", "main_ucf101", 24, dict(bv=True, gv=True, n_frames=3, predict_maps=True), 3, 2, 2, True, False),
        ("step_jhmdb_bv", "main_jhmdb", 21, dict(bv=True, n_frames=5, wt_cons=0.1, dataset="jhmdb"), 1, 3, 2, True, False),
        # BASELINE configs[1] / configs[2] at the batch size the metric is quoted on (4 labeled + 4 unlabeled)
        ("step_bv5_bs8", "main_ucf101", 24, dict(bv=True, n_frames=5, wt_cons=0.1), 1, 4, 8, True, False),
        ("step_gv_bs8", "main_ucf101", 24, dict(gv=True, wt_cons=0.1), 1, 5, 8, True, False),
        # the two other per-rank workloads bench.py times, at that batch size too: BASELINE configs[4]'s 21-class JHMDB step
        # (main_jhmdb.py:50-140: synthesised labels :68-70, gv overrides bv :121,132) and the epoch >= thresh_epoch branch
        # (argmax pseudo-labels for the unlabeled rows, capsules_ucf101.py:463)
        ("step_jhmdb_bv_bs8", "main_jhmdb", 21, dict(bv=True, n_frames=5, wt_cons=0.1, dataset="jhmdb"), 1, 7, 8, True, False),
        ("step_gv_pseudo_bs8", "main_ucf101", 24, dict(gv=True, lower_thresh=0.2, upper_thresh=0.9, wt_cons=0.1), 12, 8, 8, True, False),
        # the reference's own initialisation (PrimaryCaps std 0.1, ConvCaps.weights randn: capsules_ucf101.py:36,39,103),
        # fp32 and fp64 runs of the reference (SURVEY 8c "reference-init case")
        ("step_refinit_bv5", "main_ucf101", 24, dict(bv=True, n_frames=5, wt_cons=0.1), 1, 6, 2, False, True),
    ]
    if only_tags:
        cases = [c for c in cases if c[0] in only_tags]
    BIG = ["conv1.Conv3d_1a_7x7.conv3d.weight", "conv112.weight"]        # full gradients of two 224-only layers (bs = 8 cases)

    def run_reference(case, double):
        """One train step of the reference itself (main_*.train_model_interface + backward) -> dict of outputs."""
        tag, mainmod, ncls, akw, epoch, stepid, bs, conditioned, _with64 = case
        ref_import.install_shims(double=double)
        state = synthetic.init_state(seed=47, num_classes=ncls, conditioned=conditioned)
        pt = ref_import.synthetic_charades(state)
        model = CapsNet(pt_path=pt)
        if ncls != 24:       # the 21-class file is absent; assemble from the same classes
            model.conv_caps = ConvCaps(32, ncls, (1, 1), 4, stride=(1, 1), iters=3)
            model.upsample1 = nn.ConvTranspose2d(ncls * 16, 64, kernel_size=9, stride=1, padding=0)
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        if double:
            model.double()
            model.conv_caps.ln_2pi = model.conv_caps.ln_2pi.double()
        lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=stepid, num_classes=ncls)
        model.dropout3d = ref_import.ScriptedDropout(drops)
        model.train(True); model.training = True
        main = importlib.import_module(mainmod)
        main.model = model
        main.criterion_cls = SpreadLoss(num_class=ncls, m_min=0.2, m_max=0.9)
        main.criterion_seg_1 = nn.BCEWithLogitsLoss(size_average=True)
        main.criterion_seg_2 = DiceLoss()
        args = default_args(**akw)
        if mainmod == "main_jhmdb":
            args.wt_seg = args.wt_loc
        ramp = exp_rampup(100)(epoch)
        tomb = lambda mb: {k: torch.from_numpy(v) for k, v in mb.items()}
        orig_randperm = torch.randperm
        torch.randperm = lambda n, *a, **k: torch.from_numpy(perm.copy())
        try:
            out = main.train_model_interface(args, tomb(lab), tomb(unl), epoch, ramp)
        finally:
            torch.randperm = orig_randperm
        output, pred_action, _seg, _act, total, loc, cls, cons = out
        total.backward()
        d = dict(num_classes=np.array(ncls), epoch=np.array(epoch), stepid=np.array(stepid), ramp=np.array(ramp),
                 bs=np.array(bs), conditioned=np.array(int(conditioned)),
                 args=np.array(repr(sorted(vars(args).items()))),
                 predicted_action=pred_action, output_sample=output[:, :, :, ::8, ::8],
                 output_frame_sum=output.sum(dim=(-1, -2)), output_min=output.min(), output_max=output.max(),
                 total=total, loc=loc, cls=cls, cons=cons)
        gn = {}
        params = dict(model.named_parameters())
        for n, p in params.items():
            gn[n] = float(p.grad.norm()) if p.grad is not None else -1.0
        d["grad_names"] = np.array(list(gn.keys()))
        d["grad_norms"] = np.array(list(gn.values()))
        for n in ["conv_caps.weights", "conv_caps.beta_u", "conv_caps.beta_a", "smooth.weight", "smooth.bias",
                  "upsample4.bias", "conv1.Conv3d_1a_7x7.bn.weight", "conv1.Conv3d_1a_7x7.bn.bias",
                  "conv1.Mixed_4f.b3b.bn.weight", "conv1.Mixed_3b.b1b.bn.bias", "primary_caps.a.bias",
                  "conv1.Mixed_4f.b0.conv3d.weight", "conv28.bias"] + (BIG if bs == 8 else []):
            d["grad::" + n] = params[n].grad
        if bs == 8 or not conditioned:
            # strided samples of the big 224-only / PrimaryCaps gradients (full tensors are MBs)
            d["gsample::upsample3.weight"] = params["upsample3.weight"].grad.reshape(-1)[::7]
            d["gsample::upsample4.weight"] = params["upsample4.weight"].grad.reshape(-1)[::13]
            d["gsample::primary_caps.pose.weight"] = params["primary_caps.pose.weight"].grad.reshape(-1)[::997]
            d["gsample::primary_caps.a.weight"] = params["primary_caps.a.weight"].grad.reshape(-1)[::97]
        sd = model.state_dict()
        for n in ["conv1.Conv3d_1a_7x7.bn.running_mean", "conv1.Conv3d_1a_7x7.bn.running_var",
                  "conv1.Mixed_4f.b3b.bn.running_mean", "conv1.Mixed_4f.b3b.bn.running_var",
                  "conv1.Mixed_4f.b3b.bn.num_batches_tracked"]:
            d["buf::" + n] = sd[n]
        return d

    for case in cases:
        t0 = time.time()
        if add_f64:
            # `--only steps_f64 --steps tags` (round 5): an fp64 run of the reference added to an EXISTING fixture, whose fp32 contents stay as
            # they are -- the anchor the element-wise gradient bars of the bs = 2 fixtures are judged against
            old = np.load(os.path.join(OUT, case[0] + ".npz"))
            d = {k: old[k] for k in old.files if not k.startswith("_")}
            d64 = run_reference(case, True)
            ref_import.install_shims(double=False)
            for k, v in d64.items():
                if k.startswith(("grad::", "gsample::")) or k in ("predicted_action", "total", "loc", "cls", "cons", "grad_norms"):
                    d["f64::" + k] = v
            d["seconds_f64"] = np.array(time.time() - t0)
            save(case[0] + ".npz", d)
            continue
        d = run_reference(case, False)
        if case[8]:
            d64 = run_reference(case, True)
            ref_import.install_shims(double=False)
            for k, v in d64.items():        # the fp64 run of the reference: the anchor the fp32 runs are judged against
                if k.startswith(("grad::", "gsample::")) or k in ("predicted_action", "output_sample", "output_frame_sum",
                                                                  "total", "loc", "cls", "cons", "grad_norms"):
                    d["f64::" + k] = v
        d["seconds"] = np.array(time.time() - t0)
        save(case[0] + ".npz", d)


# ----------------------------------------------------------------------------- a training trajectory
TRAJ_STEPIDS = (20, 21, 22)
TRAJ_PARAMS = ["conv_caps.beta_u", "conv_caps.beta_a", "smooth.weight", "smooth.bias", "upsample4.bias", "conv28.bias", "primary_caps.a.bias",
               "conv1.Conv3d_1a_7x7.bn.weight", "conv1.Conv3d_1a_7x7.bn.bias", "conv1.Mixed_4f.b3b.bn.weight", "conv1.Mixed_3b.b1b.bn.bias",
               "conv1.Mixed_4f.b0.conv3d.weight", "conv_caps.weights"]
TRAJ_BUFS = ["conv1.Conv3d_1a_7x7", "conv1.Conv3d_2c_3x3", "conv1.Mixed_3b.b1b", "conv1.Mixed_4c.b0", "conv1.Mixed_4f.b3b"]


def gen_trajectory(bs=2, nsteps=3, spread_only=False, spread_threads=(1, 3, 5), adam_only=False):
    """The reference's own training loop for `nsteps` steps (main_ucf101.py:171-184: zero_grad -> train_model_interface -> backward ->
    optimizer.step(), optimizer = Adam(lr 1e-4, weight_decay 0, eps 1e-6) of main_ucf101.py:416), a FRESH minibatch per step, scripted
    permutation / dropout draws -- once in fp32 and once in fp64 (the anchor: Adam's first updates are +-lr per element whatever the
    gradient's size, so two fp32 implementations whose gradient NOISE differs drift apart from step 2 on; the fp64 run says how far the
    reference's own fp32 arithmetic is from the exact trajectory).  Stored per run: the loss scalars and class predictions of every step,
    the BatchNorm running statistics and num_batches_tracked after 2 * nsteps forward passes, a few small parameters and the norm of every
    parameter after the last step.
    Plus (`--only traj_spread` adds them to an existing fixture): the fp32 run again under other intra-op thread counts -- other reduction orders, the
    only knob the reference's arithmetic has -- loss scalars and class predictions only, prefix `t<k>::`: how far the reference's fp32 trajectory is
    from ITSELF, i.e. the spread the single fp32 run samples once."""
    import importlib
    import torch.nn as nn
    from oracle.step import default_args
    from oracle.losses import exp_rampup
    t0 = time.time()
    lr = 1e-4
    epoch, ramp = 1, exp_rampup(100)(1)
    args = default_args(bv=True, n_frames=5, wt_cons=0.1)
    d = dict(bs=np.array(bs), nsteps=np.array(nsteps), stepids=np.array(TRAJ_STEPIDS[:nsteps]), lr=np.array(lr), epoch=np.array(epoch), ramp=np.array(ramp),
             args=np.array(repr(sorted(vars(args).items()))))

    if spread_only:
        old = np.load(os.path.join(OUT, "traj_bv5.npz"))
        d = {k: old[k] for k in old.files if not (k[0] == "t" and k[1].isdigit())}
    if adam_only:
        # `--only traj_adam` (round 5): the fp32 and fp64 runs again, adding the optimiser's state after the last step -- Adam's exp_avg /
        # exp_avg_sq of the TRAJ_PARAMS tensors, the norms of both for every parameter and the step count -- to the existing fixture.  The
        # runs are bit-reproducible at a fixed thread count: every key the fixture already holds must come out identical (checked below).
        old = np.load(os.path.join(OUT, "traj_bv5.npz"))
        keep = {k: old[k] for k in old.files}
        assert int(old["default_threads"]) == torch.get_num_threads(), "run with the thread count the fixture was made with (%d)" % int(old["default_threads"])

    def run(double, pre, light=False):
        ref_import.install_shims(double=double)
        from models.capsules_ucf101 import CapsNet
        from utils.losses import SpreadLoss, DiceLoss
        state = synthetic.init_state(seed=47, num_classes=24, conditioned=True)
        model = CapsNet(pt_path=ref_import.synthetic_charades(state))
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        if double:
            model.double()
            model.conv_caps.ln_2pi = model.conv_caps.ln_2pi.double()
        model.train(True); model.training = True
        main = importlib.import_module("main_ucf101")
        main.model = model
        main.criterion_cls = SpreadLoss(num_class=24, m_min=0.2, m_max=0.9)
        main.criterion_seg_1 = nn.BCEWithLogitsLoss(size_average=True)
        main.criterion_seg_2 = DiceLoss()
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0, eps=1e-6)          # main_ucf101.py:416
        orig_randperm = torch.randperm
        for s in range(nsteps):
            lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=TRAJ_STEPIDS[s], num_classes=24)
            model.dropout3d = ref_import.ScriptedDropout(drops)
            tomb = lambda mb: {k: torch.from_numpy(v) for k, v in mb.items()}
            opt.zero_grad()
            torch.randperm = lambda n, *a, **k: torch.from_numpy(perm.copy())
            try:
                out = main.train_model_interface(args, tomb(lab), tomb(unl), epoch, ramp)
            finally:
                torch.randperm = orig_randperm
            output, pred_action, _seg, _act, total, loc, cls, cons = out
            total.backward()
            opt.step()
            for k, v in (("total", total), ("loc", loc), ("cls", cls), ("cons", cons), ("predicted_action", pred_action),
                         ("output_frame_sum", output.sum(dim=(-1, -2)))):
                if not (light and k == "output_frame_sum"):
                    d["%ss%d::%s" % (pre, s, k)] = v
            print("%strajectory step %d: total %.6f loc %.6f cls %.6f cons %.6f  (%.0f s)" % (pre, s, float(total), float(loc), float(cls), float(cons), time.time() - t0))
        if light:
            return
        sd = model.state_dict()
        for p_ in TRAJ_BUFS:
            for nm in ("running_mean", "running_var", "num_batches_tracked"):
                d["%sbuf::%s.bn.%s" % (pre, p_, nm)] = sd["%s.bn.%s" % (p_, nm)]
        params = dict(model.named_parameters())
        for n in TRAJ_PARAMS:
            d["%sparam::%s" % (pre, n)] = params[n].detach()
        d[pre + "param_names"] = np.array(list(params.keys()))
        d[pre + "param_norms"] = np.array([float(p.detach().double().norm()) for p in params.values()])
        d[pre + "param_delta_norms"] = np.array([float((p.detach().double() - torch.from_numpy(np.array(state[n])).double()).norm()) for n, p in params.items()])
        # the optimiser's state after the last step: a wrong step count, a doubled or a skipped update shows here directly
        for n in TRAJ_PARAMS:
            if params[n].numel() > 20000:                       # full moments of the small tensors only (norms of every tensor below)
                continue
            st = opt.state[params[n]]
            d["%sadam_m::%s" % (pre, n)] = st["exp_avg"].detach()
            d["%sadam_v::%s" % (pre, n)] = st["exp_avg_sq"].detach()
        d[pre + "adam_m_norms"] = np.array([float(opt.state[p]["exp_avg"].double().norm()) for p in params.values()])
        d[pre + "adam_v_norms"] = np.array([float(opt.state[p]["exp_avg_sq"].double().norm()) for p in params.values()])
        d[pre + "adam_step"] = np.array([int(opt.state[p]["step"]) for p in params.values()])
    if adam_only:
        run(False, "")
        run(True, "f64::")
        for k, v in keep.items():
            if k in d and k not in ("seconds",):
                a_, b_ = np.asarray(v), np.asarray(d[k].detach().numpy() if torch.is_tensor(d[k]) else d[k])
                if a_.dtype.kind in "fc":
                    assert np.array_equal(a_, b_.astype(a_.dtype).reshape(a_.shape)), "the re-run does not reproduce the fixture's %s" % k
        new = {k: v for k, v in d.items() if k not in keep}
        keep.update(new)
        ref_import.install_shims(double=False)
        print("added:", sorted(new))
        save("traj_bv5.npz", keep)
        return
    if not spread_only:
        run(False, "")
        run(True, "f64::")
    nt = torch.get_num_threads()
    for k in spread_threads:
        torch.set_num_threads(k)
        try:
            run(False, "t%d::" % k, light=True)
        finally:
            torch.set_num_threads(nt)
    d["spread_threads"] = np.array(list(spread_threads))
    d["default_threads"] = np.array(nt)
    ref_import.install_shims(double=False)
    d["seconds"] = np.array(time.time() - t0)
    save("traj_bv5.npz", d)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--steps", default=None, help="comma-separated step fixture tags (default: all)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    ref_import.install_shims()
    torch.manual_seed(0)
    if a.only in (None, "stages"):
        gen_stages()
    if a.only in (None, "masks"):
        gen_masks()
    if a.only in (None, "steps"):
        gen_steps(a.steps.split(",") if a.steps else None)
    if a.only == "steps_f64":
        gen_steps(a.steps.split(","), add_f64=True)
    if a.only == "traj_spread":
        gen_trajectory(spread_only=True)
    if a.only == "traj_adam":
        gen_trajectory(adam_only=True)
    if a.only in (None, "traj"):
        gen_trajectory()
