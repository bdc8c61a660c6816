"""Oracle: the semi-supervised train step (test infrastructure).

Follows /root/reference/main_ucf101.py train_model_interface :50-150 and the jhmdb variant
/root/reference/main_jhmdb.py :50-140 (label synthesis :68-70, gv overrides bv :121,132,
args.wt_seg :138), plus the zero_grad/backward/Adam loop main_ucf101.py:171-184,416.
The shuffle permutation (:73) and the Dropout3d draws are explicit inputs.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import caps, losses


def default_args(**kw):
    """CLI defaults of main_ucf101.py:285-315 relevant to the step."""
    a = dict(bv=False, gv=False, n_frames=3, predict_maps=False, lower_thresh=None,
             upper_thresh=None, bv_wt=0.5, gv_wt=0.5, wt_loc=1.0, wt_cls=1.0, wt_cons=1.0,
             thresh_epoch=11, dataset="ucf101")
    a.update(kw)
    return SimpleNamespace(**a)


def as_torch_params(state, dtype=torch.float32, requires_grad=True):
    """numpy state (reference layout) -> dict of torch tensors; trainable ones get grads."""
    P = {}
    for k, v in state.items():
        t = torch.from_numpy(np.array(v))
        if t.is_floating_point():
            t = t.to(dtype)
        if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked")):
            t.requires_grad_(True)
        P[k] = t
    return P


def train_step(P, args, label_mb, unlabel_mb, epoch, wt_ramp, perm, drops, dtype=torch.float32):
    """main_ucf101.py:50-150.  Minibatch dicts hold numpy/torch arrays per the dataloader
    contract; perm = the randperm of :73; drops = [d832_pass0, d128_pass0, d832_pass1, d128_pass1].
    Returns dict with outputs (shuffled order, like the reference) and the four loss scalars."""
    T = lambda a: torch.as_tensor(np.asarray(a))
    cat = lambda k: torch.cat([T(label_mb[k]), T(unlabel_mb[k])], dim=0)
    data = cat("data").to(dtype)
    fl_data = cat("aug_data").to(dtype)
    action = cat("action")
    seg = cat("loc_msk")
    if args.dataset == "jhmdb":                       # main_jhmdb.py:68-70
        n_l, n_u = len(label_mb["action"]), len(unlabel_mb["action"])
        labels = torch.cat([torch.ones(n_l), torch.zeros(n_u)]).long()
    else:
        labels = cat("label_vid")
    perm = torch.as_tensor(np.asarray(perm)).long()
    data, fl_data, action, labels, seg = data[perm], fl_data[perm], action[perm], labels[perm], seg[perm]
    lab_idx = torch.where(labels == 1)[0]
    d = [None if x is None else torch.as_tensor(np.asarray(x)).to(dtype) for x in drops]

    output, pred_action, _ = caps.capsnet_forward(P, data, action, labels, epoch, args.thresh_epoch,
                                                  True, d[0], d[1])
    flip_op, _, _ = caps.capsnet_forward(P, fl_data, action, labels, epoch, args.thresh_epoch,
                                         True, d[2], d[3])
    lab_op = output[lab_idx]
    lab_seg = seg[lab_idx].to(dtype)
    loc_loss = losses.bce_logits(lab_op, lab_seg) + losses.dice_loss(lab_op, lab_seg)
    class_loss, _abs = losses.spread_loss(pred_action[lab_idx], action[lab_idx])

    fpsm = torch.flip(flip_op, [4])                                      # :100
    l2 = losses.weighted_mse(fpsm, output, torch.ones_like(output))     # :105-107
    cons1 = cons2 = None
    if args.bv:                                                          # :112-124
        v_c = losses.var_mask(output, torch.flip(fpsm, [2]), args.n_frames, args.predict_maps).to(dtype)
        v_a = losses.var_mask(torch.flip(output, [2]), fpsm, args.n_frames, args.predict_maps).to(dtype)
        lv1 = losses.weighted_mse(fpsm, output, v_c)
        lv2 = losses.weighted_mse(fpsm, output, torch.flip(v_a, [2]))
        cons1 = wt_ramp * (lv1 + lv2) + (1 - wt_ramp) * l2
    if args.gv:                                                          # :129-133
        g = losses.grad_mask(output, args.lower_thresh, args.upper_thresh).to(dtype)
        cons2 = losses.weighted_mse(fpsm, output, g)
    if args.dataset == "jhmdb":                                          # main_jhmdb.py:121,132
        cons = cons2 if args.gv else (cons1 if args.bv else l2)
    elif args.bv and args.gv:                                            # :136-137
        cons = args.bv_wt * cons1 + args.gv_wt * cons2
    elif args.gv:
        cons = cons2
    elif args.bv:
        cons = cons1
    else:
        cons = l2
    total = args.wt_loc * loc_loss + args.wt_cls * class_loss + args.wt_cons * cons   # :147-148
    return dict(output=output, predicted_action=pred_action, flip_op=flip_op, seg=seg, action=action,
                labels=labels, total=total, loc=loc_loss, cls=class_loss, cons=cons)


def adam_step(P, m, v, step, lr, eps=1e-6, b1=0.9, b2=0.999):
    """optim.Adam(lr, weight_decay=0, eps=1e-6), main_ucf101.py:416 (torch semantics:
    denom = sqrt(v_hat) + eps with bias corrections)."""
    with torch.no_grad():
        for k, p in P.items():
            if not p.requires_grad or p.grad is None:
                continue
            g = p.grad
            m[k] = b1 * m.get(k, torch.zeros_like(p)) + (1 - b1) * g
            v[k] = b2 * v.get(k, torch.zeros_like(p)) + (1 - b2) * g * g
            bc1 = 1 - b1 ** step
            bc2 = 1 - b2 ** step
            denom = (v[k].sqrt() / np.sqrt(bc2)) + eps
            p.add_(-(lr / bc1) * (m[k] / denom))
