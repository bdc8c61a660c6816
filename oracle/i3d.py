"""Oracle: I3D trunk up to Mixed_4f (test infrastructure, see oracle/__init__.py).

Follows /root/reference/models/pytorch_i3d.py: Unit3D.forward :89-120 (TF-SAME pad, conv
without bias, BatchNorm3d(eps 1e-3, momentum 0.01), ReLU), MaxPool3dSamePadding.forward :21-45
(zero F.pad then max-pool), InceptionModule.forward :144-149, InceptionI3d.forward :328-346
(taps out56 after Conv3d_2c, out112 after Conv3d_1a).
"""
import torch
import torch.nn.functional as F

import picons_amd.spec as spec


def _pad6(shape_thw, k, s):
    p = [spec.same_pad(shape_thw[d], k[d], s[d]) for d in range(3)]
    # F.pad order: last dim first  (pytorch_i3d.py:109)
    return (p[2][0], p[2][1], p[1][0], p[1][1], p[0][0], p[0][1])


def unit3d(P, pre, x, stride, training, bn_state=None):
    """pytorch_i3d.py:89-120.  P maps state_dict keys to tensors; running stats are updated
    in place when training (nn.BatchNorm3d semantics, :80)."""
    w = P[pre + ".conv3d.weight"]
    x = F.pad(x, _pad6(x.shape[2:], w.shape[2:], stride))
    z = F.conv3d(x, w, None, stride)
    y = F.batch_norm(z, P[pre + ".bn.running_mean"], P[pre + ".bn.running_var"],
                     P[pre + ".bn.weight"], P[pre + ".bn.bias"], training,
                     spec.BN_MOMENTUM, spec.BN_EPS)
    if training and (pre + ".bn.num_batches_tracked") in P:
        P[pre + ".bn.num_batches_tracked"] += 1
    return F.relu(y)


def maxpool_same(x, k, s):
    """pytorch_i3d.py:21-45: zero-pad (F.pad constant 0) then MaxPool3d(padding=0)."""
    x = F.pad(x, _pad6(x.shape[2:], k, s))
    return F.max_pool3d(x, k, s)


def inception(P, pre, x, oc, training):
    """pytorch_i3d.py:144-149."""
    b0 = unit3d(P, pre + ".b0", x, (1, 1, 1), training)
    b1 = unit3d(P, pre + ".b1b", unit3d(P, pre + ".b1a", x, (1, 1, 1), training), (1, 1, 1), training)
    b2 = unit3d(P, pre + ".b2b", unit3d(P, pre + ".b2a", x, (1, 1, 1), training), (1, 1, 1), training)
    b3 = unit3d(P, pre + ".b3b", maxpool_same(x, (3, 3, 3), (1, 1, 1)), (1, 1, 1), training)
    return torch.cat([b0, b1, b2, b3], dim=1)


def trunk(P, x, training, prefix="conv1."):
    """pytorch_i3d.py:328-346 -> (Mixed_4f, out56, out112)."""
    out56 = out112 = None
    for ent in spec.TRUNK:
        name = prefix + ent[0]
        if ent[1] == "conv":
            x = unit3d(P, name, x, ent[5], training)
        elif ent[1] == "pool":
            x = maxpool_same(x, ent[2], ent[3])
        else:
            x = inception(P, name, x, ent[3], training)
        if ent[0] == "Conv3d_2c_3x3":
            out56 = x
        if ent[0] == "Conv3d_1a_7x7":
            out112 = x
    return x, out56, out112
