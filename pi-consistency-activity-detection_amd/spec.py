"""Architecture tables for the hot path: I3D trunk (to Mixed_4f) + capsule head + decoder.

Pure-python facts about the network the reference builds
(/root/reference/models/pytorch_i3d.py:221-281 and
/root/reference/models/capsules_ucf101.py:343-384), expressed as data so the plan builder,
the synthetic initialiser and the oracle all read one table.  Parameter names are the
reference's state_dict keys (SURVEY.md §5 "Checkpoint / resume": 293 entries).
"""
from collections import OrderedDict

# (name, kind, ...) in forward order.  conv: (cin, cout, kernel(t,h,w), stride(t,h,w))
# pool: (kernel, stride); mixed: (cin, [b0, b1a, b1b, b2a, b2b, b3b])
TRUNK = [
    ("Conv3d_1a_7x7", "conv", 3, 64, (7, 7, 7), (2, 2, 2)),
    ("MaxPool3d_2a_3x3", "pool", (1, 3, 3), (1, 2, 2)),
    ("Conv3d_2b_1x1", "conv", 64, 64, (1, 1, 1), (1, 1, 1)),
    ("Conv3d_2c_3x3", "conv", 64, 192, (3, 3, 3), (2, 1, 1)),
    ("MaxPool3d_3a_3x3", "pool", (1, 3, 3), (1, 2, 2)),
    ("Mixed_3b", "mixed", 192, (64, 96, 128, 16, 32, 32)),
    ("Mixed_3c", "mixed", 256, (128, 128, 192, 32, 96, 64)),
    ("MaxPool3d_4a_3x3", "pool", (3, 3, 3), (2, 1, 1)),
    ("Mixed_4b", "mixed", 480, (192, 96, 208, 16, 48, 64)),
    ("Mixed_4c", "mixed", 512, (160, 112, 224, 24, 64, 64)),
    ("Mixed_4d", "mixed", 512, (128, 128, 256, 24, 64, 64)),
    ("Mixed_4e", "mixed", 512, (112, 144, 288, 32, 64, 64)),
    ("Mixed_4f", "mixed", 528, (256, 160, 320, 32, 128, 128)),
]
TRUNK_OUT_CH = 832
BN_EPS = 1e-3          # pytorch_i3d.py:80
BN_MOMENTUM = 0.01     # pytorch_i3d.py:80
POSE = 16              # P*P, capsules_ucf101.py:82
IN_CAPS = 32           # capsules_ucf101.py:355
PRIMARY_K = 9          # capsules_ucf101.py:355
EM_ITERS = 3           # capsules_ucf101.py:356
EM_EPS = 1e-8          # capsules_ucf101.py:88
EM_LAMBDA = 1e-6       # capsules_ucf101.py:90
FRAMES = 8             # SURVEY finding 1: the only temporal extent the reference accepts


def mixed_branches(cin, oc):
    """Branch convs of one Inception module: (suffix, cin, cout, kernel)."""
    return [
        ("b0", cin, oc[0], (1, 1, 1)),
        ("b1a", cin, oc[1], (1, 1, 1)),
        ("b1b", oc[1], oc[2], (3, 3, 3)),
        ("b2a", cin, oc[3], (1, 1, 1)),
        ("b2b", oc[3], oc[4], (3, 3, 3)),
        ("b3b", cin, oc[5], (1, 1, 1)),
    ]


def trunk_units():
    """Every Unit3D (conv+BN+ReLU) of the trunk: (state_dict prefix, cin, cout, kernel, stride)."""
    out = []
    for ent in TRUNK:
        if ent[1] == "conv":
            out.append(("conv1." + ent[0], ent[2], ent[3], ent[4], ent[5]))
        elif ent[1] == "mixed":
            for suf, ci, co, k in mixed_branches(ent[2], ent[3]):
                out.append(("conv1.%s.%s" % (ent[0], suf), ci, co, k, (1, 1, 1)))
    return out


def param_shapes(num_classes=24):
    """OrderedDict name -> shape for every *trainable* parameter, reference layout, in the
    order nn.Module.parameters() yields them for the reference CapsNet
    (capsules_ucf101.py:343-384: conv1, primary_caps, conv_caps, upsample1..4, smooth,
    conv28, conv56, conv112)."""
    p = OrderedDict()
    for pre, ci, co, k, _s in trunk_units():
        p[pre + ".conv3d.weight"] = (co, ci) + tuple(k)
        p[pre + ".bn.weight"] = (co,)
        p[pre + ".bn.bias"] = (co,)
    p["primary_caps.pose.weight"] = (IN_CAPS * POSE, TRUNK_OUT_CH, PRIMARY_K, PRIMARY_K)
    p["primary_caps.pose.bias"] = (IN_CAPS * POSE,)
    p["primary_caps.a.weight"] = (IN_CAPS, TRUNK_OUT_CH, PRIMARY_K, PRIMARY_K)
    p["primary_caps.a.bias"] = (IN_CAPS,)
    p["conv_caps.beta_u"] = (num_classes, POSE)
    p["conv_caps.beta_a"] = (num_classes,)
    p["conv_caps.weights"] = (1, IN_CAPS, num_classes, 4, 4)
    p["upsample1.weight"] = (num_classes * POSE, 64, 9, 9)
    p["upsample1.bias"] = (64,)
    p["upsample2.weight"] = (128, 64, 3, 3, 3)
    p["upsample2.bias"] = (64,)
    p["upsample3.weight"] = (128, 64, 3, 3, 3)
    p["upsample3.bias"] = (64,)
    p["upsample4.weight"] = (128, 128, 3, 3, 3)
    p["upsample4.bias"] = (128,)
    p["smooth.weight"] = (128, 1, 3, 3, 3)
    p["smooth.bias"] = (1,)
    p["conv28.weight"] = (64, TRUNK_OUT_CH, 3, 3)
    p["conv28.bias"] = (64,)
    p["conv56.weight"] = (64, 192, 3, 3, 3)
    p["conv56.bias"] = (64,)
    p["conv112.weight"] = (64, 64, 3, 3, 3)
    p["conv112.bias"] = (64,)
    return p


def buffer_shapes():
    """BN buffers (running stats) in state_dict order per Unit3D."""
    b = OrderedDict()
    for pre, _ci, co, _k, _s in trunk_units():
        b[pre + ".bn.running_mean"] = (co,)
        b[pre + ".bn.running_var"] = (co,)
        b[pre + ".bn.num_batches_tracked"] = ()
    return b


def state_dict_keys(num_classes=24):
    """All 293 reference state_dict keys in the reference's order."""
    keys = []
    for pre, _ci, _co, _k, _s in trunk_units():
        keys += [pre + ".conv3d.weight", pre + ".bn.weight", pre + ".bn.bias",
                 pre + ".bn.running_mean", pre + ".bn.running_var",
                 pre + ".bn.num_batches_tracked"]
    keys += [k for k in param_shapes(num_classes) if not k.startswith("conv1.")]
    return keys


def same_pad(size, k, s):
    """TF-'SAME' dynamic padding of the reference's Unit3D / MaxPool3dSamePadding
    (pytorch_i3d.py:15-19, 82-86, 102-107): returns (front, back)."""
    total = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return total // 2, total - total // 2


def same_out(size, k, s):
    f, b = same_pad(size, k, s)
    return (size + f + b - k) // s + 1
