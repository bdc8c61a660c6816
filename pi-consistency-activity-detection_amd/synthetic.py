"""Deterministic synthetic weights and UCF101-24 / JHMDB-21 shaped minibatches.

Host-side, numpy PCG64 only (no torch RNG), so the oracle, the reference (in the authoring
container) and the HIP path can all be fed bit-identical inputs.  SURVEY.md §8(c) "Weights
for parity" / §8(d) "Synthetic inputs".  The real `rgb_charades.pt` is a download
(/root/reference/README.md:48-52) and there is no network, so weights are synthetic.
"""
import zlib
from collections import OrderedDict

import numpy as np

from . import spec


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), int(seed)]))


def init_state(seed=47, num_classes=24, conditioned=True):
    """name -> float32 ndarray in the reference's layout for all 293 state_dict entries.

    conditioned=True uses the init SURVEY finding 4 recommends for tight parity bars
    (fan-in scaled convs, primary_caps std 0.01, conv_caps.weights scale 0.5);
    conditioned=False reproduces the reference's own std choices
    (capsules_ucf101.py:36,39,97-103,359-374) with synthetic draws.
    """
    sd = OrderedDict()
    shapes = spec.param_shapes(num_classes)
    for name, shp in shapes.items():
        g = _rng(name, seed)
        n = int(np.prod(shp))
        if name.endswith(".bn.weight"):
            v = g.uniform(0.5, 1.5, n)
        elif name.endswith(".bn.bias"):
            v = g.normal(0.0, 0.1, n)
        elif name.endswith(".bias"):
            v = g.normal(0.0, 0.05, n)
        elif name.startswith("primary_caps"):
            v = g.normal(0.0, 0.01 if conditioned else 0.1, n)
        elif name == "conv_caps.weights":
            v = g.normal(0.0, 0.5 if conditioned else 1.0, n)
        elif name in ("conv_caps.beta_u", "conv_caps.beta_a"):
            v = g.normal(0.0, 1.0, n)
        elif name.startswith("upsample") or name.startswith("smooth"):
            v = g.normal(0.0, 0.02, n)
        else:  # conv weights: fan-in scaled
            fan_in = int(np.prod(shp[1:]))
            v = g.normal(0.0, np.sqrt(2.0 / fan_in), n)
        sd[name] = v.astype(np.float32).reshape(shp)
    out = OrderedDict()
    for key in spec.state_dict_keys(num_classes):
        if key in sd:
            out[key] = sd[key]
        elif key.endswith("running_mean"):
            out[key] = np.zeros(spec.buffer_shapes()[key], np.float32)
        elif key.endswith("running_var"):
            out[key] = np.ones(spec.buffer_shapes()[key], np.float32)
        else:
            out[key] = np.zeros((), np.int64)
    return out


def init_state_module(seed=47, num_classes=24):
    """What constructing the reference's CapsNet leaves in its parameters (synthetic draws of the same distributions):
    PyTorch's default Conv / ConvTranspose init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases, BatchNorm 1 / 0
    (pytorch_i3d.py:69-80), PrimaryCaps weights N(0, 0.1) (capsules_ucf101.py:36,39), ConvCaps beta_u / beta_a / weights
    randn (:97-103), upsample1-4 / smooth weights N(0, 0.02) (:359-374).  This is the drop-in CapsNet's initial state; the
    trunk is then overwritten by rgb_charades.pt when that file exists (:343-352)."""
    sd = OrderedDict()
    shapes = spec.param_shapes(num_classes)
    for name, shp in shapes.items():
        g = _rng(name, seed + 7919)
        n = int(np.prod(shp))
        wname = name.rsplit(".", 1)[0] + ".weight"
        if name.endswith(".bn.weight"):
            v = np.ones(n)
        elif name.endswith(".bn.bias"):
            v = np.zeros(n)
        elif name in ("conv_caps.weights", "conv_caps.beta_u", "conv_caps.beta_a"):
            v = g.normal(0.0, 1.0, n)
        elif name in ("primary_caps.pose.weight", "primary_caps.a.weight"):
            v = g.normal(0.0, 0.1, n)
        elif name.endswith(".weight") and (name.startswith("upsample") or name.startswith("smooth")):
            v = g.normal(0.0, 0.02, n)
        else:       # default-initialised conv weights and all conv biases: fan_in = weight.shape[1] * receptive field
            wshp = shapes[wname] if wname in shapes else shp
            bound = 1.0 / np.sqrt(int(np.prod(wshp[1:])))
            v = g.uniform(-bound, bound, n)
        sd[name] = v.astype(np.float32).reshape(shp)
    out = OrderedDict()
    for key in spec.state_dict_keys(num_classes):
        if key in sd:
            out[key] = sd[key]
        elif key.endswith("running_mean"):
            out[key] = np.zeros(spec.buffer_shapes()[key], np.float32)
        elif key.endswith("running_var"):
            out[key] = np.ones(spec.buffer_shapes()[key], np.float32)
        else:
            out[key] = np.zeros((), np.int64)
    return out


def make_minibatch(n, labeled, seed, num_classes=24, hw=224, frames=spec.FRAMES):
    """One dataloader-shaped dict (SURVEY §8(b) minibatch contract;
    /root/reference/datasets/ucf_dataloader.py:179-191): float64 `data`/`aug_data`
    (n,3,T,H,W) in [0,1) with aug_data = flip(data, W); `loc_msk` one random box per clip;
    `action` (n,1) float32; `label_vid` (n,) int64."""
    g = np.random.default_rng(seed)
    data = g.random((n, 3, frames, hw, hw), dtype=np.float64)
    msk = np.zeros((n, 1, frames, hw, hw), np.float64)
    lo, hi = max(2, hw * 40 // 224), max(3, hw * 160 // 224)
    for i in range(n):
        h = int(g.integers(lo, hi + 1)); w = int(g.integers(lo, hi + 1))
        y0 = int(g.integers(0, hw - h + 1)); x0 = int(g.integers(0, hw - w + 1))
        msk[i, 0, :, y0:y0 + h, x0:x0 + w] = 1.0
    action = g.integers(0, num_classes, (n, 1)).astype(np.float32)
    return {
        "data": data,
        "aug_data": np.ascontiguousarray(data[..., ::-1]),
        "loc_msk": msk,
        "action": action,
        "label_vid": (np.ones if labeled else np.zeros)((n,), np.int64),
    }


def make_step_inputs(bs, rank=0, step=0, num_classes=24, hw=224):
    """Labeled + unlabeled dicts (bs/2 each, main_ucf101.py:353-366), the shuffle permutation
    (main_ucf101.py:73) and the four Dropout3d keep-masks of one step
    (capsules_ucf101.py:428,507, two passes), all from PCG64(1234+rank, step)."""
    n = bs // 2
    base = 1234 + rank + 1000003 * step
    lab = make_minibatch(n, True, base * 2 + 0, num_classes, hw)
    unl = make_minibatch(n, False, base * 2 + 1, num_classes, hw)
    g = np.random.default_rng(base * 2 + 7)
    perm = g.permutation(bs)
    drops = [
        (g.random((bs, spec.TRUNK_OUT_CH)) < 0.5).astype(np.float32) * 2.0,
        (g.random((bs, 128)) < 0.5).astype(np.float32) * 2.0,
        (g.random((bs, spec.TRUNK_OUT_CH)) < 0.5).astype(np.float32) * 2.0,
        (g.random((bs, 128)) < 0.5).astype(np.float32) * 2.0,
    ]
    return lab, unl, perm, drops


def make_eval_videos(n, seed=1234, num_classes=24, hw=224):
    """Synthetic evaluation set in the format the reference's eval loader yields (datasets/ucf_dataloader_eval.py via
    evaluate_ucf101.py:74-77): (video [F,hw,hw,3] float32 in [0,1], bbox [F,hw,hw,1] float32 {0,1}, label), F = 8..40, one
    moving box per video over a sub-range of its frames."""
    rng = np.random.default_rng(seed)
    out = []
    for vi in range(n):
        F = int(rng.integers(8, 41))
        bbox = np.zeros((F, hw, hw, 1), np.float32)
        f0 = int(rng.integers(0, max(1, F - 6))); f1 = int(rng.integers(f0 + 3, F + 1))
        h, w = int(rng.integers(hw // 6, hw // 2)), int(rng.integers(hw // 6, hw // 2))
        y, x = int(rng.integers(0, hw - h)), int(rng.integers(0, hw - w))
        for f in range(f0, f1):
            yy, xx = min(hw - h, y + (f - f0)), min(hw - w, x + 2 * (f - f0))
            bbox[f, yy:yy + h, xx:xx + w, 0] = 1.0
        video = rng.random((F, hw, hw, 3), dtype=np.float32)
        out.append((video, bbox, vi % num_classes))
    return out


def make_decoded_video(seed, labeled=True, num_classes=24, frames_hw=(240, 320)):
    """One synthetic decoded training video in the form datasets/ucf_dataloader.py `load_video` consumes after `vread`:
    (uint8 frames [F,H,W,3], annotations [(start, end, label, [[x,y,w,h] per frame], [annotated frame ids], labeled_vid)])."""
    rng = np.random.default_rng(seed)
    F = int(rng.integers(32, 48))
    H, W = frames_hw
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    s, e = int(rng.integers(0, 4)), int(rng.integers(F - 6, F - 1))
    x, y = int(rng.integers(0, W - 160)), int(rng.integers(0, H - 140))
    bw, bh = int(rng.integers(40, 140)), int(rng.integers(40, 120))
    boxes = [[min(W - bw, x + (f - s)), min(H - bh, y + (f - s) // 2), bw, bh] for f in range(s, e + 1)]
    annot = sorted(int(v) for v in rng.choice(np.arange(s + 4, e - 4), size=3, replace=False))
    return frames, [(s, e, int(rng.integers(0, num_classes)), boxes, annot, 1 if labeled else 0)]
