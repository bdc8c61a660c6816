"""Oracle: attentive masks and losses (test infrastructure).

Masks follow /root/reference/utils/helpers.py: measure_pixelwise_var_v2 :8-67 and
measure_pixelwise_gradient :70-95 (host numpy, float32 variance / gradient, float64 fold).
Losses follow /root/reference/utils/losses.py: SpreadLoss :14-37, DiceLoss :44-57,
weighted_mse_loss :74-76, and nn.BCEWithLogitsLoss(mean) (main_ucf101.py:390).
"""
import numpy as np
import torch
import torch.nn.functional as F


def var_mask(pred, flip_pred, frames_cnt=5, use_sig_output=False):
    """helpers.py:8-67.  pred, flip_pred (B,1,8,H,W) -> float64 tensor (B,1,8,H,W).
    Per clip: cyclic 14-frame sequence cat(pred[0:8], flip_pred[1:7]) (:29); population
    variance over a cyclic window of 3 or 5 frames (:33-50, float32 np.var); fold the 14
    variances onto 8 frames (:53-57); per-clip min-max normalise (:59-61)."""
    assert frames_cnt in (3, 5)
    half = frames_cnt // 2
    if use_sig_output:
        pred, flip_pred = torch.sigmoid(pred), torch.sigmoid(flip_pred)
    p = pred.detach().cpu().numpy()
    q = flip_pred.detach().cpu().numpy()
    B = p.shape[0]
    out = np.zeros((B, 1) + p.shape[2:], np.float64)
    for z in range(B):
        cyc = np.concatenate([p[z, 0], q[z, 0, 1:7]], axis=0)          # (14,H,W) float32
        V = np.zeros(cyc.shape, np.float64)
        for t in range(14):
            idx = [(t + k) % 14 for k in range(-half, half + 1)]
            V[t] = np.var(cyc[idx], axis=0)
        M = np.empty((8,) + cyc.shape[1:], np.float64)
        M[0] = 2 * V[0]
        M[7] = 2 * V[7]
        for k in range(1, 7):
            M[k] = V[k] + V[14 - k]
        M -= M.min()
        M /= (M.max() - M.min() + 1e-7)
        out[z, 0] = M
    return torch.from_numpy(out)


def _grad_t(x):
    """np.gradient along axis 0 with unit spacing (second-order interior, first-order edges)."""
    g = np.empty_like(x)
    g[0] = x[1] - x[0]
    g[-1] = x[-1] - x[-2]
    g[1:-1] = (x[2:] - x[:-2]) / 2.0
    return g


def grad_mask(pred, conf_thresh_lower=None, conf_thresh_upper=None):
    """helpers.py:70-95.  pred (B,1,8,H,W) -> float64 tensor (B,8,H,W) (no channel dim).
    sigmoid, optional clamps (:82-85), np.gradient twice along T (:87, float32),
    per-clip min-max normalise (:88-89)."""
    s = torch.sigmoid(pred.detach()).cpu().numpy()
    B = s.shape[0]
    out = np.zeros((B,) + s.shape[2:], np.float64)
    for z in range(B):
        c = s[z, 0].copy()
        if conf_thresh_lower is not None:
            c[c < conf_thresh_lower] = 0
        if conf_thresh_upper is not None:
            c[c > conf_thresh_upper] = 1
        g = _grad_t(_grad_t(c))
        g -= g.min()
        g /= (g.max() - g.min() + 1e-7)
        out[z] = g
    return torch.from_numpy(out)


def weighted_mse(inp, target, weight):
    """losses.py:74-76 (numpy-style broadcasting kept: SURVEY finding 5-i)."""
    return (weight * (inp - target) ** 2).mean()


def dice_loss(logits, targets, smooth=1):
    """losses.py:44-57 (joint over all elements passed in)."""
    s = torch.sigmoid(logits).reshape(-1)
    t = targets.reshape(-1)
    inter = (s * t).sum()
    return 1 - (2.0 * inter + smooth) / (s.sum() + t.sum() + smooth)


def bce_logits(logits, targets):
    """nn.BCEWithLogitsLoss(size_average=True), main_ucf101.py:390."""
    return F.binary_cross_entropy_with_logits(logits, targets)


def spread_loss(x, target, m_min=0.2, m_max=0.9):
    """losses.py:14-37: margin fixed at m_min (r = 0, :15,21); `loss` divides by b twice
    (:34-35); returns (loss, absloss)."""
    b, E = x.shape
    margin = m_min + (m_max - m_min) * 0
    at = x.gather(1, target.long().view(b, 1)).expand(b, E)
    absl = torch.clamp(0.9 - (at - x), min=0) ** 2
    l = torch.clamp(margin - (at - x), min=0) ** 2
    absloss = absl.sum() / b - 0.9 ** 2
    loss = (l.sum() / b - margin ** 2) / b
    return loss, absloss


def exp_rampup(rampup_length):
    """/root/reference/utils/ramp_ups.py:15-24."""
    def f(epoch):
        if epoch < rampup_length:
            e = float(np.clip(epoch, 0.0, rampup_length))
            ph = 1.0 - e / rampup_length
            return float(np.exp(-5.0 * ph * ph))
        return 1.0
    return f
