"""Drop-in for /root/reference/utils/losses.py (SpreadLoss, DiceLoss, weighted_mse_loss).  These
module-level losses stay composable autograd functions over device tensors (the fused HIP loss kernel
is used by the fused step, picons_amd.step); semantics follow losses.py:14-37, 44-57, 74-76 exactly,
including SpreadLoss's double division by b and weighted_mse_loss's numpy-style broadcasting."""
import torch
import torch.nn as nn
from torch.nn.modules.loss import _Loss


class SpreadLoss(_Loss):
    def __init__(self, m_min=0.2, m_max=0.9, num_class=24):
        super().__init__()
        self.m_min, self.m_max, self.num_class = m_min, m_max, num_class

    def forward(self, x, target):
        b, E = x.shape
        assert E == self.num_class
        margin = self.m_min                      # r = 0 (losses.py:15,21)
        at = x.gather(1, target.long().view(b, 1)).expand(b, E)
        absloss = torch.clamp(.9 - (at - x), min=0) ** 2
        loss = torch.clamp(margin - (at - x), min=0) ** 2
        absloss = absloss.sum() / b - .9 ** 2
        loss = (loss.sum() / b - margin ** 2) / b
        return loss, absloss


class DiceLoss(nn.Module):
    def __init__(self, weight=None, size_average=True):
        super().__init__()

    def forward(self, inputs, targets, smooth=1):
        s = torch.sigmoid(inputs).reshape(-1)
        t = targets.reshape(-1)
        return 1 - (2. * (s * t).sum() + smooth) / (s.sum() + t.sum() + smooth)


def weighted_mse_loss(input, target, weight):
    return (weight * (input - target) ** 2).mean()
