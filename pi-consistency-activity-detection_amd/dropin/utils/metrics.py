"""Host-side logging helpers of /root/reference/utils/metrics.py used by the train/val loops:
get_accuracy :7-13 (argmax match rate) and IOU2 :171-193 (binary masks; NaN when the truth is empty)."""
import numpy as np
import torch


def get_accuracy(predicted_actor, actor):
    _maxm, prediction = torch.max(predicted_actor, 1)
    prediction = prediction.view(-1, 1)
    actor = actor.view(-1, 1).to(prediction.device)
    correct = torch.sum(actor == prediction.float()).item()
    return correct / float(prediction.shape[0])


def IOU2(gt, img):
    gt = np.asarray(gt)
    img = np.asarray(img)
    s = gt + img
    intersection_sum = float((s >= 2).sum())
    union_sum = float(np.minimum(s, 1).sum())
    if gt.sum() > 0:
        return intersection_sum / union_sum
    return float('NaN')
