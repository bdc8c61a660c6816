"""Build-owned synthetic evaluation set + stand-in network for the f-mAP / v-mAP tests.

The reference's evaluation loop (evaluate_ucf101.py:73-191) consumes whole videos, cuts them into 8-frame clips, runs
the network and accumulates per-class frame / video IoU hit counts at 20 thresholds.  Its arithmetic is pinned by
running THAT loop (tools/make_eval_golden.py, authoring container only) on the videos below with `FakeNet` standing
in for the network, and recording its accumulators in tests/golden/eval_map.npz.  The same videos and the same
stand-in feed the oracle restatement and the HIP accumulator in the tests.  Everything here is numpy PCG64 /
plain torch: no reference code."""
import numpy as np
import torch

HW = 224
NCLS = 24


def videos(seed=11, ncls=NCLS):
    """-> list of (video [F,HW,HW,3] f32 in [0,1], bbox [F,HW,HW,1] f32 {0,1}, label).  One video per class plus
    three extras: a repeat class, a video whose boxes sit only in the last frames, and a video without any box."""
    rng = np.random.default_rng(seed)
    out = []
    labels = list(range(ncls)) + [3, 17, 5]
    for vi, lab in enumerate(labels):
        F = int(rng.integers(8, 41))
        bbox = np.zeros((F, HW, HW, 1), np.float32)
        if vi == len(labels) - 1:
            pass                                    # no boxes at all: the loop skips the video
        else:
            f0 = int(rng.integers(0, max(1, F - 6)))
            f1 = F if vi == len(labels) - 2 else int(rng.integers(f0 + 3, F + 1))
            if vi == len(labels) - 2:
                f0 = F - 3
            y, x = int(rng.integers(20, 120)), int(rng.integers(20, 120))
            h, w = int(rng.integers(40, 100)), int(rng.integers(40, 100))
            for f in range(f0, f1):
                yy = min(HW - h, max(0, y + (f - f0)))
                xx = min(HW - w, max(0, x + 2 * (f - f0)))
                bbox[f, yy:yy + h, xx:xx + w, 0] = 1.0
        # the clip carries a displaced, partly erased copy of the box, so the stand-in's masks overlap the truth by
        # anything between 0 and 1
        dy, dx = int(rng.integers(-30, 31)), int(rng.integers(-30, 31))
        ghost = np.roll(bbox, (dy, dx), axis=(1, 2)) * (rng.random((F, 1, 1, 1)) < 0.85)
        video = (0.25 * rng.random((F, HW, HW, 3), dtype=np.float32) + 0.6 * ghost).astype(np.float32)
        out.append((video, bbox, lab))
    return out


class FakeNet(torch.nn.Module):
    """Stand-in with the network's eval signature (capsules_ucf101.py:413,512): masks from the clip's brightness, class
    scores from a fixed projection of per-frame means."""

    def __init__(self, ncls=NCLS, seed=5):
        super().__init__()
        g = np.random.default_rng(seed)
        self.register_buffer("proj", torch.from_numpy(g.standard_normal((8 * 3, ncls)).astype(np.float32)))

    def load_previous_weights(self, path):
        pass

    def forward(self, data, classification=None, concat_labels=None, epoch=0, thresh_ep=0):
        data = data.float()
        seg = 10.0 * (data.mean(1, keepdim=True) - 0.40)                      # (B,1,8,H,W) logits
        seg = (torch.floor(seg * 1024.0) + 0.5) / 1024.0      # never within 4e-4 of 0: sigmoid(x) >= 0.5 does not hang on the last ulp of exp
        feat = data.mean((3, 4)).permute(0, 2, 1).reshape(data.shape[0], -1)   # (B, 8*3)
        pred = torch.sigmoid(feat @ self.proj.to(feat.device) * 4.0)
        return seg, pred, None
