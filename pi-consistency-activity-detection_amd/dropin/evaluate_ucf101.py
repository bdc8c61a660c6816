#!/usr/bin/env python3
"""Drop-in for /root/reference/evaluate_ucf101.py (and, with PICONS_DATASET=jhmdb, evaluate_jhmdb.py): same flags
(`--ckpt`, `--seed`, :28-31), same per-checkpoint loop (`best_model_<split>*.pth`, :48-55), same clip construction,
thresholds, printed line (:79-186) and the same pruning of all but the best f-mAP / v-mAP checkpoints (:190-204) - with
the network on the HIP kernels and the per-frame IoU / per-class accumulation on the device
(picons_amd.evalmetrics, csrc/evalmetrics.hip) instead of numpy on the host.

Environment only (the CLI is unchanged):
  PICONS_SYNTHETIC=1      synthetic videos (the dataset / decoder libraries are not on the box); 0 imports the caller's
                          datasets.ucf_dataloader_eval.UCF101DataLoader from PYTHONPATH
  PICONS_EVAL_VIDEOS=<n>  number of synthetic videos (default 4)
  PICONS_EVAL_PACK=1      clips of consecutive videos share full batches (same tables, ~2x the clips/s on short videos)
  PICONS_KEEP_CKPTS=1     do not delete the checkpoints that are neither best f-mAP nor best v-mAP
  PICONS_DATASET=jhmdb    21 classes (evaluate_jhmdb.py:45)
"""
import argparse
import glob
import os
import os.path as osp
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _bootstrap  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402

from picons_amd import evalmetrics, synthetic  # noqa: E402


def _videos(n_classes):
    if os.environ.get("PICONS_SYNTHETIC", "1") == "1":
        return synthetic.make_eval_videos(int(os.environ.get("PICONS_EVAL_VIDEOS", "4")), num_classes=n_classes)
    from datasets.ucf_dataloader_eval import UCF101DataLoader          # the caller's loader (needs skvideo / the dataset)
    ds = UCF101DataLoader('validation', [224, 224], 1, file_id="testing_annots.pkl", use_random_start_frame=False)
    return (ds[i] for i in range(len(ds)))


def iou(split, argv=None):
    """Accuracy, f-mAP and v-mAP over the test set for every `best_model_<split>*.pth` under --ckpt."""
    parser = argparse.ArgumentParser(description='evaluation')
    parser.add_argument('--ckpt', type=str, help='experiment name')
    parser.add_argument('--seed', type=int, default=47, help='seed for initializing training.')
    args = parser.parse_args(argv)
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)

    synthetic_mode = os.environ.get("PICONS_SYNTHETIC", "1") == "1"      # read only: the process environment is not written to
    jhmdb = os.environ.get("PICONS_DATASET", "ucf101") == "jhmdb"
    n_classes = 21 if jhmdb else 24
    if jhmdb:
        from models.capsules_jhmdb_semi_sup_pa import CapsNet
    else:
        from models.capsules_ucf101 import CapsNet
    # every checkpoint loaded below overwrites the whole state, so in synthetic mode a missing rgb_charades.pt (no network) is passed
    # over EXPLICITLY with pt_path=None; with real data CapsNet() raises for a missing trunk file like the reference does
    pt_path = '../weights/rgb_charades.pt'
    model = (CapsNet(pt_path=None) if (synthetic_mode and not os.path.exists(pt_path)) else CapsNet()).cuda()
    clip_batch_size = 14
    model_names, fmap_best, vmap_best, results = [], [], [], []
    files = sorted(glob.glob(osp.join(args.ckpt, 'best_model_' + split + '*.pth')))
    for saved_wts in files:
        model.load_previous_weights(saved_wts)
        model_names.append(saved_wts)
        model.eval()
        model.training = False
        r = evalmetrics.evaluate(model, _videos(n_classes), n_classes=n_classes, clip_batch_size=clip_batch_size,
                                 pack=os.environ.get("PICONS_EVAL_PACK", "0") == "1").result()
        thr = np.arange(0, 20, dtype=np.float32) / 20
        print('Accuracy:', r["accuracy"], 'IoU/fmap/vmap', thr[4], r["fmAP"][4], r["vmAP"][4], thr[10], r["fmAP"][10], r["vmAP"][10])
        fmap_best.append(r["fmAP"][10]); vmap_best.append(r["vmAP"][10]); results.append(r)
    if not files:
        return results
    best = {model_names[fmap_best.index(max(fmap_best))], model_names[vmap_best.index(max(vmap_best))]}
    if os.environ.get("PICONS_KEEP_CKPTS", "0") != "1":
        for f in files:
            if f not in best:
                os.remove(f)
    print(os.listdir(args.ckpt))
    return results


if __name__ == '__main__':
    iou('train')
