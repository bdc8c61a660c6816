#!/usr/bin/env python3
"""Evaluation path on one MI355X: eval-mode forward of 14-clip batches (the reference's
clip_batch_size, evaluate_ucf101.py:41) + device f-mAP / v-mAP accumulation.  Prints clips/s of the whole loop, the time
of the two metric kernels per batch and the HBM rate of pc_seg_frame_counts (algorithmic bytes: one fp32 logit + one
fp32 truth pixel per pixel, read once) against the 8 TB/s roof.

    python tools/bench_eval.py [n_videos]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import picons_amd  # noqa: F401,E402
from picons_amd import evalmetrics, model as pmodel, ops, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    m = pmodel.CapsNet(pt_path=None, init="conditioned").cuda()
    m.eval(); m.training = False
    vids = synthetic.make_eval_videos(n, seed=5)
    nclips = sum(evalmetrics.make_clips(v, b)[0].shape[0] for v, b, _l in vids)
    evalmetrics.evaluate(m, vids)                          # warm-up: one plan per distinct clips-per-video count, kernels loaded
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = evalmetrics.evaluate(m, vids)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evalmetrics.evaluate(m, vids, pack=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    accp = evalmetrics.evaluate(m, vids, pack=True)
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    same = all((accp.result()[k] == acc.result()[k]).all() for k in ("frame_ious", "video_ious", "n_tot_frames", "n_vids"))
    # metric kernels alone on one full batch
    nb = int(os.environ.get("PICONS_EVAL_BENCH_CLIPS", "14"))
    x = torch.randn(nb, 1, 8, 224, 224, device="cuda"); gt = (torch.rand(nb, 8, 224, 224, device="cuda") < 0.2).float()
    for _ in range(3):
        c = ops.seg_frame_counts(x, gt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = ops.seg_frame_counts(x, gt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = 2 * x.numel() * 4 / 1e9
    print(json.dumps({"metric": "eval clips/sec (bs=14 clips, 8x224x224, eval forward + f-mAP/v-mAP accumulation)", "value": nclips / dt, "unit": "clips/s",
                      "clips": nclips, "videos": n, "clip_building_and_upload_included": True,
                      "packed_batches": {"value": nclips / dtp, "unit": "clips/s", "same_tables": bool(same)},
                      "seg_frame_counts": {"ms_per_batch": ms, "bound": "hbm", "achieved": gb / (ms * 1e-3), "peak": 8000.0, "unit": "GB/s",
                                           "frac": gb / (ms * 1e-3) / 8000.0, "algorithmic_bytes": gb * 1e9},
                      "fmAP@0.5": float(acc.result()["fmAP"][10])}))


if __name__ == "__main__":
    main()
