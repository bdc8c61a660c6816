#!/usr/bin/env python3
"""Input pipeline on one MI355X: samples/s of picons_amd.inputpipe.get_item (host decisions + upload of
the 8 selected uint8 frames + pc_clip_from_u8) and the kernel alone on frames resident in HBM, with its HBM rate
(algorithmic bytes per clip: 8*224*224 * (3 read + 28 written) = 12.4 MB) against the 8 TB/s roof.

    python tools/bench_input.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import picons_amd  # noqa: F401,E402
from picons_amd import inputpipe, ops  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (40, 240, 320, 3), dtype=np.uint8)
    boxes = [[40 + f, 30 + f, 90, 80] for f in range(31)]
    ann = [(5, 35, 7, boxes, [12, 20, 25], 1)]
    np.random.seed(0)
    for _ in range(5):
        inputpipe.get_item(frames, ann, True)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        s = inputpipe.get_item(frames, ann, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    dev = torch.from_numpy(frames).cuda()
    rects = torch.zeros(8, 1, 4, dtype=torch.int32, device="cuda")
    span = list(range(4, 20, 2))
    for _ in range(5):
        ops.clip_from_u8(dev, span, 8, 48, rects)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.clip_from_u8(dev, span, 8, 48, rects)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    nbytes = 8 * 224 * 224 * (3 + 28)
    print(json.dumps({"metric": "input samples/sec (uint8 frames -> fp32 NCDHW clip + flipped clip + mask)", "value": 1.0 / dt, "unit": "samples/s",
                      "ms_per_sample_incl_upload": dt * 1e3,
                      "clip_from_u8": {"ms": ms, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                       "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": nbytes,
                                       "note": "one clip per launch (12.4 MB): launch-bound"}}))


if __name__ == "__main__":
    main()
