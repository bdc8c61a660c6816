#!/usr/bin/env python3
"""PCIe-inclusive step rate with the device input pipeline: every step prepares its 8 samples from decoded uint8 videos held in
host memory (picons_amd.inputpipe.get_item: host decisions + upload of the 8 selected frames + pc_clip_from_u8), stages them and
runs the full train step.  Compare with bench.py, whose inputs are resident in HBM before the timed region.

    python tools/bench_step_u8.py [steps]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import picons_amd  # noqa: F401,E402
from picons_amd import inputpipe, step as pstep, synthetic  # noqa: E402


def collate(samples):
    return {'data': torch.stack([s['data'] for s in samples]), 'aug_data': torch.stack([s['aug_data'] for s in samples]),
            'loc_msk': torch.stack([s['loc_msk'] for s in samples]), 'action': torch.stack([s['action'] for s in samples]),
            'label_vid': torch.tensor([s['label_vid'] for s in samples])}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    bs = 8
    args = pstep.default_args(bv=True, gv=False, n_frames=5, wt_cons=0.1, lr=1e-4, epochs=100, thresh_epoch=11)
    eng = pstep.StepEngine(args, bs=bs, hw=224, num_classes=24)
    vids = [synthetic.make_decoded_video(100 + i, labeled=(i % 8) < 4) for i in range(16)]      # decoded videos in host memory
    np.random.seed(0)
    ramp = pstep.exp_rampup(100)(1)
    g = torch.Generator().manual_seed(0)

    def one(i):
        lab = collate([inputpipe.get_item(*vids[(8 * i + j) % 16], train=True) for j in range(4)])
        unl = collate([inputpipe.get_item(*vids[(8 * i + 4 + j) % 16], train=True) for j in range(4)])
        perm = torch.randperm(bs, generator=g).numpy()
        drops = [(torch.rand(bs, c, generator=g) < 0.5).float().numpy() * 2 for c in (832, 128, 832, 128)]
        return eng.train_step(lab, unl, 1, ramp, perm, drops)
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = one(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3

    # the same with the next step's samples prepared on a side stream while the current step runs (the step's own
    # read-back stays where it is: after Adam)
    side = torch.cuda.Stream()

    def prep(i):
        with torch.cuda.stream(side):
            lab = collate([inputpipe.get_item(*vids[(8 * i + j) % 16], train=True) for j in range(4)])
            unl = collate([inputpipe.get_item(*vids[(8 * i + 4 + j) % 16], train=True) for j in range(4)])
            ev = torch.cuda.Event(); ev.record(side)
        perm = torch.randperm(bs, generator=g).numpy()
        drops = [(torch.rand(bs, c, generator=g) < 0.5).float().numpy() * 2 for c in (832, 128, 832, 128)]
        return lab, unl, perm, drops, ev
    nxt = prep(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        lab, unl, perm, drops, ev = nxt
        torch.cuda.current_stream().wait_event(ev)
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(1, ramp)
        eng.adam(args.lr, 1.0)
        nxt = prep(i + 1)                       # host decisions + uploads + pc_clip_from_u8 overlap the step enqueued above
        out2 = eng.read_scalars()               # the step's one host sync
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"metric": "train clips/sec incl. input preparation from host uint8 frames (bs=8)", "value": bs / ms * 1e3, "unit": "clips/s",
                      "ms_per_step": ms, "steps": steps, "loss_total": out["total"],
                      "overlapped": {"value": bs / ms2 * 1e3, "unit": "clips/s", "ms_per_step": ms2, "loss_total": out2["total"],
                                     "how": "next step's samples prepared on a side stream between Adam's enqueue and the loss read-back"}}))


if __name__ == "__main__":
    main()
