#!/usr/bin/env python3
"""Is the step power-limited?  The same step (same kernels, same launches, same bytes) on the ordinary synthetic minibatch and on an all-zero
minibatch: with zero clips every activation is a per-channel constant, the matrix cores and data paths toggle far fewer bits, and the chip holds a
higher clock (MI355X_MICROARCH.md, DVFS give-back).  The ratio of the two step times bounds what lower energy per operation could buy.
    python tools/probe_power.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import picons_amd  # noqa
from picons_amd import step as pstep, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1, lr=0.0, epochs=100, thresh_epoch=11)      # lr 0: the parameters stay what they are
eng = pstep.StepEngine(args, bs=8, hw=224, num_classes=24, device="cuda:0")
ramp = pstep.exp_rampup(100)(1)
lab, unl, perm, drops = synthetic.make_step_inputs(8, rank=0, step=0, num_classes=24)


def timed(tag):
    for _ in range(5):
        eng.run_staged(1, ramp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.run_staged(1, ramp)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print("%-28s %.3f ms/step   loss %.5f" % (tag, ms, out["total"]), flush=True)
    return ms


for rep in range(2):
    eng.stage(lab, unl, perm, drops)
    a = timed("synthetic minibatch")
    z = lambda mb: {k: (np.zeros_like(v) if k in ("data", "aug_data", "loc_msk") else v) for k, v in mb.items()}
    eng.stage(z(lab), z(unl), perm, drops)
    b = timed("all-zero clips and masks")
    print("ratio %.3f" % (a / b))
