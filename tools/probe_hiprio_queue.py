#!/usr/bin/env python3
"""Diagnostic: what does work on a HIGH-PRIORITY stream (what torch.distributed's NCCL backend uses for its collectives: a fifth
hardware queue, from ROCm's separate high-priority pool) do to the four-lane step?
    python tools/probe_hiprio_queue.py [none|once|step|step_normal]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import picons_amd
from picons_amd import step as pstep, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1, lr=1e-4)
eng = pstep.StepEngine(args, bs=8, hw=224)
eng.stage(*synthetic.make_step_inputs(8, rank=0, step=0))
ramp = pstep.exp_rampup(100)(1)
hp = torch.cuda.Stream(priority=-1) if mode in ("once", "step") else torch.cuda.Stream()
buf = torch.zeros(36_000_000, device="cuda")        # 147 MB: the size of the big gradient bucket


def poke():
    with torch.cuda.stream(hp):
        buf.add_(1.0)


if mode == "once":
    poke(); torch.cuda.synchronize()
for it in range(5):
    eng.run_staged(1, ramp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(60):
    eng.run_staged(1, ramp)
    if mode in ("step", "step_normal"):
        poke()
torch.cuda.synchronize()
print(mode, "ms/step %.3f" % ((time.perf_counter() - t0) / 60 * 1e3))
