#!/usr/bin/env python3
"""Diagnostic: does a bench leg leave something behind that slows the next one?  One engine, the legs of bench.py in several orders, 50 steps each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from picons_amd import step as pstep, synthetic
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
eng = pstep.StepEngine(args, bs=8, hw=224)
ramp = pstep.exp_rampup(100)(1)
lab, unl, perm, drops = synthetic.make_step_inputs(8, rank=0, step=0)


def resident(n=50):
    eng.stage(lab, unl, perm, drops)
    for _ in range(3):
        eng.run_staged(1, ramp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.run_staged(1, ramp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


si = di = None
def staged(n=50):
    global si
    si = si or bench.StagedInputs(eng, 8, 24, 0)
    si.run(3, 1, ramp, None, 1e-4)
    sec, _o, _n = si.run(n, 1, ramp, None, 1e-4)
    return sec / n * 1e3


def dicts(n=50):
    global di
    di = di or bench.DictInputs(eng, 8, 24, 0)
    di.run(3, 1, ramp, None, 1e-4)
    sec, _o, _w = di.run(n, 1, ramp, None, 1e-4)
    return sec / n * 1e3


for name in sys.argv[1:] or ["resident", "staged", "resident", "dicts", "resident", "staged", "dicts", "resident"]:
    print("%-9s %.3f ms/step" % (name, dict(resident=resident, staged=staged, dicts=dicts)[name]()), flush=True)
