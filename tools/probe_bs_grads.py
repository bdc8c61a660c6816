#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of one step against the fp64 oracle at an arbitrary batch / frame size (the check of
tests/test_step_gpu.py::check_gradients_fp64_anchored, printing every tensor that is clearly worse than the fp32 oracle).
    python tools/probe_bs_grads.py [bs] [hw] [stepid]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_step_gpu as T

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 112
stepid = int(sys.argv[3]) if len(sys.argv) > 3 else 7
eng, ref, P, P64 = T.run_pair(dict(bv=True, gv=True, n_frames=5, wt_cons=0.1), hw, bs, 1, 24, False, stepid=stepid)
rows = []
for name in eng.plan.pshape:
    g = eng.grad(name).cpu().double(); r32 = P[name].grad.double(); r64 = P64[name].grad
    den = r64.norm().item() + 1e-12
    rows.append((name, (g - r64).norm().item() / den, (r32 - r64).norm().item() / den, den))
print("bs %d hw %d lanes %s" % (bs, hw, os.environ.get("PICONS_LANES", "4")))
for r in rows:
    if r[1] > max(3 * r[2], 3e-3):
        print("%-44s hip %.3e  cpu32 %.3e  |g| %.3e" % r)

if len(sys.argv) > 4:
    for name in sys.argv[4:]:
        g = eng.grad(name).cpu().double().flatten(); r32 = P[name].grad.double().flatten(); r64 = P64[name].grad.flatten()
        d = (g - r64).abs(); d32 = (r32 - r64).abs()
        top = torch.argsort(d, descending=True)[:8]
        print(name, "n", g.numel(), "max|d| %.3e at %s; |r64| max %.3e" % (d.max().item(), top.tolist(), r64.abs().max().item()))
        for i in top.tolist():
            print("   [%d] hip %.6e  r32 %.6e  r64 %.6e   d %.2e  d32 %.2e" % (i, g[i], r32[i], r64[i], d[i], d32[i]))
