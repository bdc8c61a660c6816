#!/usr/bin/env python3
"""Diagnostic: one bs = 8, 8x224x224 step of tests/test_step_gpu.py's BS8 cases against the fp64 oracle, element by element for the named
parameter gradients (is a tensor's error spread over its channels, or carried by one or two elements -- a ReLU mask that came out the
other way for a pre-activation within rounding of zero?).
    python tools/probe_tensor_grad.py <case index in BS8> <param name> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_step_gpu as T

tag, akw, stepid, epoch, ncls, jhmdb = T.BS8[int(sys.argv[1])]
eng, ref, P, P64 = T.run_pair(akw, 224, 8, epoch, ncls, jhmdb, stepid=stepid)
print("case", tag, "PICONS_SPLIT", os.environ.get("PICONS_SPLIT", "1"))
for name in sys.argv[2:]:
    g = eng.grad(name).cpu().double().flatten(); r32 = P[name].grad.double().flatten(); r64 = P64[name].grad.flatten()
    d = (g - r64).abs(); d32 = (r32 - r64).abs()
    den = r64.norm().item()
    print("%s: n %d  rel-L2 hip %.3e  cpu32 %.3e; without the two worst elements: hip %.3e  cpu32 %.3e" % (
        name, g.numel(), d.norm().item() / den, d32.norm().item() / den,
        torch.sort(d)[0][:-2].norm().item() / den, torch.sort(d32)[0][:-2].norm().item() / den))
    for i in torch.argsort(d, descending=True)[:6].tolist():
        print("   [%d] hip %.6e  r32 %.6e  r64 %.6e   d %.2e  d32 %.2e" % (i, g[i], r32[i], r64[i], d[i], d32[i]))
