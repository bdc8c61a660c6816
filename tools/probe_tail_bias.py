#!/usr/bin/env python3
"""Repeat the bias + ReLU comparison of tests/test_x6_gpu.py::test_x6_tail_split_matches_and_is_deterministic (split against unsplit launch) and print
the distance in units of the test's tolerance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from picons_amd import capi, desc, ops, spec
from tests.test_kernels_gpu import cl, w_oki
DEV = "cuda"
for Ci, Co, k, thw, N in [(64, 128, (3, 3, 3), (2, 28, 28), 8), (64, 128, (3, 3, 3), (4, 28, 28), 32)]:
    g = torch.Generator().manual_seed(41)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g)))
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    pf = [p[0] for p in pads]
    xg, wk = cl(x), w_oki(w)
    wp = ops.split_planes(wk)
    bias = torch.randn(Co, generator=g).to(DEV)
    d3 = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw, act=capi.ACT_RELU, flags=capi.F_BIAS)
    d0 = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw)
    res = []
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        r0 = ops.conv_fwd_x6(d3, xg, wp, torch.empty(N, *thw, Co, device=DEV), bias=bias)
        ws = torch.zeros(ops.conv_x6_ws_floats(d3), device=DEV)
        r1 = ops.conv_fwd_x6(d3, xg, wp, torch.empty(N, *thw, Co, device=DEV), bias=bias, ws=ws)
        p0 = ops.conv_fwd_x6(d0, xg, wp, torch.empty(N, *thw, Co, device=DEV))
        p1 = ops.conv_fwd_x6(d0, xg, wp, torch.empty(N, *thw, Co, device=DEV), ws=torch.zeros(ops.conv_x6_ws_floats(d0), device=DEV))
        ref = torch.relu(p0 + bias)
        b0, b1 = int((r0 != ref).sum()), int(((r1 - ref).abs() > 2e-6 * ref.abs().max()).sum())
        if b0 or b1 or int((p0 != res[0][4]).sum() if res else 0):
            print("  iteration %d: unsplit bias+relu launch differs from relu(plain + bias) in %d elements; split launch beyond tolerance in %d; plain launch differs from its first run in %d"
                  % (it, b0, b1, int((p0 != res[0][4]).sum()) if res else 0))
            bad = ((r1 - ref).abs() > 2e-6 * ref.abs().max()).nonzero()
            if len(bad):
                print("    first / last bad index of the split launch (n, t, h, w, c):", bad[0].tolist(), bad[-1].tolist(), "values", float(r1[tuple(bad[0])]), float(ref[tuple(bad[0])]))
        res.append(((r0 - r1).abs().max().item() / (2e-6 * r0.abs().max().item()), (p0 - p1).abs().max().item() / (2e-6 * p0.abs().max().item()),
                    int((r0 != r1).sum()), int((p0 != p1).sum()), p0.clone() if not res else None))
    print((Ci, Co, k, thw, N), "bias+relu / plain distance in tolerances, differing elements:", ["%.2f/%.2f %d/%d" % r[:4] for r in res[:3]], "max %.2f / %.2f" % (max(r[0] for r in res), max(r[1] for r in res)))
