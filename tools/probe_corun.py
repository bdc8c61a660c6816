#!/usr/bin/env python3
"""Diagnostic: how long does a tiny kernel (one 256-thread block, torch add_ on 1024 floats) take on stream B while one of the step's big GEMM
kernels runs on stream A?  Separates 'waits for a place on a CU' from everything else: alone it takes ~7 us."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import desc, ops

dev = torch.device("cuda:0")
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
small = torch.zeros(1024, device=dev)
big_ew = torch.zeros(200_000_000, device=dev)

def conv(Ci, Co, thw, k):
    pad = tuple(x // 2 for x in k)
    d = desc.trim_conv(desc.conv_fwd(16, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pad, thw, groups=2))
    x = torch.randn(16, *thw, Ci, device=dev); w = torch.randn(Co, k[0] * k[1] * k[2], Ci, device=dev) * 0.05
    out = torch.empty(16, *thw, Co, device=dev)
    return lambda: ops.conv_fwd(d, x, w, out)

def wgrad(Ci, Co, thw, k):
    pad = tuple(x // 2 for x in k)
    d = desc.wgrad(16, thw, Co, Co, thw, Ci, Ci, k, (1, 1, 1), pad)
    x = torch.randn(16, *thw, Ci, device=dev); dy = torch.randn(16, *thw, Co, device=dev)
    g = torch.zeros(Co, k[0] * k[1] * k[2], Ci, device=dev)
    return lambda: ops.conv_wgrad(d, dy, x, g)

def wino(Ci, Co, thw):
    d = ops.wino_desc(16, thw[0], thw[1], thw[2], Ci, Ci, Co, Co, 3)
    x = torch.randn(16, *thw, Ci, device=dev); w = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.05
    U = ops.wino_weights(w, Co, Ci, 3); out = torch.empty(16, *thw, Co, device=dev)
    return lambda: ops.wino_conv(d, x, U, out)

CASES = [("nothing", lambda: None),
         ("elementwise add_ 0.8 GB", lambda: big_ew.add_(1.0)),
         ("conv 128->128 3x3x3 @4x112x112 (6272 blocks, 2/CU)", conv(128, 128, (4, 112, 112), (3, 3, 3))),
         ("conv 160->320 1x3x3 @28x28 (490 blocks: one round)", conv(160, 320, (1, 28, 28), (1, 3, 3))),
         ("wgrad conv112 (504 blocks: one round)", wgrad(64, 64, (4, 112, 112), (3, 3, 3))),
         ("wgrad 128->192 3x3x3 @2x28x28", wgrad(128, 192, (2, 28, 28), (3, 3, 3))),
         ("winograd conv112 (3136 whole-CU blocks)", wino(64, 64, (4, 112, 112))),
         ("winograd 96->128 @2x28x28 (256 whole-CU blocks)", wino(96, 128, (2, 28, 28)))]
delay_us = float(os.environ.get("PROBE_DELAY_US", "40"))
for name, fn in CASES:
    res = []
    for _ in range(6):
        ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            ea0.record(); fn(); ea1.record()
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < delay_us:      # let A's blocks take their places first
            pass
        with torch.cuda.stream(B):
            eb0.record(); small.add_(1.0); eb1.record()
        torch.cuda.synchronize()
        res.append((ea0.elapsed_time(ea1), eb0.elapsed_time(eb1), ea0.elapsed_time(eb0)))
    res = res[2:]
    print("%-58s A %7.3f ms | tiny kernel on B: %7.1f us (B started %6.1f us after A)" %
          (name, sum(r[0] for r in res) / len(res), 1e3 * sum(r[1] for r in res) / len(res), 1e3 * sum(r[2] for r in res) / len(res)), flush=True)
