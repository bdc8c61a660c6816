#!/usr/bin/env python3
"""What the hi / lo accumulator pair buys in the generic bf16-split weight gradient (csrc/conv.hip wgrad_body<..., X6, HILO>):
error against an fp64 torch gradient and time per launch for native fp32 MFMA, the split with the pair, and the split with one accumulator
(flag bit 4, diagnostic library only).  Runs on the GPU box:  make -C pi-consistency-activity-detection_amd/csrc diag && python tools/wgrad_x6_acc_probe.py"""
import os, sys
os.environ.setdefault("PICONS_DIAG_LIB", "1")      # the one-accumulator variant lives in libpicons_diag.so only
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from picons_amd import capi, desc, ops, spec
DEV = "cuda:0"


def cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n


for Ci, Co, k, thw, N in [(256, 288, (1, 1, 1), (1, 28, 28), 16), (528, 128, (1, 1, 1), (4, 14, 14), 16), (832, 384, (1, 1, 1), (2, 7, 7), 32), (192, 64, (1, 1, 1), (4, 28, 28), 8), (64, 96, (1, 3, 3), (2, 14, 30), 4)]:
    g = torch.Generator().manual_seed(31)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g)))
    w = (torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))).double().requires_grad_(True)
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    taps = int(np.prod(k))
    ref = w.grad.reshape(Co, Ci, taps).permute(0, 2, 1)
    pf = [p[0] for p in pads]
    xg, dyg = cl(x), cl(dy)
    wd = desc.wgrad(N, thw, Co, Co, thw, Ci, Ci, k, (1, 1, 1), pf)
    out = {}
    for tag, fl in (("native", 0), ("x6 hi/lo", capi.WG_X6), ("x6 one acc", capi.WG_X6 | 4)):
        d = dict(wd, flags=fl)
        got = ops.conv_wgrad(d, dyg, xg, torch.zeros(Co, taps, Ci, device=DEV))
        buf = torch.zeros(Co, taps, Ci, device=DEV)
        out[tag] = (rel(got.cpu(), ref), timeit(lambda: ops.conv_wgrad(d, dyg, xg, buf)))
    print("Ci %4d Co %4d k %s thw %s N %2d: " % (Ci, Co, k, thw, N) + "   ".join("%s err %.3e %.1f us" % (t, e, ms * 1e3) for t, (e, ms) in out.items()))
