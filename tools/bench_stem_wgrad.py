#!/usr/bin/env python3
"""The stem's weight gradient at the bench size (16 clip-passes of 8x224x224x3 -> 64 x 7x7x7, stride 2): fp32 MFMA (wgrad4_kernel) against the
bf16-split kernel (wgrad4_x6_kernel), atomic and K-slice-image forms; ms per launch (operands: a clip in [0, 1], a random gradient)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import capi, desc, ops, spec

N, thw, Co = 16, (8, 224, 224), 64
k, s = (7, 7, 7), (2, 2, 2)
pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
othw = tuple((thw[i] + pads[i][0] + pads[i][1] - k[i]) // s[i] + 1 for i in range(3))
g = torch.Generator().manual_seed(5)
x = torch.rand(N, *thw, 4, generator=g); x[..., 3] = 0
x = x.cuda()
dy = torch.randn(N, *othw, Co, generator=g).cuda()
wd = dict(desc.wgrad(N, othw, Co, Co, thw, 4, 4, k, s, [p[0] for p in pads]), flags=capi.WG_CS3)
res = {}
for name, fl in (("fp32 MFMA (wgrad4_kernel)", capi.WG_CS3), ("bf16 split (wgrad4_x6_kernel)", capi.WG_CS3 | capi.WG_X6)):
    d = dict(wd, flags=fl)
    ns = ops.wgrad_slices(d)
    ws = torch.zeros(ns * Co * 343 * 4, device="cuda")
    gw = torch.zeros(Co, 343, 4, device="cuda")
    for mode, fn in (("atomic", lambda: ops.conv_wgrad(d, dy, x, gw)), ("slices", lambda: ops.conv_wgrad(dict(d, ws_slices=ns), dy, x, ws))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 20
        gw.zero_()
        e0.record()
        for _ in range(R):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("%-32s %-7s %3d slices  %7.3f ms" % (name, mode, ns, e0.elapsed_time(e1) / R), flush=True)
    res[name] = ws.view(ns, -1).sum(0)
a, b = list(res.values())
print("rel-L2 between the two kernels' gradients: %.3e" % ((a - b).norm() / a.norm()).item())
