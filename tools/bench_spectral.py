#!/usr/bin/env python3
"""Stage timings of the row-spectral PrimaryCaps at the bench size (N = 16 clip-passes, 28x28x832 -> 20x20x544)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import capi, ops, spectral

N, H, W, Ci, Co, K = 16, 28, 28, 832, 544, 9
dev = "cuda"
L = spectral.Layout(N, H, W, Ci, Ci, Co, Co, K, K)
m = {k: torch.from_numpy(v).to(dev) for k, v in spectral.matrices(W, K).items()}
f32 = dict(device=dev, dtype=torch.float32)
x = torch.randn(N, H, W, Ci, **f32); wf = torch.randn(Co, K * K, Ci, **f32) * 0.01; wt = torch.randn(Ci, K * K, Co, **f32) * 0.01
bias = torch.zeros(Co, **f32); dy = torch.randn(N, L.OH, L.OW, Co, **f32)
xp = torch.empty(L.G * L.x_g, **f32); wv = torch.empty(L.G * L.w_g, **f32); tp = torch.empty(L.G * L.t_g, **f32)
y = torch.empty(N, L.OH, L.OW, Co, **f32); dtp = torch.empty_like(tp); dv = torch.empty_like(wv); kg = torch.empty(Co, K * K, Ci, **f32)
wvt = torch.empty_like(wv); dxp = torch.empty_like(xp); dx = torch.empty_like(x)


def dgrad_all():
    for dd in L.dgrad():
        ops.conv_fwd(dd, dtp, wvt, dxp)


stages = [
    ("x -> operand planes (row DFT)", lambda: ops.axis_linear(L.x_to_planes(), x, m["F"], xp)),
    ("weight planes (fwd layout)", lambda: ops.wspec_fwd(wf, m["tw"], Co, Ci, K, K, L.nu, L.Ur, wv)),
    ("grouped conv 9x1, %d groups" % L.G, lambda: ops.conv_fwd(L.conv(), xp, wv, tp)),
    ("result planes -> y (+bias, sigmoid)", lambda: ops.axis_linear(L.planes_to_y(capi.ACT_SIGMOID, 512), tp, m["G"], y, bias=bias)),
    ("dy -> result-plane grads", lambda: ops.axis_linear(L.dy_to_planes(Co), dy, m["Gt"], dtp)),
    ("wgrad, %d groups in one launch" % L.G, lambda: ops.conv_wgrad(L.wgrad(), dtp, xp, dv)),
    ("weight-plane adjoint", lambda: ops.wspec_bwd(dv, m["tw"], Co, Ci, K, K, L.nu, L.Ur, kg)),
    ("weight planes (dgrad layout)", lambda: ops.wspec_fwd(wt, m["tw"], Ci, Co, K, K, L.nu, L.Ur, wvt)),
    ("grouped dgrad", dgrad_all),
    ("operand-plane grads -> dx", lambda: ops.axis_linear(L.planes_to_dx(Ci, False), dxp, m["Ft"], dx)),
]
tot = 0.0
for name, fn in stages:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 5
    for _ in range(R):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    tot += ms
    print("%-36s %7.3f ms" % (name, ms), flush=True)
print("%-34s %7.3f ms   (direct form: fwd 4.66 + dgrad 4.36+0.40 + wgrad 4.98 = 14.4 ms)" % ("total", tot))
print("GEMM FLOPs each: %.1f G (direct 469 G)" % (L.flops() / 1e9))
