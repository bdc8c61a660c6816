#!/usr/bin/env python3
"""Generic / row-segment bf16-split weight gradients at a few shapes of the step: ms per launch (K-slice-image form)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import capi, desc, ops, spec

CASES = [  # (N, thw, Ci, Co, k)
    (16, (1, 28, 28), 512, 288, (1, 1, 1)),
    (16, (1, 28, 28), 528, 448, (1, 1, 1)),
    (16, (2, 28, 28), 256, 288, (1, 1, 1)),
    (16, (2, 56, 56), 64, 128, (3, 3, 3)),      # generic 128 x 128 (Cd = 128, Cs = 64 -> row3 needs Cs % 64: goes where the plan sends it)
    (16, (4, 112, 112), 64, 64, (3, 3, 3)),     # conv112: wgrad3_x6
    (16, (2, 56, 56), 192, 64, (3, 3, 3)),
]
g = torch.Generator().manual_seed(1)
for N, thw, Ci, Co, k in CASES:
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    x = torch.relu(torch.randn(N, *thw, Ci, generator=g)).cuda()
    dy = torch.randn(N, *thw, Co, generator=g).cuda()
    wd = desc.trim_wgrad(dict(desc.wgrad(N, thw, Co, Co, thw, Ci, Ci, k, (1, 1, 1), [p[0] for p in pads]), flags=capi.WG_X6))
    ns = ops.wgrad_slices(wd)
    taps = k[0] * k[1] * k[2]
    ws = torch.zeros(ns * Co * taps * Ci, device="cuda")
    d = dict(wd, ws_slices=ns)
    for _ in range(3):
        ops.conv_wgrad(d, dy, x, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R):
        ops.conv_wgrad(d, dy, x, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    fl = 2.0 * N * thw[0] * thw[1] * thw[2] * Ci * Co * wd["ntap"][0] * wd["ntap"][1] * wd["ntap"][2]
    print("N %2d thw %-14s Ci %4d Co %4d k %s  slices %3d  %7.3f ms  %6.1f TF/s  checksum %.6e" % (N, thw, Ci, Co, k, ns, ms, fl / ms / 1e9, ws.double().sum().item()), flush=True)
