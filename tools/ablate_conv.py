#!/usr/bin/env python3
"""Diagnostic: price the phases of conv_gemm_kernel on representative layer shapes.
Run one process per mode:  PICONS_CONV_ABLATE=<0|1|2|3> python tools/ablate_conv.py
(0 = product kernel; 1-3 are wrong-result ablations, see csrc/conv.hip launch_conv)."""
import os
os.environ.setdefault("PICONS_DIAG_LIB", "1")      # the ablation / stamp variants live in libpicons_diag.so only (make -C .../csrc diag)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import desc, ops
from picons_amd.plan import conv_work

SHAPES = [  # name, N, thw, Ci, Co, k
    ("conv112 3x3x3 64->64 @4x112x112", 16, (4, 112, 112), 64, 64, (3, 3, 3)),
    ("1x1x1 128->128 @4x112x112", 16, (4, 112, 112), 128, 128, (1, 1, 1)),
    ("3x3x3 128->128 @4x112x112", 16, (4, 112, 112), 128, 128, (3, 3, 3)),
    ("primary caps 9x9 832->544 @28x28", 16, (1, 28, 28), 832, 544, (1, 9, 9)),
    ("3x3 160->320 @28x28", 16, (1, 28, 28), 160, 320, (1, 3, 3)),
    ("1x1 832->256 @28x28", 16, (1, 28, 28), 832, 256, (1, 1, 1)),
    ("3x3x3 64->192 @2x56x56", 16, (2, 56, 56), 64, 192, (3, 3, 3)),
    ("3x3x3 192->64 @2x56x56", 16, (2, 56, 56), 192, 64, (3, 3, 3)),
    ("3x3x3 128->192 @1x28x28 (T=2 in)", 16, (2, 28, 28), 128, 192, (3, 3, 3)),
    ("1x1 480->304 @28x28 (stacked)", 16, (1, 28, 28), 480, 304, (1, 1, 1)),
]
only = os.environ.get("ABLATE_ONLY")
mode = os.environ.get("PICONS_CONV_ABLATE", "0") + "/var" + os.environ.get("PICONS_CONV_VARIANT", "-")
R = int(os.environ.get("ABLATE_REPS", "5"))
for name, N, thw, Ci, Co, k in SHAPES:
    if only and only not in name:
        continue
    valid = name.startswith("primary")
    pad = (0, 0, 0) if valid else tuple(x // 2 for x in k)
    othw = tuple(thw[i] + 2 * pad[i] - k[i] + 1 for i in range(3))
    d = desc.trim_conv(desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pad, othw, groups=2))
    w_ = conv_work(d)          # FLOPs as the kernel runs them (taps that are padding for a whole tile are skipped)
    x = torch.randn(N, *thw, Ci, device="cuda")
    w = torch.randn(Co, k[0] * k[1] * k[2], Ci, device="cuda") * 0.05
    out = torch.empty(N, *othw, Co, device="cuda")
    for _ in range(2):
        ops.conv_fwd(d, x, w, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        ops.conv_fwd(d, x, w, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print("ABL=%s %-40s %8.3f ms  executed %6.1f TF/s (%.3f of peak)  mfma-issued %6.1f  blocks %5d tile %dx%d" %
          (mode, name, dt * 1e3, 2 * w_["executed"] / dt / 1e12, 2 * w_["executed"] / dt / 157.3e12, 2 * w_["issued"] / dt / 1e12, w_["blocks"], w_["bm"], w_["bn"]), flush=True)
