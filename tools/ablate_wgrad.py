#!/usr/bin/env python3
"""Diagnostic: wgrad_kernel on representative shapes; PICONS_WGRAD_ABLATE=1 removes the tile fetch from the K loop."""
import os
os.environ.setdefault("PICONS_DIAG_LIB", "1")      # the ablation / stamp variants live in libpicons_diag.so only (make -C .../csrc diag)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import desc, ops

SHAPES = [  # name, N, thw, Ci(S), Co(D), k, pad
    ("conv112 wgrad 64x64 3x3x3 @4x112x112", 16, (4, 112, 112), 64, 64, (3, 3, 3), (1, 1, 1)),
    ("3x3x3 128->128 @4x112x112", 16, (4, 112, 112), 128, 128, (3, 3, 3), (1, 1, 1)),
    ("primary caps 832->544 9x9", 16, (1, 28, 28), 832, 544, (1, 9, 9), (0, 0, 0)),
    ("3x3 160->320 @28x28", 16, (1, 28, 28), 160, 320, (1, 3, 3), (0, 1, 1)),
    ("1x1 832->256 @28x28", 16, (1, 28, 28), 832, 256, (1, 1, 1), (0, 0, 0)),
]
mode = os.environ.get("PICONS_WGRAD_ABLATE", "0")
for name, N, thw, Ci, Co, k, pad in SHAPES:
    othw = tuple(thw[i] + 2 * pad[i] - k[i] + 1 for i in range(3))
    d = desc.wgrad(N, othw, Co, Co, thw, Ci, Ci, k, (1, 1, 1), pad)
    x = torch.randn(N, *thw, Ci, device="cuda")
    dy = torch.randn(N, *othw, Co, device="cuda")
    g = torch.zeros(Co, k[0] * k[1] * k[2], Ci, device="cuda")
    for _ in range(2):
        ops.conv_wgrad(d, dy, x, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for _ in range(R):
        ops.conv_wgrad(d, dy, x, g)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    fl = 2.0 * N * othw[0] * othw[1] * othw[2] * Co * Ci * k[0] * k[1] * k[2]
    print("WABL=%s %-42s %8.3f ms  %6.1f TF/s" % (mode, name, dt * 1e3, fl / dt / 1e12), flush=True)
