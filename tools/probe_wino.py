#!/usr/bin/env python3
"""Diagnostic for csrc/wino.hip: one process per PICONS_WINO_VARIANT (ablations give wrong results by design), conv112 and conv56 shapes.
    python tools/probe_wino.py            # runs every variant in a child process
Variant 32 prints in-kernel s_memtime spans (prologue / K loop / epilogue, cycles per K chunk)."""
import os
os.environ.setdefault("PICONS_DIAG_LIB", "1")      # the ablation / stamp variants live in libpicons_diag.so only (make -C .../csrc diag)
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("conv112 64->64 @4x112x112", (4, 112, 112), 64, 64), ("192->64 @2x56x56", (2, 56, 56), 192, 64), ("64->192 @2x56x56", (2, 56, 56), 64, 192)]
M = int(os.environ.get("PROBE_M", "2"))             # 4: csrc/wino4.hip (F(4x4, 3x3)) on the shapes it takes
if M == 4:
    SHAPES = [("64->64 @4x224x224 (N=4)", (4, 224, 224), 64, 64), ("192->64 @2x56x56", (2, 56, 56), 192, 64), ("64->192 @2x56x56", (2, 56, 56), 64, 192),
              ("96->128 @2x28x28", (2, 28, 28), 96, 128), ("128->192 @2x28x28", (2, 28, 28), 128, 192)]
NAMES = {0: "product", 1: "no patch loads", 2: "no U DMA", 3: "no loads, no DMA", 4: "no transform stores", 7: "MFMA + LDS reads only", 8: "no output stores",
         15: "MFMA loop + inverse only", 32: "stamps", 64: "product + epilogue stamps"}


def child(var):
    import numpy as np
    import torch
    import picons_amd  # noqa
    from picons_amd import ops
    R = 10
    for name, thw, Ci, Co in SHAPES:
        N = 4 if thw[1] == 224 else 16
        x = torch.randn(N, *thw, Ci, device="cuda").clamp_min(0)
        w = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * (1.0 / np.sqrt(27 * Ci))
        out = torch.empty(N, *thw, Co, device="cuda")
        U = ops.wino_weights(w, Co, Ci, 3, m=M)
        wd = ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, m=M)
        dbg = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
        fn = lambda: ops.wino_conv(wd, x, U, out, bnpart=dbg)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(R):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / R
        line = "var %2d %-26s %-28s %7.3f ms" % (var, NAMES.get(var, "?"), name, dt * 1e3)
        if var == 64:
            d = dbg.cpu().numpy().reshape(-1, 8)
            d = d[d[:, 5] > 0]
            med = lambda a: float(np.median(a))
            line += "   blocks %d  prologue %.0f  loop %.0f (%.0f / chunk)  epilogue: transform + send %.0f  barrier + finalise %.0f  barrier + stores %.0f cycles" % (
                len(d), med(d[:, 0]), med(d[:, 1]), med(d[:, 1] / d[:, 5]), med(d[:, 2]), med(d[:, 3]), med(d[:, 4]))
        if var == 32:
            d = dbg.cpu().numpy().reshape(-1, 4)
            d = d[d[:, 3] > 0]
            med = lambda a: float(np.median(a))
            line += "   blocks %d  prologue %.0f  loop %.0f (%.0f / chunk)  epilogue %.0f cycles (medians; 100 MHz memtime units x ?)" % (
                len(d), med(d[:, 0]), med(d[:, 1]), med(d[:, 1] / d[:, 3]), med(d[:, 2]))
        print(line, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for v in [int(q) for q in os.environ.get("PROBE_VARS", "0,1,2,3,4,7,8,15,32").split(",")]:
            subprocess.run([sys.executable, os.path.abspath(__file__), str(v)], env=dict(os.environ, PICONS_WINO_VARIANT=str(v)))
