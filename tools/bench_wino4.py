#!/usr/bin/env python3
"""Go / no-go bench for the Winograd F(4x4, 3x3) convolution (csrc/wino4.hip) against the F(2x2, 3x3) kernel (csrc/wino.hip) on the step's
stride-1 3x3x3 layers whose H, W are multiples of 4, on post-ReLU-like inputs; errors against an fp64 convolution of a sample slice.
    python tools/bench_wino4.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import torch.nn.functional as F
import picons_amd  # noqa
from picons_amd import capi, ops

# (name, N, (T, H, W), Ci, Co): the step's launches (profiles/r05_launch_table.txt)
SHAPES = [("conv112 64->64 @4x112x112", 16, (4, 112, 112), 64, 64), ("Conv3d_2c 64->192 @2x56x56", 16, (2, 56, 56), 64, 192),
          ("2c dgrad 192->64 @4x56x56", 16, (4, 56, 56), 192, 64), ("conv56 192->64 @2x56x56", 16, (2, 56, 56), 192, 64),
          ("Mixed_3b 96->128 @2x28x28", 16, (2, 28, 28), 96, 128), ("Mixed_3c 128->192 @2x28x28", 16, (2, 28, 28), 128, 192),
          ("Mixed_3c dgrad 192->128 @2x28x28", 16, (2, 28, 28), 192, 128)]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    # the fp64 reference of the previous shape ran on the host for seconds: wake the chip up first (launches right after an idle period
    # were timed at 8 - 9 ms here, 50x their steady time)
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R


for name, N, thw, Ci, Co in SHAPES:
    KT = 3
    x = torch.randn(N, *thw, Ci, device="cuda").clamp_min(0)
    w = torch.randn(Co, Ci, KT, 3, 3, device="cuda") * (1.0 / np.sqrt(9 * KT * Ci))
    res = {}
    for m in (2, 4):
        U = ops.wino_weights(w, Co, Ci, KT, m=m)
        wd = ops.wino_desc(N, *thw, Ci, Ci, Co, Co, KT, m=m)
        out = torch.empty(N, *thw, Co, device="cuda")
        t = timeit(lambda: ops.wino_conv(wd, x, U, out))
        tu = timeit(lambda: ops.wino_weights(w, Co, Ci, KT, out=U, m=m))
        o3 = (C.c_double * 3)()
        capi.check(capi.lib().pc_wino_work(C.byref(wd), o3))
        res[m] = (t, tu, o3[0], o3[1], out)
    # fp64 reference of the first sample's first frames
    xs = x[:1].double().cpu().permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xs, w.double().cpu(), padding=(1, 1, 1)).permute(0, 2, 3, 4, 1)
    rms = ref.pow(2).mean().sqrt().item()
    e = {m: (res[m][4][:1].double().cpu() - ref) for m in (2, 4)}
    print("%-36s F(2,3) %7.3f ms (%5.1f TF/s issued)   F(4,3) %7.3f ms (%5.1f TF/s issued, %5.1f executed)   x%.2f   weights %.3f / %.3f ms   "
          "err/rms: F(2,3) max %.1e rms %.1e   F(4,3) max %.1e rms %.1e" % (
              name, res[2][0] * 1e3, 2 * res[2][2] / res[2][0] / 1e12, res[4][0] * 1e3, 2 * res[4][2] / res[4][0] / 1e12, 2 * res[4][3] / res[4][0] / 1e12,
              res[2][0] / res[4][0], res[2][1] * 1e3, res[4][1] * 1e3, e[2].abs().max().item() / rms, e[2].pow(2).mean().sqrt().item() / rms,
              e[4].abs().max().item() / rms, e[4].pow(2).mean().sqrt().item() / rms), flush=True)
