#!/usr/bin/env python3
"""Go / no-go bench for the Winograd F(2x2, 3x3) convolution: the fused kernel (csrc/wino.hip) against the gather-GEMM kernel on the
stride-1 3x3x3 layers that dominate the step, on post-ReLU-like inputs (the clock the chip holds depends on the data).
    python tools/bench_wino.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import picons_amd  # noqa
from picons_amd import capi, desc as D, ops
from picons_amd.plan import conv_work

SHAPES = [("conv112 64->64 @4x112x112", (4, 112, 112), 64, 64), ("Conv3d_2c 64->192 @2x56x56", (2, 56, 56), 64, 192),
          ("conv56 / 2c dgrad 192->64 @2x56x56", (2, 56, 56), 192, 64), ("128->128 @4x112x112", (4, 112, 112), 128, 128),
          ("Mixed_3b b1b 96->128 @2x28x28", (2, 28, 28), 96, 128), ("Mixed_4 160->320 @1x28x28 (1x3x3)", (1, 28, 28), 160, 320)]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 16


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R


for name, thw, Ci, Co in SHAPES:
    KT = 3 if thw[0] > 1 else 1
    x = torch.randn(N, *thw, Ci, device="cuda").clamp_min(0)
    w = torch.randn(Co, Ci, KT, 3, 3, device="cuda") * (1.0 / np.sqrt(9 * KT * Ci))
    wk = w.permute(0, 2, 3, 4, 1).reshape(Co, 9 * KT, Ci).contiguous()
    dd = D.trim_conv(D.conv_fwd(N, thw, Ci, Ci, Co, Co, (KT, 3, 3), (1, 1, 1), (KT // 2, 1, 1), thw, groups=2))
    ref = torch.empty(N, *thw, Co, device="cuda")
    out = torch.empty_like(ref)
    U = ops.wino_weights(w, Co, Ci, KT)
    wd = ops.wino_desc(N, *thw, Ci, Ci, Co, Co, KT)
    t_dir = timeit(lambda: ops.conv_fwd(dd, x, wk, ref))
    t_win = timeit(lambda: ops.wino_conv(wd, x, U, out))
    t_u = timeit(lambda: ops.wino_weights(w, Co, Ci, KT, out=U))
    err = (out - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    wk_ = conv_work(dd)
    import ctypes as C
    o3 = (C.c_double * 3)()
    capi.check(capi.lib().pc_wino_work(C.byref(wd), o3))
    print("%-38s direct %7.3f ms (%5.1f TF/s executed)   winograd %7.3f ms (%5.1f TF/s issued, %5.1f TF/s direct-equivalent)   x%.2f   "
          "weights %.3f ms   rel err %.1e   blocks %d" % (name, t_dir * 1e3, 2 * wk_["executed"] / t_dir / 1e12, t_win * 1e3, 2 * o3[0] / t_win / 1e12,
                                                        2 * wk_["executed"] / t_win / 1e12, t_dir / t_win, t_u * 1e3, err, int(o3[2])), flush=True)

