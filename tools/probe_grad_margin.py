#!/usr/bin/env python3
"""How close the element-wise gradient bars of tests/test_step_gpu.py::test_step_vs_reference_golden_full_size sit to the engine's own
noise: for every bs = 2 golden, the worst |g - g_ref32|.max() / |g_ref32|.max() over the parameter tensors with the Winograd layers in
F(2x2, 3x3) only (PICONS_WINO4=0) and with the 112 x 112 / 56 x 56 layers in F(4x4, 3x3) (default).
    python tools/probe_grad_margin.py"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import picons_amd  # noqa
from picons_amd import step as pstep, synthetic

GOLDEN = os.path.join(ROOT, "tests", "golden")
for tag in ("step_bv5", "step_gv_pseudo", "step_bvgv3", "step_jhmdb_bv"):
    S = np.load(os.path.join(GOLDEN, tag + ".npz"))
    for w4 in ("0", "1"):
        os.environ["PICONS_WINO4"] = w4
        akw = dict(ast.literal_eval(str(S["args"])))
        jh = akw.pop("dataset", "ucf101") == "jhmdb"
        akw.pop("wt_seg", None)
        bs = int(S["bs"]) if "bs" in S.files else 2
        eng = pstep.StepEngine(pstep.default_args(**akw), bs=bs, hw=224, num_classes=int(S["num_classes"]), jhmdb=jh)
        lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=0, step=int(S["stepid"]), num_classes=int(S["num_classes"]))
        eng.stage(lab, unl, perm, drops)
        eng.forward_backward(int(S["epoch"]), float(S["ramp"]))
        got = eng.read_scalars()
        rows, rows64 = [], []
        for k in S.files:
            if k.startswith("grad::"):
                ref, g = S[k], eng.grad(k[6:]).cpu().numpy()
                rows.append((float(np.abs(g - ref).max() / (np.abs(ref).max() + 1e-30)), k[6:]))
                if "f64::" + k in S.files:       # the reference's fp64 run: engine's and fp32 reference's distance from it, max-relative
                    g64 = S["f64::" + k]
                    rows64.append((float(np.abs(g - g64).max() / np.abs(g64).max()), float(np.abs(ref - g64).max() / np.abs(g64).max()), k[6:]))
        rows.sort(reverse=True)
        rows64.sort(key=lambda q: -q[0] / max(2 * q[1], 1e-2))
        print("    vs fp64 (engine, fp32 reference): " + "  ".join("%s %.4f %.4f" % (n.replace("conv1.", "").replace(".conv3d", ""), a, b) for a, b, n in rows64[:5]))
        r = np.array([q[0] for q in rows])
        print("%-16s WINO4=%s  loss err %.1e  worst %s  median %.4f  over 1.5 %%: %d of %d" % (
            tag, w4, max(abs(got[k] - float(S[k])) for k in ("total", "loc", "cls", "cons")),
            "  ".join("%s %.4f" % (n.replace("conv1.", "").replace(".conv3d", ""), v) for v, n in rows[:4]), float(np.median(r)), int((r > 0.015).sum()), len(r)), flush=True)
        del eng
