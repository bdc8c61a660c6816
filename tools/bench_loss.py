import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, picons_amd
from picons_amd import step as pstep, synthetic, ops, capi
import numpy as np
for gv in (False, True):
    args = pstep.default_args(bv=not gv, gv=gv, n_frames=5, wt_cons=0.1)
    eng = pstep.StepEngine(args, bs=8, hw=224, lanes=1)
    lab, unl, perm, drops = synthetic.make_step_inputs(8)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(1, 0.01)
    torch.cuda.synchronize()
    arr = eng.ops["loss"]
    for _ in range(3): ops.run_ops(arr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): ops.run_ops(arr)
    torch.cuda.synchronize()
    print("NC=%s gv=%s loss list: %.1f us" % (os.environ.get("PICONS_LOSS_NC", "1"), gv, (time.perf_counter() - t0) / 50 * 1e6), eng.read_scalars())
