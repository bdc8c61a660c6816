"""Diagnostic: host time the replay of every op list takes (no profiler attached), per step of the resident-input loop, and how long the host
waits in read_scalars().  A list whose enqueue costs more than the GPU time in front of its first kernel leaves the GPU waiting for the host."""
import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
import torch
from picons_amd import ops, step as pstep, synthetic
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
eng = pstep.StepEngine(args, bs=8, hw=224)
eng.stage(*synthetic.make_step_inputs(8, step=0))
names = {id(v): k for k, v in eng.ops.items() if hasattr(v, "dtype")}
T, N = {}, {}
orig = ops.run_ops


def timed(arr, side=None):
    t = time.perf_counter()
    r = orig(arr, side=side)
    k = names.get(id(arr), "?")
    T[k] = T.get(k, 0.0) + time.perf_counter() - t
    N[k] = len(arr)
    return r


ops.run_ops = timed
pstep.ops.run_ops = timed
R = 0.0
for i in range(60):
    if i == 30:
        torch.cuda.synchronize(); T.clear(); R = 0.0; t00 = time.perf_counter()
    eng.arm_early_adam(1e-4, True)
    eng.forward_backward(1, 0.01, None)
    eng.adam(1e-4, 1.0)
    t = time.perf_counter(); eng.read_scalars(); R += time.perf_counter() - t
torch.cuda.synchronize()
tot = (time.perf_counter() - t00) / 30 * 1e3
print("step %.3f ms; host ms per list (ops):" % tot, {k: (round(v / 30 * 1e3, 3), N[k]) for k, v in T.items()}, "read_scalars wait %.3f ms" % (R / 30 * 1e3))
print("host busy outside read_scalars: %.3f ms of %.3f" % (tot - R / 30 * 1e3, tot))
