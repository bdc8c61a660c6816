"""Diagnostic: host time per phase of bench.py's `staged` leg (every step's minibatch prepared from host uint8 frames).  The host must stay
non-blocking outside read_scalars -- any pageable host-to-device copy blocks until the stream has drained (found: 21 ms per step)."""
import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
import torch, numpy as np
import bench
from picons_amd import step as pstep
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
eng = pstep.StepEngine(args, bs=8, hw=224)
si = bench.StagedInputs(eng, 8, 24, 0)
T = dict(commit=0, fb=0, adam=0, prep=0, read=0)
st = si.st
si.prep(0, 0)
for i in range(60):
    if i == 40:
        torch.cuda.synchronize(); T = {k: 0 for k in T}; t00 = time.perf_counter()
    slot = i & 1
    t = time.perf_counter(); st.commit(slot); T['commit'] += time.perf_counter() - t
    t = time.perf_counter(); eng.arm_early_adam(1e-4, True); eng.forward_backward(1, 0.01, None); T['fb'] += time.perf_counter() - t
    t = time.perf_counter(); eng.adam(1e-4, 1.0); st.release(slot); T['adam'] += time.perf_counter() - t
    t = time.perf_counter(); si.prep(i + 1, slot ^ 1); T['prep'] += time.perf_counter() - t
    t = time.perf_counter(); out = eng.read_scalars(); T['read'] += time.perf_counter() - t
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print('per step ms: total %.2f' % (tot / 20 * 1e3), {k: round(v / 20 * 1e3, 2) for k, v in T.items()})
