#!/usr/bin/env python3
"""Diagnostic: does the place of a process' FIRST collective change the step time?  (It did: ROCm binds a stream to one of its
four hardware queues at first use; a one-rank RCCL all-reduce or barrier issued before the engine's lanes had been used took a
queue, two lanes then shared one, and every later step ran 0.8 - 4 ms slower.  StepEngine now uses its lanes, in order, at
construction -- PICONS_BIND_LANES=0 shows the old behaviour.)

    PROBE_ONES=<mode> python tools/probe_dp_queues.py        # one GPU; a one-rank "nccl" group, GradReducer forced active
    modes: "" (first collective = the first bucket inside a step), 1 (all-reduce of ones on the default stream before any step),
           comm (the same on the reducer's comm stream), barrier, late / latebarrier (after eight steps)
    also try GPU_MAX_HW_QUEUES=2..8 with bench.py: 4 (the default) is the only good value (28.2 / 25.2 / 24.6 / 30.2 / 29.8 ms for 2..6)
"""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
import torch, torch.distributed as dist
import picons_amd
from picons_amd import step as pstep, synthetic, ops
dist.init_process_group("nccl", rank=0, world_size=1)
args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1, lr=1e-4)
eng = pstep.StepEngine(args, bs=8, hw=224)
eng.stage(*synthetic.make_step_inputs(8, rank=0, step=0))
red = eng.make_reducer(force=True)
mode = os.environ.get("PROBE_ONES", "")
def do_ones():
    ones = torch.ones(1, device="cuda:0"); dist.all_reduce(ones); print("ones", ones.item())
if mode == "1":
    do_ones()
if mode == "comm":
    with torch.cuda.stream(red.comm_stream):
        do_ones()
if mode == "barrier":
    dist.barrier()
ramp = pstep.exp_rampup(100)(1)
orig = red.launch
times = []
def timed(i, streams=()):
    t0 = time.perf_counter(); orig(i, streams); times.append((i, (time.perf_counter() - t0) * 1e3))
red.launch = timed
for it in range(8):
    t0 = time.perf_counter()
    eng.run_staged(1, ramp, reducer=red)
    host = (time.perf_counter() - t0) * 1e3
    if it >= 5:
        print("step host time %.2f ms; launch() host ms:" % host, ["%d:%.3f" % t for t in times[-5:]])
torch.cuda.synchronize()
if mode == "late":
    do_ones()
if mode == "latebarrier":
    dist.barrier()
t0 = time.perf_counter()
for it in range(50): eng.run_staged(1, ramp, reducer=red)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 50 * 1e3)
dist.destroy_process_group()
