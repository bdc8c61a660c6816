#!/usr/bin/env python3
"""Diagnostic: does a tiny kernel on one HIP stream wait for a big-grid kernel that another stream is still dispatching?
For every pair of streams (A, B): A runs a streaming kernel with a very large grid (~0.5 ms), B a 4-block kernel right behind its launch;
prints B's start-to-end and launch-to-end times.  If B's time tracks A's, the two streams' hardware queues are served by one dispatcher."""
import sys, time
import torch

dev = torch.device("cuda:0")
big = torch.zeros(300_000_000, device=dev)          # 1.2 GB: ~0.45 ms per pass
small = torch.zeros(1024, device=dev)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(4)]
for s in streams:                                   # bind every stream to its hardware queue
    with torch.cuda.stream(s):
        small.add_(1.0)
torch.cuda.synchronize()

def trial(a, b, heavy):
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[a]):
        ea0.record()
        if heavy:
            big.add_(1.0)
        ea1.record()
    with torch.cuda.stream(streams[b]):
        eb0.record()
        small.add_(1.0)
        eb1.record()
    torch.cuda.synchronize()
    return ea0.elapsed_time(ea1), eb0.elapsed_time(eb1), ea0.elapsed_time(eb1)

for heavy in (False, True):
    print("big kernel on A: %s" % heavy)
    for a in range(len(streams)):
        for b in range(len(streams)):
            if a == b:
                continue
            r = [trial(a, b, heavy) for _ in range(5)][2:]
            ta = sum(x[0] for x in r) / len(r); tb = sum(x[1] for x in r) / len(r); tt = sum(x[2] for x in r) / len(r)
            print("  A=stream %d  B=stream %d   A %.3f ms   tiny kernel on B: own span %.3f ms, A-start -> B-end %.3f ms" % (a, b, ta, tb, tt))
