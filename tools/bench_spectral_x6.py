#!/usr/bin/env python3
"""The two row-spectral PrimaryCaps GEMMs on the bf16-split kernel at the bench size (N = 16 clip-passes, 41 frequency groups):
forward 832 -> 544 over 320 rows per group, input gradient 544 -> 832 over 448 rows per group.  Prints ms per launch and a checksum;
run once per tile choice (PICONS_X6_TALL=0/1, PICONS_X6_MFAST=0/1) and compare.  `--save f` / `--check f`: store / compare the outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picons_amd  # noqa
from picons_amd import capi, ops, spectral

N, H, W, Ci, Co, K = 16, 28, 28, 832, 544, 9
dev = "cuda"
L = spectral.Layout(N, H, W, Ci, Ci, Co, Co, K, K)
g = torch.Generator().manual_seed(3)
f32 = dict(device=dev, dtype=torch.float32)
xp = torch.relu(torch.randn(L.G * L.x_g, generator=g)).to(dev)
dtp = torch.randn(L.G * L.t_g, generator=g).to(dev)
wv = (torch.randn(L.G * L.w_g, generator=g) * 0.02).to(dev)
wvt = (torch.randn(L.G * L.w_g, generator=g) * 0.02).to(dev)
pv, pvt = ops.split_planes(wv), ops.split_planes(wvt)
tp = torch.empty(L.G * L.t_g, **f32); dxp = torch.empty(L.G * L.x_g, **f32)
dc = L.conv(); dds = L.dgrad()
wsf = torch.zeros(max(int(capi.lib().pc_conv_x6_ws_floats(ops.conv_desc(dict(dc, flags=dc.get("flags", 0) | capi.F_X6)))), 4), **f32)
wsd = [torch.zeros(max(int(capi.lib().pc_conv_x6_ws_floats(ops.conv_desc(dict(dd, flags=dd.get("flags", 0) | capi.F_X6)))), 4), **f32) for dd in dds]


def fwd():
    ops.conv_fwd_x6(dc, xp, pv, tp, ws=wsf)


def dgrad():
    for dd, w in zip(dds, wsd):
        ops.conv_fwd_x6(dd, dtp, pvt, dxp, ws=w)


only = os.environ.get("ONLY", "")
for name, fn, out in (("forward  832->544, 320 rows/group", fwd, tp), ("dgrad    544->832, 448 rows/group", dgrad, dxp)):
    if only and only not in name:
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = int(os.environ.get("REPS", "20"))
    e0.record()
    for _ in range(R):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-36s %7.3f ms   checksum %.9e" % (name, e0.elapsed_time(e1) / R, out.double().sum().item()), flush=True)
if "--save" in sys.argv:
    torch.save({"tp": tp.cpu(), "dxp": dxp.cpu()}, sys.argv[sys.argv.index("--save") + 1])
if "--check" in sys.argv:
    ref = torch.load(sys.argv[sys.argv.index("--check") + 1])
    for k, t in (("tp", tp), ("dxp", dxp)):
        d = (t.cpu() - ref[k]).abs().max().item()
        print("%s: max |diff| vs saved %.3e (max |ref| %.3e)" % (k, d, ref[k].abs().max().item()))
