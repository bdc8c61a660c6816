#!/usr/bin/env python3
"""Where does the HIP step's gradient leave the fp64 oracle's under the reference's own initialisation?  Compares the
intermediate gradients around the capsule head (d comb, d caps_in, d masked) of a bs=2 step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import picons_amd  # noqa: F401,E402
from oracle import caps as ocaps, step as ostep  # noqa: E402
from picons_amd import step as pstep, synthetic  # noqa: E402


def main():
    cond = len(sys.argv) > 1 and sys.argv[1] == "cond"
    hw = 224
    torch.set_num_threads(16)
    akw = dict(bv=True, n_frames=5, wt_cons=0.1)
    state = synthetic.init_state(47, 24, conditioned=cond)
    eng = pstep.StepEngine(pstep.default_args(**akw), bs=2, hw=hw, state=state)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, rank=0, step=6, hw=hw)
    ramp = pstep.exp_rampup(100)(1)
    eng.stage(lab, unl, perm, drops)
    eng.forward_backward(1, ramp)
    torch.cuda.synchronize()
    p = eng.plan

    def view(t):
        return eng.aview(t.ref, t.rows * t.ld).view(t.rows, t.ld)[:, :t.C].cpu().double()
    hip = {"comb": view(p.named["comb"]), "caps_in": view(p.named["caps_in"]), "d_comb": view(p.named["d_comb"]),
           "d_caps_in": view(p.named["d_caps_in"]), "d_masked": view(p.grads["masked"][0]), "masked": view(p.named["masked"]),
           "d_cat28": view(p.grads["cat28"][0]), "d_x832d": view(p.grads["x832d"][0])}
    # oracle fp64 with taps
    P = ostep.as_torch_params(state, dtype=torch.float64)
    caught = []
    orig = ocaps.capsnet_forward

    def wrapped(*a, **k):
        taps = {}
        out = orig(*a, taps=taps, **k)
        for t in taps.values():
            if t.requires_grad:
                t.retain_grad()
        caught.append(taps)
        return out
    ocaps.capsnet_forward = wrapped
    ref = ostep.train_step(P, ostep.default_args(**akw), lab, unl, 1, ramp, perm, drops, dtype=torch.float64)
    ref["total"].backward()
    ocaps.capsnet_forward = orig
    C = 24
    for name, key, sl in (("comb", "comb", None), ("caps_in", "caps_in", None), ("d_comb", "comb", "grad"), ("d_caps_in", "caps_in", "grad")):
        o = torch.cat([(t[key].grad if sl else t[key]).reshape(-1, t[key].shape[-1]) for t in caught], 0).detach()
        h = hip[name]
        npose = o.shape[1] * 16 // 17
        for part, a, b in (("pose", 0, npose), ("act", npose, o.shape[1])):
            den = o[:, a:b].norm().item() + 1e-300
            print("%-10s %-5s rel-L2 hip vs fp64 oracle %.3e   |ref| %.3e   max|ref| %.3e" % (name, part, (h[:, a:b] - o[:, a:b]).norm().item() / den, den, o[:, a:b].abs().max().item()))
    # per-sample breakdown of d_comb pose error
    o = torch.cat([t["comb"].grad.reshape(-1, C * 17) for t in caught], 0)
    h = hip["d_comb"]
    n = o.shape[0] // 4
    for s in range(4):
        e = (h[s * n:(s + 1) * n, :C * 16] - o[s * n:(s + 1) * n, :C * 16])
        print("  sample %d: d_comb pose rel err %.3e  mask %s" % (s, e.norm().item() / (o[s * n:(s + 1) * n, :C * 16].norm().item() + 1e-300), caught[s // 2]["mask"][s % 2].tolist()[:6]))
    print("grad rel errs:", {k: float((eng.grad(k).cpu().double() - P[k].grad).norm() / P[k].grad.norm()) for k in
                             ("upsample1.weight", "upsample1.bias", "upsample2.weight", "conv28.weight", "conv_caps.weights", "primary_caps.pose.weight", "primary_caps.a.weight", "conv1.Mixed_4f.b0.conv3d.weight")})


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "em"):
    main()


def em_on_real_data():
    """Feed the oracle's own caps_in / d comb (fp64 run, cast to fp32) to the EM backward kernel alone."""
    from picons_amd import ops
    hw = 224
    torch.set_num_threads(16)
    akw = dict(bv=True, n_frames=5, wt_cons=0.1)
    state = synthetic.init_state(47, 24, conditioned=False)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, rank=0, step=6, hw=hw)
    ramp = pstep.exp_rampup(100)(1)
    P = ostep.as_torch_params(state, dtype=torch.float64)
    caught = []
    orig = ocaps.capsnet_forward

    def wrapped(*a, **k):
        taps = {}
        out = orig(*a, taps=taps, **k)
        for t in taps.values():
            if t.requires_grad:
                t.retain_grad()
        caught.append(taps)
        return out
    ocaps.capsnet_forward = wrapped
    ref = ostep.train_step(P, ostep.default_args(**akw), lab, unl, 1, ramp, perm, drops, dtype=torch.float64)
    ref["total"].backward()
    ocaps.capsnet_forward = orig
    C, B = 24, 32
    x = torch.cat([t["caps_in"].reshape(-1, B * 17) for t in caught], 0).detach()
    dcomb = torch.cat([t["comb"].grad.reshape(-1, C * 17) for t in caught], 0).detach()
    dref = torch.cat([t["caps_in"].grad.reshape(-1, B * 17) for t in caught], 0).detach()
    npos = x.shape[0]
    W = P["conv_caps.weights"].detach()
    bu, ba = P["conv_caps.beta_u"].detach(), P["conv_caps.beta_a"].detach()
    for tag, dd in (("full", dcomb), ("pose-seeds only", torch.cat([dcomb[:, :C * 16], 0 * dcomb[:, C * 16:]], 1)),
                    ("act-seeds only", torch.cat([0 * dcomb[:, :C * 16], dcomb[:, C * 16:]], 1))):
        # oracle on exactly this seed, fp64 and fp32
        res = {}
        for dt in (torch.float64, torch.float32):
            xx = x.detach().to(dt).clone().requires_grad_(True)
            v = ocaps.votes(xx[:, :B * 16].reshape(npos, B, 16), W.to(dt))
            mu, a = ocaps.em_routing(v, xx[:, B * 16:].reshape(npos, B, 1), bu.to(dt), ba.to(dt))
            torch.cat([mu.reshape(npos, C * 16), a.reshape(npos, C)], 1).backward(dd.to(dt))
            res[dt] = xx.grad.double()
        dW = torch.zeros(B, C, 4, 4, device="cuda"); dbu = torch.zeros(C, 16, device="cuda"); dba = torch.zeros(C, device="cuda")
        dx = ops.em_bwd(x.float().cuda(), W[0].float().contiguous().cuda(), bu.float().cuda(), ba.float().cuda(), dd.float().cuda(), npos, B, C, dW, dbu, dba).cpu().double()
        r64 = res[torch.float64]
        den = r64[:, :B * 16].norm().item()
        e = (dx[:, :B * 16] - r64[:, :B * 16])
        print("%-16s dpose rel-L2 vs fp64: hip %.3e  fp32 oracle %.3e  |ref| %.3e" % (tag, e.norm().item() / den, (res[torch.float32][:, :B * 16] - r64[:, :B * 16]).norm().item() / den, den))
        pe = e.norm(dim=1)
        worst = torch.argsort(-pe)[:5]
        print("   worst positions:", [(int(i), float(pe[i]), float(r64[i, :B * 16].norm())) for i in worst])
        if tag == "full":
            i = int(worst[0])
            a_in = x[i, B * 16:]
            print("   a_in of worst position: min %.3e max %.3e  #(<1e-6) %d  #(>1-1e-6) %d" % (a_in.min(), a_in.max(), int((a_in < 1e-6).sum()), int((a_in > 1 - 1e-6).sum())))
            ib = torch.argsort(-e[i].abs().reshape(B, 16).sum(1))[:4]
            print("   worst input capsules:", [(int(j), float(a_in[j]), float(e[i].reshape(B, 16)[j].abs().sum()), float(r64[i, :B * 16].reshape(B, 16)[j].abs().sum())) for j in ib])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "em":
    em_on_real_data()
