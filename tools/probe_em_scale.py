#!/usr/bin/env python3
"""EM routing forward / backward kernels against the fp64 oracle over pose / weight / activation scales (the reference's own
initialisation gives poses of std ~13, saturated activations and randn weights; the parity-test init ~1 / 0.5)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import picons_amd  # noqa: F401,E402
from oracle import caps as ocaps  # noqa: E402
from picons_amd import ops  # noqa: E402

DEV = "cuda"


def em_oracle(x, W, bu, ba, dout, npos, B, C_, dt):
    x = x.to(dt).requires_grad_(True); W = W.to(dt).requires_grad_(True)
    bu = bu.to(dt).requires_grad_(True); ba = ba.to(dt).requires_grad_(True)
    v = ocaps.votes(x[:, :B * 16].reshape(npos, B, 16), W)
    mu, a = ocaps.em_routing(v, x[:, B * 16:].reshape(npos, B, 1), bu, ba)
    out = torch.cat([mu.reshape(npos, C_ * 16), a.reshape(npos, C_)], 1)
    out.backward(dout.to(dt))
    return [out[:, :C_ * 16].detach(), out[:, C_ * 16:].detach(), x.grad[:, :B * 16], x.grad[:, B * 16:], W.grad[0], bu.grad, ba.grad]


def main():
    B, C_, npos = 32, 24, 200
    for pscale, wscale, ascale, dscale in [(0.3, 0.5, 0, 1.0), (3, 1, 0, 1.0), (13, 1, 0, 1.0), (13, 1, 13, 1.0), (13, 1, 13, 0.0), (1, 0.5, 13, 1.0)]:
        g = torch.Generator().manual_seed(10)
        act = torch.rand(npos, B, generator=g) if ascale == 0 else torch.sigmoid(torch.randn(npos, B, generator=g) * ascale)
        x = torch.cat([torch.randn(npos, B * 16, generator=g) * pscale, act], 1)
        W = torch.randn(1, B, C_, 4, 4, generator=g) * wscale
        bu = torch.randn(C_, 16, generator=g); ba = torch.randn(C_, generator=g)
        dout = torch.randn(npos, C_ * 17, generator=g)
        dout[:, C_ * 16:] *= dscale
        r64 = em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float64)
        r32 = em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float32)
        xg = x.to(DEV); Wg = W[0].contiguous().to(DEV)
        og = ops.em_fwd(xg, Wg, bu.to(DEV), ba.to(DEV), npos, B, C_)
        dW = torch.zeros(B, C_, 4, 4, device=DEV); dbu = torch.zeros(C_, 16, device=DEV); dba = torch.zeros(C_, device=DEV)
        dx = ops.em_bwd(xg, Wg, bu.to(DEV), ba.to(DEV), dout.to(DEV), npos, B, C_, dW, dbu, dba)
        got = [og[:, :C_ * 16], og[:, C_ * 16:], dx[:, :B * 16], dx[:, B * 16:], dW, dbu, dba]
        print("pose %.1f  W %.1f  act %s  d(a_out) x%.0f" % (pscale, wscale, "uniform" if ascale == 0 else "sigmoid(N(0,%d))" % ascale, dscale))
        for n, gt, a32, a64 in zip(["mu", "a_out", "dpose", "da_in", "dW", "dbeta_u", "dbeta_a"], got, r32, r64):
            den = a64.norm().item() + 1e-300
            print("   %-8s rel-L2 vs fp64: hip %.3e   fp32 oracle %.3e   |ref| %.3e" % (n, (gt.cpu().double() - a64).norm().item() / den, (a32.double() - a64).norm().item() / den, den))


if __name__ == "__main__":
    main()
