"""Row-spectral form of PrimaryCaps (capsules_ucf101.py:43-49: Conv2d(832, 32*16 + 32, kernel 9, stride 1)).

y[n,oy,ox,co] = sum_{ci,ky,kx} w[co,ci,ky,kx] x[n,oy+ky,ox+kx,ci] is a correlation along x, so with a length-P
(P = input width) DFT along the rows   Y^[u] = X^[u] * conj(W^[u])   per frequency u, and what is left is a 9-tap
conv along y with complex channels.  No wrap-around reaches the valid outputs (ox + kx <= P-1).  In real form
(channels [re | im]) it is ONE grouped conv, group = frequency u = 0..P/2, 2*Ci -> 2*Co channels, 9x1 taps:
15*9*2*2 = 540 real multiply-adds per (ci, co, output row) instead of 81*20 = 1620 -- a third of the direct form's
FLOPs, equal to it in exact arithmetic.  The GEMMs are the ordinary conv / wgrad kernels; the DFTs, the weight
spectrum and its adjoint are the three small kernels of csrc/spectral.hip.

This module builds the constant matrices and lays out the descriptors; `primary_caps_fwd_bwd` runs the whole
thing on torch tensors for the kernel-level parity test (tests/test_kernels_gpu.py)."""
import numpy as np

from . import capi, desc as D


def n_freq(P):
    return P // 2 + 1


def dft_matrix(P):
    """F [2*nu][P]: rows (u, re) = cos(2 pi u x / P), (u, im) = -sin(2 pi u x / P)."""
    u = np.arange(n_freq(P), dtype=np.float64)[:, None]
    x = np.arange(P, dtype=np.float64)[None, :]
    th = 2.0 * np.pi * u * x / P
    F = np.empty((2 * n_freq(P), P), dtype=np.float64)
    F[0::2] = np.cos(th)
    F[1::2] = -np.sin(th)
    return F


def twiddles(P, KX):
    """tw [nu][KX][2] = (cos, -sin)(2 pi u kx / P): the row DFT restricted to the kernel's taps."""
    F = dft_matrix(P)[:, :KX]
    return np.stack([F[0::2], F[1::2]], axis=-1).astype(np.float32)


def idft_matrix(P, OW):
    """G [OW][2*nu]: y[ox] = sum_u G[ox][(u,re)] Yr_u + G[ox][(u,im)] Yi_u  (real inverse DFT from the half spectrum)."""
    nu = n_freq(P)
    u = np.arange(nu, dtype=np.float64)[None, :]
    ox = np.arange(OW, dtype=np.float64)[:, None]
    th = 2.0 * np.pi * u * ox / P
    c = np.full(nu, 2.0)
    c[0] = 1.0
    if P % 2 == 0:
        c[-1] = 1.0
    G = np.empty((OW, 2 * nu), dtype=np.float64)
    G[:, 0::2] = c * np.cos(th) / P
    G[:, 1::2] = -c * np.sin(th) / P
    return G


def matrices(P, KX):
    """-> dict of float32 arrays: F (x -> X^), Ft (dX^ -> dx), G (Y^ -> y), Gt (dy -> dY^), tw."""
    OW = P - KX + 1
    F, G = dft_matrix(P), idft_matrix(P, OW)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(F=f32(F), Ft=f32(F.T), G=f32(G), Gt=f32(G.T), tw=twiddles(P, KX))


def axis(R, I, O, C, in_sr, in_hi, in_lo, in_split, out_sr, out_hi, out_lo, out_split, act=0, act_c0=0, accum=0):
    return dict(R=R, I=I, O=O, C=C, in_split=in_split, out_split=out_split, act=act, act_c0=act_c0, accum=accum,
                in_sr=in_sr, in_hi=in_hi, in_lo=in_lo, out_sr=out_sr, out_hi=out_hi, out_lo=out_lo)


class Layout:
    """Extents and descriptors of one spectral PrimaryCaps instance: x [N][H][W][Ci] (row stride ldx) ->
    y [N][OH][OW][Co] (row stride ldy), kernel KY x KX, W = P."""

    def __init__(self, N, H, W, Ci, ldx, Co, ldy, KY, KX):
        self.N, self.H, self.W, self.Ci, self.ldx, self.Co, self.ldy, self.KY, self.KX = N, H, W, Ci, ldx, Co, ldy, KY, KX
        self.OH, self.OW = H - KY + 1, W - KX + 1
        self.nu = n_freq(W)
        self.Ci2, self.Co2 = 2 * Ci, 2 * Co
        self.xhat_u = N * H * self.Ci2            # floats per frequency of X^ [u][n][iy][2Ci]
        self.yhat_u = N * self.OH * self.Co2      # floats per frequency of Y^ [u][n][oy][2Co]
        self.wg_u = self.Co2 * KY * self.Ci2      # floats per frequency of the real-form weights (either layout)

    # --- the four axis transforms
    def x_to_xhat(self):
        return axis(self.N * self.H, self.W, 2 * self.nu, self.Ci, self.W * self.ldx, self.ldx, 0, 1, self.Ci2, self.xhat_u, self.Ci, 2)

    def yhat_to_y(self, act, act_c0):
        return axis(self.N * self.OH, 2 * self.nu, self.OW, self.Co, self.Co2, self.yhat_u, self.Co, 2, self.OW * self.ldy, self.ldy, 0, 1,
                    act=act, act_c0=act_c0)

    def dy_to_dyhat(self, lddy):
        return axis(self.N * self.OH, self.OW, 2 * self.nu, self.Co, self.OW * lddy, lddy, 0, 1, self.Co2, self.yhat_u, self.Co, 2)

    def dxhat_to_dx(self, lddx, accum):
        return axis(self.N * self.H, 2 * self.nu, self.W, self.Ci, self.Ci2, self.xhat_u, self.Ci, 2, self.W * lddx, lddx, 0, 1, accum=int(accum))

    # --- the GEMMs
    def conv(self):
        d = D.conv_fwd(self.nu * self.N, (1, self.H, 1), self.Ci2, self.Ci2, self.Co2, self.Co2, (1, self.KY, 1), (1, 1, 1), (0, 0, 0),
                       (1, self.OH, 1), groups=self.nu)
        d["wgstride"] = self.wg_u
        return d

    def dgrad(self):
        out = D.transposed_classes(self.nu * self.N, (1, self.OH, 1), self.Co2, self.Co2, (1, self.H, 1), self.Ci2, self.Ci2,
                                   (1, self.KY, 1), (1, 1, 1), (0, 0, 0), groups=self.nu, ldw=self.Co2)
        for d in out:
            d["wgstride"] = self.wg_u
        return out

    def wgrad(self):
        """dWg[u] [2Co][KY][2Ci] = dY^[u]^T . X^[u] for every frequency in ONE launch (blockIdx.z = u): K is only the
        N*OH rows of one frequency (10 chunks at bs=8), so one slice each, plain stores, no zero-fill."""
        d = D.wgrad(self.N, (1, self.OH, 1), self.Co2, self.Co2, (1, self.H, 1), self.Ci2, self.Ci2, (1, self.KY, 1), (1, 1, 1), (0, 0, 0),
                    splitk=-1)
        d.update(nbatch=self.nu, dbstride=self.yhat_u, sbstride=self.xhat_u, gbstride=self.wg_u)
        return d

    def flops(self):
        """Issued-algorithmic FLOPs of the forward grouped conv (= dgrad = wgrad): 2 * rows * 2Co * 9 * 2Ci per frequency."""
        return 2 * self.nu * self.N * self.OH * self.Co2 * self.KY * self.Ci2


def primary_caps_fwd_bwd(x, w, bias, dy, act_c0=None):
    """Tensor-level runner (tests): x [N][H][W][Ci] cuda, w [Co][Ci][KY][KX], bias [Co], dy [N][OH][OW][Co] = gradient of
    the PRE-activation output.  -> (y incl. bias and sigmoid from act_c0, dx, dw [Co][Ci][KY][KX])."""
    import torch
    from . import ops
    dev = x.device
    N, H, W, Ci = x.shape
    Co, _, KY, KX = w.shape
    L = Layout(N, H, W, Ci, Ci, Co, Co, KY, KX)
    m = {k: torch.from_numpy(v).to(dev) for k, v in matrices(W, KX).items()}
    wf = w.reshape(Co, Ci, KY * KX).permute(0, 2, 1).contiguous()          # [Co][taps][Ci]
    wt = w.reshape(Co, Ci, KY * KX).permute(1, 2, 0).contiguous()          # [Ci][taps][Co]
    f32 = dict(device=dev, dtype=torch.float32)
    xhat = torch.empty(L.nu * L.xhat_u, **f32)
    ops.axis_linear(L.x_to_xhat(), x, m["F"], xhat)
    wg = torch.empty(L.nu * L.wg_u, **f32)
    ops.wspec_fwd(wf, m["tw"], Co, Ci, KY, KX, L.nu, 1, wg)
    yhat = torch.empty(L.nu * L.yhat_u, **f32)
    ops.conv_fwd(L.conv(), xhat, wg, yhat)
    y = torch.empty(N, L.OH, L.OW, Co, **f32)
    ops.axis_linear(L.yhat_to_y(capi.ACT_SIGMOID if act_c0 is not None else capi.ACT_NONE, act_c0 or 0), yhat, m["G"], y, bias=bias)
    # backward
    dyhat = torch.empty(L.nu * L.yhat_u, **f32)
    ops.axis_linear(L.dy_to_dyhat(Co), dy, m["Gt"], dyhat)
    dwg = torch.empty(L.nu * L.wg_u, **f32)
    ops.conv_wgrad(L.wgrad(), dyhat, xhat, dwg)
    kg = torch.empty(Co, KY * KX, Ci, **f32)
    ops.wspec_bwd(dwg, m["tw"], Co, Ci, KY, KX, L.nu, 1, kg)
    dw = kg.permute(0, 2, 1).reshape(Co, Ci, KY, KX)
    wgt = torch.empty(L.nu * L.wg_u, **f32)
    ops.wspec_fwd(wt, m["tw"], Ci, Co, KY, KX, L.nu, -1, wgt)
    dxhat = torch.empty(L.nu * L.xhat_u, **f32)
    for dd in L.dgrad():
        ops.conv_fwd(dd, dyhat, wgt, dxhat)
    dx = torch.empty(N, H, W, Ci, **f32)
    ops.axis_linear(L.dxhat_to_dx(Ci, False), dxhat, m["Ft"], dx)
    return y, dx, dw
