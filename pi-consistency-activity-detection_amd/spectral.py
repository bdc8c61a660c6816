"""Row-spectral form of PrimaryCaps (capsules_ucf101.py:43-49: Conv2d(832, 32*16 + 32, kernel 9, stride 1)).

y[n,oy,ox,co] = sum_{ci,ky,kx} w[co,ci,ky,kx] x[n,oy+ky,ox+kx,ci] is a correlation along x, so with a length-P
(P = input width) DFT along the rows   Y^[u] = X^[u] * conj(W^[u])   per frequency u = 0..P/2, and what is left is a
9-tap conv along y with complex channels.  No wrap-around reaches the valid outputs (ox + kx <= P-1).  The complex
product is taken with three real multiplications (X = Xr + i Xi, conj(W) = Wr - i Wi):

    t0 = (Xr + Xi) Wr,   t1 = Xi (Wr - Wi),   t2 = Xr (Wr + Wi);     Re Y = t0 - t1,   Im Y = t0 - t2

so the layer is ONE grouped real conv, 3 groups per complex frequency and 1 for DC / Nyquist (where everything is real),
Ci -> Co channels, 9x1 taps: (13*3 + 2)*9 = 369 real multiply-adds per (ci, co, output row) instead of 81*20 = 1620 -- under
a quarter of the direct form's FLOPs, equal to it in exact arithmetic.  The operand sums (Xr + Xi) and the result differences are folded into the DFT / inverse-DFT
matrices, so backward needs nothing but their transposes.  The GEMMs are the ordinary conv / wgrad kernels; the DFTs,
the weight planes and their adjoint are the three small kernels of csrc/spectral.hip.

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


def freq_order(P):
    """-> (complex frequencies, real frequencies): u = 1..ceil(P/2)-1 carry a complex spectrum and get three planes; DC and
    (even P) Nyquist are real for real data and get one.  Planes and twiddles are laid out complex first, then real."""
    nu = n_freq(P)
    real = [0] + ([nu - 1] if P % 2 == 0 else [])
    cplx = [u for u in range(nu) if u not in real]
    return cplx, real


def n_planes(P):
    c, r = freq_order(P)
    return 3 * len(c) + len(r)


def matrices(P, KX):
    """-> dict of float32 arrays for the three-multiplication form.  Planes: (Xr + Xi, Xi, Xr) per complex frequency, then
    Xr per real frequency (freq_order):
    F  [G][P]   x -> operand planes;       Ft = F^T   (operand-plane grads -> dx)
    G  [OW][G]  result planes -> y = Gr (t0 - t1) + Gi (t0 - t2)  (real frequencies: Gr t);   Gt = G^T   (dy -> result-plane grads)
    tw [nu][KX][2] in the same frequency order; also U, Ur."""
    OW = P - KX + 1
    cplx, real = freq_order(P)
    F2, G2 = dft_matrix(P), idft_matrix(P, OW)
    Fr, Fi = F2[0::2], F2[1::2]
    Gr, Gi = G2[:, 0::2], G2[:, 1::2]
    Frows, Gcols = [], []
    for u in cplx:
        Frows += [Fr[u] + Fi[u], Fi[u], Fr[u]]
        Gcols += [Gr[:, u] + Gi[:, u], -Gr[:, u], -Gi[:, u]]
    for u in real:
        Frows.append(Fr[u])
        Gcols.append(Gr[:, u])
    F = np.stack(Frows, 0)
    G = np.stack(Gcols, 1)
    tw = twiddles(P, KX)[cplx + real]
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(F=f32(F), Ft=f32(F.T), G=f32(G), Gt=f32(G.T), tw=f32(tw))


def matrices_full(P, W, KX):
    """The same for a FULL convolution along x (stride-1 ConvTranspose: y = x * w, W + KX - 1 = P outputs from W inputs):
    Y^ = X^ W^ without the conjugate.  With the conjugate twiddle table (cos, +sin) the weight-plane kernels produce
    V1 = Wr + Wi and V2 = -(Wi - Wr), for which Re = t0 - t1, Im = t0 - t2 again, so F (restricted to the W input columns)
    and G (all P outputs) keep their form."""
    assert W + KX - 1 == P
    m = matrices(P, 1)                       # OW = P: every output column
    F = np.ascontiguousarray(m["F"][:, :W])
    cplx, real = freq_order(P)
    tw = twiddles(P, KX)[cplx + real].copy()
    tw[..., 1] = -tw[..., 1]
    return dict(F=F, Ft=np.ascontiguousarray(F.T), G=m["G"], Gt=m["Gt"], tw=np.ascontiguousarray(tw, dtype=np.float32))


def axis(R, I, O, C, in_sr, in_hi, in_lo, in_split, out_sr, out_hi, out_lo, out_split, act=0, act_c0=0, accum=0):
    return dict(R=R, I=I, O=O, C=C, in_split=in_split, out_split=out_split, act=act, act_c0=act_c0, accum=accum,
                in_sr=in_sr, in_hi=in_hi, in_lo=in_lo, out_sr=out_sr, out_hi=out_hi, out_lo=out_lo)


class Layout:
    """Extents and descriptors of one spectral PrimaryCaps instance: x [N][H][W][Ci] (row stride ldx) ->
    y [N][OH][OW][Co] (row stride ldy), kernel KY x KX, W = P.  Groups g = 3*u + j; operand planes X [g][n][iy][Ci],
    result planes T [g][n][oy][Co], weight planes [g][Co][KY][Ci] (forward) / [g][Ci][KY][Co] (dgrad)."""

    def __init__(self, N, H, W, Ci, ldx, Co, ldy, KY, KX):
        self.N, self.H, self.W, self.Ci, self.ldx, self.Co, self.ldy, self.KY, self.KX = N, H, W, Ci, ldx, Co, ldy, KY, KX
        self.OH, self.OW = H - KY + 1, W - KX + 1
        self.nu = n_freq(W)
        self.Ur = len(freq_order(W)[1])
        self.G = n_planes(W)
        self.x_g = N * H * Ci                     # floats per operand plane
        self.t_g = N * self.OH * Co               # floats per result plane
        self.w_g = Co * KY * Ci                   # floats per weight plane (either layout)

    # --- the four axis transforms (index o or i = 3*u + j: offset = g * plane)
    def x_to_planes(self):
        return axis(self.N * self.H, self.W, self.G, self.Ci, self.W * self.ldx, self.ldx, 0, 1, self.Ci, self.x_g, 0, 1)

    def planes_to_y(self, act, act_c0):
        return axis(self.N * self.OH, self.G, self.OW, self.Co, self.Co, self.t_g, 0, 1, self.OW * self.ldy, self.ldy, 0, 1,
                    act=act, act_c0=act_c0)

    def dy_to_planes(self, lddy):
        return axis(self.N * self.OH, self.OW, self.G, self.Co, self.OW * lddy, lddy, 0, 1, self.Co, self.t_g, 0, 1)

    def planes_to_dx(self, lddx, accum):
        return axis(self.N * self.H, self.G, self.W, self.Ci, self.Ci, self.x_g, 0, 1, self.W * lddx, lddx, 0, 1, accum=int(accum))

    # --- the GEMMs
    def conv(self):
        d = D.conv_fwd(self.G * self.N, (1, self.H, 1), self.Ci, self.Ci, self.Co, self.Co, (1, self.KY, 1), (1, 1, 1), (0, 0, 0),
                       (1, self.OH, 1), groups=self.G)
        d["wgstride"] = self.w_g
        return d

    def dgrad(self):
        # sample-fastest row order inside each group: a 64-row tile is 4 rows of y for all samples, so the taps that only
        # reach the zero padding of the 'full' correlation (H rows gathered from OH real ones) are skipped per tile
        out = D.transposed_classes(self.G * self.N, (1, self.OH, 1), self.Co, self.Co, (1, self.H, 1), self.Ci, self.Ci,
                                   (1, self.KY, 1), (1, 1, 1), (0, 0, 0), groups=self.G, ldw=self.Co, flags=capi.F_NFAST)
        for d in out:
            d["wgstride"] = self.w_g
        return out

    def wgrad(self):
        """dV[g] [Co][KY][Ci] = dT[g]^T . X[g] for every group in ONE launch (blockIdx.z = g): K is only the N*OH rows
        of one plane (10 chunks at bs=8), so one slice each, plain stores, no zero-fill."""
        # the rows of a plane are described as the w axis (memory is the same: one position per (n, row)), which is what the
        # row-segment wgrad kernel shares its LDS tile along: the KY taps read one tile of OH + KY - 1 rows
        d = D.wgrad(self.N, (1, 1, self.OH), self.Co, self.Co, (1, 1, self.H), self.Ci, self.Ci, (1, 1, self.KY), (1, 1, 1), (0, 0, 0),
                    splitk=-1)
        d.update(nbatch=self.G, dbstride=self.t_g, sbstride=self.x_g, gbstride=self.w_g)
        return d

    def flops(self):
        """Issued-algorithmic FLOPs of the forward grouped conv (= dgrad = wgrad): 2 * rows * Co * 9 * Ci per group."""
        return 2 * self.G * self.N * self.OH * self.Co * self.KY * self.Ci


class LayoutT:
    """Stride-1 ConvTranspose2d with a KY x KX kernel (upsample1, capsules_ucf101.py:375,488): x [N][H][W][Ci] (row stride
    ldx) -> y [N][H+KY-1][W+KX-1][Co] (row stride ldy).  Same planes as Layout; along y the grouped GEMM is the
    transposed (scatter-adjoint) form, its dgrad an ordinary grouped conv, the weight gradient in ConvTranspose
    convention dV[g] [Ci][KY][Co] = X[g]^T . dT[g]."""

    def __init__(self, N, H, W, Ci, ldx, Co, ldy, KY, KX):
        self.N, self.H, self.W, self.Ci, self.ldx, self.Co, self.ldy, self.KY, self.KX = N, H, W, Ci, ldx, Co, ldy, KY, KX
        self.OH, self.OW = H + KY - 1, W + KX - 1
        self.P = self.OW
        self.nu = n_freq(self.P)
        self.Ur = len(freq_order(self.P)[1])
        self.G = n_planes(self.P)
        self.x_g = N * H * Ci
        self.t_g = N * self.OH * Co
        self.w_g = Co * KY * Ci

    def matrices(self):
        return matrices_full(self.P, self.W, self.KX)

    def x_to_planes(self):
        return axis(self.N * self.H, self.W, self.G, self.Ci, self.W * self.ldx, self.ldx, 0, 1, self.Ci, self.x_g, 0, 1)

    def planes_to_y(self, act, act_c0):
        return axis(self.N * self.OH, self.G, self.OW, self.Co, self.Co, self.t_g, 0, 1, self.OW * self.ldy, self.ldy, 0, 1,
                    act=act, act_c0=act_c0)

    def dy_to_planes(self, lddy):
        return axis(self.N * self.OH, self.OW, self.G, self.Co, self.OW * lddy, lddy, 0, 1, self.Co, self.t_g, 0, 1)

    def planes_to_dx(self, lddx, accum):
        return axis(self.N * self.H, self.G, self.W, self.Ci, self.Ci, self.x_g, 0, 1, self.W * lddx, lddx, 0, 1, accum=int(accum))

    def convT(self):
        out = D.transposed_classes(self.G * self.N, (1, self.H, 1), self.Ci, self.Ci, (1, self.OH, 1), self.Co, self.Co,
                                   (1, self.KY, 1), (1, 1, 1), (0, 0, 0), groups=self.G, flags=capi.F_NFAST)
        for d in out:
            d["wgstride"] = self.w_g
        return out

    def dgrad(self):
        d = D.conv_fwd(self.G * self.N, (1, self.OH, 1), self.Co, self.Co, self.Ci, self.Ci, (1, self.KY, 1), (1, 1, 1), (0, 0, 0),
                       (1, self.H, 1), groups=self.G, ldw=self.Co)
        d["wgstride"] = self.w_g
        return d

    def wgrad(self):
        d = D.wgrad(self.N, (1, 1, self.H), self.Ci, self.Ci, (1, 1, self.OH), self.Co, self.Co, (1, 1, self.KY), (1, 1, 1), (0, 0, 0),
                    splitk=-1)
        d.update(nbatch=self.G, dbstride=self.x_g, sbstride=self.t_g, gbstride=self.w_g)
        return d

    def flops(self):
        """Issued-algorithmic FLOPs of one grouped GEMM: 2 * input rows * Co * KY * Ci per group."""
        return 2 * self.G * self.N * self.H * self.Co * self.KY * self.Ci


def conv_transpose_fwd_bwd(x, w, bias, dy):
    """Tensor-level runner (tests): x [N][H][W][Ci] cuda, w [Ci][Co][KY][KX] (ConvTranspose2d layout), bias [Co],
    dy [N][OH][OW][Co] = gradient of the pre-activation output.  -> (relu(y), dx, dw [Ci][Co][KY][KX])."""
    import torch
    from . import ops
    dev = x.device
    N, H, W, Ci = x.shape
    _, Co, KY, KX = w.shape
    L = LayoutT(N, H, W, Ci, Ci, Co, Co, KY, KX)
    m = {k: torch.from_numpy(v).to(dev) for k, v in L.matrices().items()}
    wf = w.reshape(Ci, Co, KY * KX).permute(1, 2, 0).contiguous()          # [Co][taps][Ci]
    wt = w.reshape(Ci, Co, KY * KX).permute(0, 2, 1).contiguous()          # [Ci][taps][Co]
    f32 = dict(device=dev, dtype=torch.float32)
    xp = torch.empty(L.G * L.x_g, **f32)
    ops.axis_linear(L.x_to_planes(), x, m["F"], xp)
    wv = torch.empty(L.G * L.w_g, **f32)
    ops.wspec_fwd(wf, m["tw"], Co, Ci, KY, KX, L.nu, L.Ur, wv)
    tp = torch.zeros(L.G * L.t_g, **f32)
    for dd in L.convT():
        ops.conv_fwd(dd, xp, wv, tp)
    y = torch.empty(N, L.OH, L.OW, Co, **f32)
    ops.axis_linear(L.planes_to_y(capi.ACT_RELU, 0), tp, m["G"], y, bias=bias)
    dtp = torch.empty(L.G * L.t_g, **f32)
    ops.axis_linear(L.dy_to_planes(Co), dy, m["Gt"], dtp)
    dv = torch.empty(L.G * L.w_g, **f32)
    ops.conv_wgrad(L.wgrad(), xp, dtp, dv)
    kg = torch.empty(Ci, KY * KX, Co, **f32)
    ops.wspec_bwd(dv, m["tw"], Ci, Co, KY, KX, L.nu, L.Ur, kg)
    dw = kg.permute(0, 2, 1).reshape(Ci, Co, KY, KX)
    wvt = torch.empty(L.G * L.w_g, **f32)
    ops.wspec_fwd(wt, m["tw"], Ci, Co, KY, KX, L.nu, L.Ur, wvt)
    dxp = torch.empty(L.G * L.x_g, **f32)
    ops.conv_fwd(L.dgrad(), dtp, wvt, dxp)
    dx = torch.empty(N, H, W, Ci, **f32)
    ops.axis_linear(L.planes_to_dx(Ci, False), dxp, m["Ft"], dx)
    return y, dx, dw


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
    xp = torch.empty(L.G * L.x_g, **f32)
    ops.axis_linear(L.x_to_planes(), x, m["F"], xp)
    wv = torch.empty(L.G * L.w_g, **f32)
    ops.wspec_fwd(wf, m["tw"], Co, Ci, KY, KX, L.nu, L.Ur, wv)
    tp = torch.empty(L.G * L.t_g, **f32)
    ops.conv_fwd(L.conv(), xp, wv, tp)
    y = torch.empty(N, L.OH, L.OW, Co, **f32)
    ops.axis_linear(L.planes_to_y(capi.ACT_SIGMOID if act_c0 is not None else capi.ACT_NONE, act_c0 or 0), tp, m["G"], y, bias=bias)
    # backward
    dtp = torch.empty(L.G * L.t_g, **f32)
    ops.axis_linear(L.dy_to_planes(Co), dy, m["Gt"], dtp)
    dv = torch.empty(L.G * L.w_g, **f32)
    ops.conv_wgrad(L.wgrad(), dtp, xp, dv)
    kg = torch.empty(Co, KY * KX, Ci, **f32)
    ops.wspec_bwd(dv, m["tw"], Co, Ci, KY, KX, L.nu, L.Ur, kg)
    dw = kg.permute(0, 2, 1).reshape(Co, Ci, KY, KX)
    wvt = torch.empty(L.G * L.w_g, **f32)
    ops.wspec_fwd(wt, m["tw"], Ci, Co, KY, KX, L.nu, L.Ur, wvt)
    dxp = torch.empty(L.G * L.x_g, **f32)
    for dd in L.dgrad():
        ops.conv_fwd(dd, dtp, wvt, dxp)
    dx = torch.empty(N, H, W, Ci, **f32)
    ops.axis_linear(L.planes_to_dx(Ci, False), dxp, m["Ft"], dx)
    return y, dx, dw
