"""GPU parity of every HIP kernel family (through the C-ABI) against the CPU oracle / plain torch
fp32 on the same seeded inputs.  Tolerances: fp32 contraction noise only (1e-4 relative)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import caps as ocaps, i3d as oi3d, losses as olosses
from picons_amd import capi, desc, ops, spec

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cl(x):
    """NCDHW cpu -> NDHWC cuda contiguous fp32"""
    return x.detach().permute(0, 2, 3, 4, 1).contiguous().float().to(DEV)


def uncl(x):
    return x.detach().cpu().permute(0, 4, 1, 2, 3).contiguous()


def close(a, b, rtol=2e-4, atol=None, what=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = b.abs().max().item() + 1e-30
    atol = rtol * scale if atol is None else atol
    err = (a - b).abs().max().item()
    assert err <= atol, "%s: max err %.3e > %.3e (scale %.3e)" % (what, err, atol, scale)


def w_oki(w, pad_to=None):
    O, I = w.shape[:2]
    taps = int(np.prod(w.shape[2:]))
    t = w.detach().reshape(O, I, taps).permute(0, 2, 1).contiguous()
    if pad_to and pad_to > I:
        t = F.pad(t, (0, pad_to - I))
    return t.contiguous().float().to(DEV)


def w_iko(w):
    O, I = w.shape[:2]
    taps = int(np.prod(w.shape[2:]))
    return w.detach().reshape(O, I, taps).permute(1, 2, 0).contiguous().float().to(DEV)


CONV_CASES = [
    (16, 24, (3, 3, 3), (2, 1, 1), (4, 12, 12), 2),
    (4, 64, (7, 7, 7), (2, 2, 2), (8, 30, 30), 2),
    (4, 64, (7, 7, 7), (2, 2, 2), (8, 12, 56), 2),        # stem shape with 28-position row segments: wgrad4_kernel, 8-tap LDS-DMA conv
    (4, 24, (7, 7, 7), (2, 2, 2), (5, 9, 112), 1),        # fewer than 64 output channels, odd extents along t / h
    (64, 16, (1, 1, 1), (1, 1, 1), (2, 9, 9), 3),
    (32, 200, (3, 3, 3), (1, 1, 1), (2, 14, 14), 2),
    (48, 136, (1, 3, 3), (1, 1, 1), (1, 20, 20), 4),
    (64, 192, (3, 3, 3), (2, 1, 1), (4, 6, 56), 2),       # temporal stride 2: row-segment wgrad with 64-channel blocks
    (96, 128, (3, 3, 3), (1, 1, 1), (2, 5, 28), 2),       # row-segment wgrad with 32-channel blocks
    (160, 320, (3, 3, 3), (1, 1, 1), (1, 4, 28), 2),
    (64, 64, (3, 3, 3), (1, 1, 1), (2, 16, 32), 2),       # 128 x 64 tiles on whole 8 x 16 position blocks
    (32, 192, (3, 3, 3), (2, 1, 1), (4, 24, 48), 1),      # temporal stride 2, three 64-column tiles
    (96, 64, (1, 3, 3), (1, 1, 1), (1, 8, 16), 4),        # 1x3x3 taps, one tile per plane
]


@pytest.mark.parametrize("Ci,Co,k,s,thw,N", CONV_CASES)
def test_conv_same_fwd_dgrad_wgrad(Ci, Co, k, s, thw, N):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Ci, *thw, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))).requires_grad_(True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, s)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    othw = tuple(y.shape[2:]); pf = [p[0] for p in pads]
    xg, dyg = cl(x), cl(dy)
    out = torch.empty(N, *othw, Co, device=DEV)
    ops.conv_fwd(desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, s, pf, othw), xg, w_oki(w), out)
    close(uncl(out), y, what="fwd")
    dx = torch.full((N, *thw, Ci), 7.0, device=DEV)
    for dd in desc.transposed_classes(N, othw, Co, Co, thw, Ci, Ci, k, s, pf):
        ops.conv_fwd(dd, dyg, w_iko(w), dx)
    close(uncl(dx), x.grad, what="dgrad")
    taps = int(np.prod(k))
    gw = torch.zeros(Co, taps, Ci, device=DEV)
    ops.conv_wgrad(desc.wgrad(N, othw, Co, Co, thw, Ci, Ci, k, s, pf), dyg, xg, gw)
    close(gw.cpu(), w.grad.reshape(Co, Ci, taps).permute(0, 2, 1), what="wgrad")


@pytest.mark.parametrize("Co,thw,N", [(64, (8, 12, 56), 2), (24, (5, 9, 112), 1), (64, (4, 28, 56), 3)])
def test_stem_padding_channel_flags(Co, thw, N):
    """PC_F_CI3 / PC_WG_CS3: the 4th channel of the RGB clip's 16-byte pieces is padding.  The forward must give the 3-channel
    convolution whatever that channel and its weights hold (its MFMAs are not issued), the weight gradient must be right in the
    three real channels and leave the padding column of g alone (the packed 5-accumulator wgrad4 kernel)."""
    from picons_amd import capi
    g = torch.Generator().manual_seed(9)
    k, s_ = (7, 7, 7), (2, 2, 2)
    x3 = torch.randn(N, 3, *thw, generator=g)
    w3 = (torch.randn(Co, 3, *k, generator=g) / np.sqrt(3 * 343)).requires_grad_(True)
    pads = [spec.same_pad(thw[i], k[i], s_[i]) for i in range(3)]
    xp = F.pad(x3, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w3, None, s_)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    othw = tuple(y.shape[2:]); pf = [p[0] for p in pads]
    junk = torch.randn(N, 1, *thw, generator=g) * 3.0
    x4 = cl(torch.cat([x3, junk], 1))                                  # garbage in the padding channel
    w4 = torch.cat([w3.detach(), torch.randn(Co, 1, *k, generator=g)], 1)   # and in its weights
    out = torch.empty(N, *othw, Co, device=DEV)
    d = desc.conv_fwd(N, thw, 4, 4, Co, Co, k, s_, pf, othw, flags=capi.F_CI3)
    ops.conv_fwd(d, x4, w_oki(w4), out)
    close(uncl(out), y, what="fwd with PC_F_CI3")
    x4z = cl(torch.cat([x3, torch.zeros(N, 1, *thw)], 1))
    gw = torch.full((Co, 343, 4), 5.0, device=DEV)
    gw[:, :, :3] = 0
    wd = desc.wgrad(N, othw, Co, Co, thw, 4, 4, k, s_, pf)
    wd["flags"] = capi.WG_CS3
    ops.conv_wgrad(wd, cl(dy), x4z, gw)
    close(gw[:, :, :3].cpu(), w3.grad.reshape(Co, 3, 343).permute(0, 2, 1), what="wgrad, real channels")
    assert torch.equal(gw[:, :, 3].cpu(), torch.full((Co, 343), 5.0)), "padding column of g was written"
    gw2 = torch.zeros(Co, 343, 4, device=DEV)                            # the same launch without the flag agrees
    ops.conv_wgrad(desc.wgrad(N, othw, Co, Co, thw, 4, 4, k, s_, pf), cl(dy), x4z, gw2)
    close(gw2[:, :, :3].cpu(), gw[:, :, :3].cpu(), what="packed vs 7-accumulator wgrad4")


def test_wgrad_primarycaps_shape_remainder_rows():
    """PrimaryCaps weight gradient at its real channel counts (544 = 4*128 + 32 rows, 81*832 columns): the grid is
    deep enough that the last 32 rows go to a second launch with 64-row tiles; both row ranges are checked
    against torch on their own scale."""
    g = torch.Generator().manual_seed(11)
    N, Ci, Co, hw, k = 2, 832, 544, 28, 9
    x = torch.randn(N, Ci, hw, hw, generator=g)
    w = (torch.randn(Co, Ci, k, k, generator=g) * 0.01).requires_grad_(True)
    y = F.conv2d(x, w)
    dy = torch.randn(y.shape, generator=g)
    dy[:, 512:] *= 1e-3                                    # the activation-capsule rows carry much smaller gradients
    y.backward(dy)
    o = y.shape[-1]
    xg = x.permute(0, 2, 3, 1).contiguous().view(N, 1, hw, hw, Ci).to(DEV)
    dyg = dy.permute(0, 2, 3, 1).contiguous().view(N, 1, o, o, Co).to(DEV)
    gw = torch.zeros(Co, k * k, Ci, device=DEV)
    ops.conv_wgrad(desc.wgrad(N, (1, o, o), Co, Co, (1, hw, hw), Ci, Ci, (1, k, k), (1, 1, 1), (0, 0, 0)), dyg, xg, gw)
    ref = w.grad.reshape(Co, Ci, k * k).permute(0, 2, 1)
    close(gw[:512].cpu(), ref[:512], what="rows 0..511 (128-row tiles)")
    close(gw[512:].cpu(), ref[512:], what="rows 512..543 (64-row remainder launch)")


@pytest.mark.parametrize("N,HW,Ci,Co,K", [(3, 14, 32, 40, 5), (2, 13, 16, 24, 3), (2, 28, 832, 544, 9)])
def test_primary_caps_row_spectral_form_vs_torch(N, HW, Ci, Co, K):
    """PrimaryCaps as DFT along the rows + grouped 9x1 complex conv + inverse DFT (spectral.py, csrc/spectral.hip):
    output (bias, sigmoid on the activation channels), input gradient and weight gradient against F.conv2d + autograd;
    even and odd widths, and the real layer size."""
    from picons_amd import spectral
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Ci, HW, HW, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, K, K, generator=g) / np.sqrt(Ci * K * K)).requires_grad_(True)
    b = torch.randn(Co, generator=g) * 0.1
    c0 = Co - 8
    z = F.conv2d(x, w, b)
    y = torch.cat([z[:, :c0], torch.sigmoid(z[:, c0:])], 1)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dzg = dz.permute(0, 2, 3, 1).contiguous().to(DEV)
    yg, dxg, dwg = spectral.primary_caps_fwd_bwd(xg, w.detach().to(DEV), b.to(DEV), dzg, act_c0=c0)
    close(yg.permute(0, 3, 1, 2), y, rtol=2e-5, what="output")
    close(dxg.permute(0, 3, 1, 2), x.grad, what="dgrad")
    close(dwg, w.grad, what="wgrad")


@pytest.mark.parametrize("N,HW,Ci,Co,K", [(3, 10, 24, 16, 5), (2, 9, 8, 12, 3), (2, 20, 384, 64, 9)])
def test_conv_transpose_row_spectral_form_vs_torch(N, HW, Ci, Co, K):
    """Stride-1 ConvTranspose2d (upsample1) in the row-spectral form (full convolution along x: conjugate twiddles,
    grouped transposed conv along y): relu(output), input gradient, weight gradient against torch."""
    from picons_amd import spectral
    g = torch.Generator().manual_seed(22)
    x = torch.randn(N, Ci, HW, HW, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, K, K, generator=g) / np.sqrt(Ci * K * K)).requires_grad_(True)
    b = torch.randn(Co, generator=g) * 0.1
    z = F.conv_transpose2d(x, w, b)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dzg = dz.permute(0, 2, 3, 1).contiguous().to(DEV)
    yg, dxg, dwg = spectral.conv_transpose_fwd_bwd(xg, w.detach().to(DEV), b.to(DEV), dzg)
    close(yg.permute(0, 3, 1, 2), torch.relu(z), rtol=2e-5, what="output")
    close(dxg.permute(0, 3, 1, 2), x.grad, what="dgrad")
    close(dwg, w.grad, what="wgrad")


def test_transpose_multi_many_jobs():
    """pc_transpose_multi: 97 independent jobs (more than one argument pack) of ragged sizes, batched and accumulating
    ones included, against torch."""
    g = torch.Generator().manual_seed(31)
    jobs, checks = [], []
    for q in range(97):
        B = 1 + q % 3; R = 1 + (q * 7) % 70; Cc = 1 + (q * 13) % 90
        src = torch.randn(B, R, Cc, generator=g).to(DEV)
        accum = q % 5 == 0
        dst = torch.randn(B, Cc, R, generator=g).to(DEV) if accum else torch.full((B, Cc, R), 7.0, device=DEV)
        want = src.transpose(1, 2) + (dst if accum else 0)
        jobs.append((src, dst, B, R, Cc, R * Cc, Cc, Cc * R, R, accum))
        checks.append((dst, want.clone()))
    ops.transpose_multi(jobs)
    torch.cuda.synchronize()
    for dst, want in checks:
        assert torch.equal(dst, want)


def test_axis_linear_general_strides():
    """pc_axis_linear with two-level input and output indices, bias, ReLU from a channel on, and accumulation."""
    g = torch.Generator().manual_seed(32)
    R, I, O, C_ = 5, 6, 4, 8
    x = torch.randn(3, R, 2, C_, generator=g)            # index i = (i_hi in 0..2, i_lo in 0..1): [i_hi][r][i_lo][c]
    M = torch.randn(O, I, generator=g)
    b = torch.randn(C_, generator=g)
    old = torch.randn(R, 2, 2, C_, generator=g)          # o = (o_hi, o_lo): [r][o_hi][o_lo][c]
    xi = x.permute(1, 0, 2, 3).reshape(R, I, C_)         # [r][i][c]
    want = torch.einsum("oi,ric->roc", M, xi) + b + old.reshape(R, O, C_)
    want[:, :, 4:] = torch.relu(want[:, :, 4:])
    d = dict(R=R, I=I, O=O, C=C_, in_split=2, out_split=2, act=capi.ACT_RELU, act_c0=4, accum=1,
             in_sr=2 * C_, in_hi=R * 2 * C_, in_lo=C_, out_sr=4 * C_, out_hi=2 * C_, out_lo=C_)
    out = old.clone().to(DEV)
    ops.axis_linear(d, x.to(DEV), M.to(DEV), out, bias=b.to(DEV))
    close(out.cpu().reshape(R, O, C_), want, rtol=1e-5, what="axis_linear")


def test_weight_planes_from_master_layout_match_generic_path():
    """pc_wspec_master_fwd / _bwd (planes straight from / to the OIHW master tensor, two row ranges like pose + activation
    capsules) against pc_wspec_fwd / _bwd on the kernel layouts."""
    from picons_amd import spectral
    g = torch.Generator().manual_seed(41)
    A1, A2, B, K, P = 24, 8, 16, 9, 28
    A = A1 + A2
    w1 = torch.randn(A1, B, K, K, generator=g).to(DEV); w2 = torch.randn(A2, B, K, K, generator=g).to(DEV)
    w = torch.cat([w1, w2], 0)
    m = {k: torch.from_numpy(v).to(DEV) for k, v in spectral.matrices(P, K).items()}
    U, Ur, G = spectral.n_freq(P), len(spectral.freq_order(P)[1]), spectral.n_planes(P)
    wf = w.reshape(A, B, K * K).permute(0, 2, 1).contiguous(); wt = w.reshape(A, B, K * K).permute(1, 2, 0).contiguous()
    ref_f = torch.empty(G, A, K, B, device=DEV); ref_t = torch.empty(G, B, K, A, device=DEV)
    ops.wspec_fwd(wf, m["tw"], A, B, K, K, U, Ur, ref_f)
    ops.wspec_fwd(wt, m["tw"], B, A, K, K, U, Ur, ref_t)
    out_f = torch.full_like(ref_f, 5.0); out_t = torch.full_like(ref_t, 5.0)
    ops.wspec_master_fwd(w1, m["tw"], A1, 0, A, B, K, K, U, Ur, out_f, out_t)
    ops.wspec_master_fwd(w2, m["tw"], A2, A1, A, B, K, K, U, Ur, out_f, out_t)
    close(out_f, ref_f, rtol=1e-5, what="forward-layout planes"); close(out_t, ref_t, rtol=1e-5, what="dgrad-layout planes")
    dV = torch.randn(G, A, K, B, generator=g).to(DEV)
    kg = torch.empty(A, K * K, B, device=DEV)
    ops.wspec_bwd(dV, m["tw"], A, B, K, K, U, Ur, kg)
    want = kg.permute(0, 2, 1).reshape(A, B, K, K)
    d1 = torch.full((A1, B, K, K), 5.0, device=DEV); d2 = torch.ones(A2, B, K, K, device=DEV)
    ops.wspec_master_bwd(dV, m["tw"], A1, 0, A, B, K, K, U, Ur, d1, accum=False)
    ops.wspec_master_bwd(dV, m["tw"], A2, A1, A, B, K, K, U, Ur, d2, accum=True)
    close(d1, want[:A1], rtol=1e-5, what="master gradient"); close(d2 - 1.0, want[A1:], rtol=1e-5, what="master gradient (accumulate)")


@pytest.mark.parametrize("Ci,Co,thw,N,trim", [(64, 64, (2, 5, 56), 2, False), (64, 192, (3, 4, 28), 2, False), (128, 96, (1, 6, 28), 3, True),
                                              (64, 48, (2, 3, 112), 1, False)])
def test_wgrad_row_segment_kernel(Ci, Co, thw, N, trim):
    """Stride-1 SAME 3x3x3 weight gradient through the row-segment kernel (K chunk = a segment of one image row, the three
    kw taps share one LDS tile): widths 28 / 56 / 112, 64- and 128-row tiles, ragged Cd, and the 1x3x3 trimmed form of a
    one-frame input."""
    g = torch.Generator().manual_seed(51)
    x = torch.randn(N, Ci, *thw, generator=g)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1).requires_grad_(True)
    y = F.conv3d(x, w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    d = desc.wgrad(N, thw, Co, Co, thw, Ci, Ci, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    if trim:
        d = desc.trim_wgrad(d)
        assert d["ntap"][0] == 1
    gw = torch.zeros(Co, 27, Ci, device=DEV)
    ops.conv_wgrad(d, cl(dy), cl(x), gw)
    close(gw.cpu(), w.grad.reshape(Co, Ci, 27).permute(0, 2, 1), what="wgrad (row-segment kernel)")


def test_conv_epilogue_bias_act_cscale_accum_slice():
    g = torch.Generator().manual_seed(6)
    N, Ci, Co, thw = 2, 8, 40, (2, 6, 6)
    x = torch.randn(N, Ci, *thw, generator=g); w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g); cs = (torch.rand(N, Co, generator=g) < 0.5).float() * 2
    ref = torch.relu(F.conv3d(x, w, b, padding=1)) * cs.view(N, Co, 1, 1, 1)
    wide = torch.full((N, *thw, 64), 3.0, device=DEV)          # write into channel slice [8, 48) of a 64-wide tensor
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), thw, act=capi.ACT_RELU,
                      flags=capi.F_BIAS | capi.F_CSCALE)
    ops.conv_fwd(d, cl(x), w_oki(w), wide[..., 8:], bias=b.to(DEV), cscale=cs.to(DEV))
    close(uncl(wide[..., 8:48]), ref, what="epilogue")
    assert (wide[..., :8] == 3.0).all() and (wide[..., 48:] == 3.0).all()
    d["flags"] |= capi.F_ACCUM
    ops.conv_fwd(d, cl(x), w_oki(w), wide[..., 8:], bias=b.to(DEV), cscale=cs.to(DEV))
    close(uncl(wide[..., 8:48]), 2 * ref, what="accumulate")
    d2 = desc.conv_fwd(N, thw, Ci, Ci, 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), thw, act=capi.ACT_SIGMOID, flags=capi.F_BIAS)
    o2 = torch.empty(N, *thw, 32, device=DEV)
    ops.conv_fwd(d2, cl(x), w_oki(w[:32]), o2, bias=b[:32].to(DEV))
    close(uncl(o2), torch.sigmoid(F.conv3d(x, w[:32], b[:32], padding=1)), what="sigmoid")


TCASES = [
    (64, 96, (1, 9, 9), (1, 1, 1), (0, 0, 0), (0, 0, 0), (1, 20, 20), 6),
    (128, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (1, 7, 7), 2),
    (16, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (2, 6, 5), 2),
    (48, 64, (1, 9, 9), (1, 1, 1), (0, 0, 0), (0, 0, 0), (1, 6, 6), 2),
    (128, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 6, 6), 1),
]


@pytest.mark.parametrize("Ci,Co,k,s,pd,op,thw,N", TCASES)
def test_conv_transpose(Ci, Co, k, s, pd, op, thw, N):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Ci, *thw, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, *k, generator=g) * 0.05).requires_grad_(True)
    y = F.conv_transpose3d(x, w, None, s, pd, op)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    othw = tuple(y.shape[2:]); taps = int(np.prod(k))
    wt_oki = w.detach().reshape(Ci, Co, taps).permute(1, 2, 0).contiguous().to(DEV)
    wt_iko = w.detach().reshape(Ci, Co, taps).permute(0, 2, 1).contiguous().to(DEV)
    out = torch.full((N, *othw, Co), -5.0, device=DEV)
    for dd in desc.transposed_classes(N, thw, Ci, Ci, othw, Co, Co, k, s, pd):
        ops.conv_fwd(dd, cl(x), wt_oki, out)
    close(uncl(out), y, what="convT fwd")
    if s == (1, 1, 1):      # n-fastest row order + tile-level tap skipping must not change the result
        out2 = torch.full((N, *othw, Co), -5.0, device=DEV)
        for dd in desc.transposed_classes(N, thw, Ci, Ci, othw, Co, Co, k, s, pd, flags=capi.F_NFAST):
            ops.conv_fwd(dd, cl(x), wt_oki, out2)
        close(uncl(out2), y, what="convT fwd (NFAST)")
    dx = torch.empty(N, *thw, Ci, device=DEV)
    ops.conv_fwd(desc.conv_fwd(N, othw, Co, Co, Ci, Ci, k, s, pd, thw), cl(dy), wt_iko, dx)
    close(uncl(dx), x.grad, what="convT dgrad")
    gw = torch.zeros(Ci, taps, Co, device=DEV)
    ops.conv_wgrad(desc.wgrad(N, thw, Ci, Ci, othw, Co, Co, k, s, pd), cl(x), cl(dy), gw)
    close(gw.cpu(), w.grad.reshape(Ci, Co, taps).permute(0, 2, 1), what="convT wgrad")


@pytest.mark.parametrize("groups", [1, 2])
def test_unit3d_bn_relu_fwd_bwd(groups):
    """conv (BN partials in the epilogue) -> finalize -> apply, and the backward, vs oracle Unit3D
    run once per group (the reference's two sequential forward passes)."""
    g = torch.Generator().manual_seed(8)
    N, Ci, Co, thw, k, s = 4, 16, 40, (4, 10, 10), (3, 3, 3), (2, 1, 1)
    x = torch.randn(N, Ci, *thw, generator=g)
    w = torch.randn(Co, Ci, *k, generator=g) * 0.1
    gamma = torch.rand(Co, generator=g) + 0.5; beta = torch.randn(Co, generator=g) * 0.1
    P = {"u.conv3d.weight": w.clone().requires_grad_(True), "u.bn.weight": gamma.clone().requires_grad_(True),
         "u.bn.bias": beta.clone().requires_grad_(True), "u.bn.running_mean": torch.zeros(Co), "u.bn.running_var": torch.ones(Co)}
    xs = x.clone().requires_grad_(True)
    ys = [oi3d.unit3d(P, "u", xs[i * (N // groups):(i + 1) * (N // groups)], s, True) for i in range(groups)]
    y = torch.cat(ys)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    othw = tuple(y.shape[2:]); pf = [spec.same_pad(thw[i], k[i], s[i])[0] for i in range(3)]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, s, pf, othw, flags=capi.F_BNPART, groups=groups)
    rows = N * int(np.prod(othw))
    nrows = ops.conv_bnpart_rows(d)
    part = torch.zeros(nrows, 2, Co, device=DEV)
    z = torch.empty(N, *othw, Co, device=DEV)
    ops.conv_fwd(d, cl(x), w_oki(w), z, bnpart=part)
    rm = torch.zeros(Co, device=DEV); rv = torch.ones(Co, device=DEV)
    stat = ops.bn_finalize(part, nrows // groups, groups, Co, rows // groups, gamma.to(DEV), beta.to(DEV), spec.BN_EPS, spec.BN_MOMENTUM, rm, rv)
    yg = torch.empty_like(z)
    ops.bn_apply(z, Co, stat, Co, rows, groups, yg, Co, True)
    close(uncl(yg), y, what="bn+relu fwd")
    close(rm.cpu(), P["u.bn.running_mean"], atol=1e-6, what="running_mean")
    close(rv.cpu(), P["u.bn.running_var"], atol=1e-6, what="running_var")
    dz = torch.empty_like(z); dgam = torch.zeros(Co, device=DEV); dbet = torch.zeros(Co, device=DEV)
    ops.bn_bwd(cl(dy), Co, z, Co, stat, Co, rows, groups, True, dz, Co, dgam, dbet)
    close(dgam.cpu(), P["u.bn.weight"].grad, what="dgamma")
    close(dbet.cpu(), P["u.bn.bias"].grad, what="dbeta")
    dx = torch.empty(N, *thw, Ci, device=DEV)
    for dd in desc.transposed_classes(N, othw, Co, Co, thw, Ci, Ci, k, s, pf):
        ops.conv_fwd(dd, dz, w_iko(w), dx)
    close(uncl(dx), xs.grad, what="dx through bn")
    ev = ops.bn_eval_stat(gamma.to(DEV), beta.to(DEV), rm, rv, spec.BN_EPS)
    ye = torch.empty_like(z)
    ops.bn_apply(z, Co, ev, Co, rows, 1, ye, Co, True)
    ref_e = torch.relu(F.batch_norm(uncl(z), rm.cpu(), rv.cpu(), gamma, beta, False, 0.0, spec.BN_EPS))
    close(uncl(ye), ref_e, what="bn eval")


@pytest.mark.parametrize("rows_g,C,ld,npg", [(6272, 512, 512, 98), (12544, 208, 832, 196), (1000, 40, 48, 7), (300, 64, 64, 256)])
def test_bn_finalize_folded_into_apply_matches_the_two_launches(rows_g, C, ld, npg):
    """pc_bn_finalize_apply (round 5): every block of the apply kernel reduces the partial rows of its own 64 channels.  Against
    pc_bn_finalize + pc_bn_apply on the same partial rows: statistics and running statistics to the last bit or one ulp (both sum in fp64,
    in different fixed orders), outputs to one ulp of the scale, channel slices of wider tensors untouched outside, repeated runs identical."""
    g = torch.Generator().manual_seed(rows_g + C)
    groups, rows = 2, 2 * rows_g
    z = torch.randn(rows, ld, generator=g).to(DEV)
    zs = z[:, 8:8 + C] if ld > C else z
    # partial rows as a conv epilogue leaves them: sums / sums of squares of row blocks of each group
    per = -(-rows_g // npg)
    part = torch.zeros(groups * npg, 2, C, device=DEV)
    for gi in range(groups):
        for q in range(npg):
            blk = zs[gi * rows_g + q * per: gi * rows_g + min(rows_g, (q + 1) * per)].double()
            if blk.numel():
                part[gi * npg + q, 0] = blk.sum(0).float(); part[gi * npg + q, 1] = (blk * blk).sum(0).float()
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV); beta = torch.randn(C, generator=g).to(DEV)
    rm0 = torch.randn(C, generator=g).to(DEV); rv0 = (torch.rand(C, generator=g) + 0.5).to(DEV)
    rm, rv = rm0.clone(), rv0.clone()
    st = ops.bn_finalize(part, npg, groups, C, rows_g, gamma, beta, spec.BN_EPS, spec.BN_MOMENTUM, rm, rv)
    y = torch.full((rows, ld), 3.5, device=DEV)
    ops.bn_apply(zs, ld, st, C, rows, groups, y[:, 8:] if ld > C else y, ld, True)
    outs = []
    for _ in range(2):
        rm2, rv2 = rm0.clone(), rv0.clone()
        y2 = torch.full((rows, ld), 3.5, device=DEV)
        st2 = ops.bn_finalize_apply(part, npg, groups, C, rows_g, gamma, beta, spec.BN_EPS, spec.BN_MOMENTUM, rm2, rv2, zs, ld, rows, y2[:, 8:] if ld > C else y2, ld, True)
        outs.append((st2.clone(), rm2, rv2, y2))
    st2, rm2, rv2, y2 = outs[0]
    assert torch.allclose(st2, st, rtol=3e-7, atol=1e-9) and torch.allclose(rm2, rm, rtol=3e-7, atol=1e-9) and torch.allclose(rv2, rv, rtol=3e-7, atol=1e-9)
    assert (y2 - y).abs().max().item() <= 4e-7 * max(1.0, y.abs().max().item())
    if ld > C:
        assert torch.all(y2[:, :8] == 3.5) and torch.all(y2[:, 8 + C:] == 3.5)
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    # the mean really is the mean of the rows
    assert torch.allclose(st2[:, 0].double().cpu(), zs.double().view(groups, rows_g, C).mean(1).cpu(), atol=2e-6)
    assert capi.lib().pc_bn_finalize_apply_ok(257, 64) == 0 and capi.lib().pc_bn_finalize_apply_ok(256, 64) == 1


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("rows_g,C", [(6272, 512), (12544, 112), (20000, 64)])
def test_bn_backward_both_forms_vs_autograd(rows_g, C, fused):
    """pc_bn_bwd in its three-launch form (default) and with the finalize folded into the apply kernel (bit 1 of its `relu` argument; layers of
    <= 16 384 rows per group) against torch autograd of BatchNorm(train) + ReLU in float64, two batch groups."""
    fused = fused == "1"
    g = torch.Generator().manual_seed(C)
    groups, rows = 2, 2 * rows_g
    z = torch.randn(rows, C, generator=g, dtype=torch.float64) * 2 + 0.3
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    zr = z.clone().requires_grad_(True)
    ys = [torch.relu(F.batch_norm(zr[i * rows_g:(i + 1) * rows_g], None, None, gamma, beta, True, 0.0, spec.BN_EPS)) for i in range(groups)]
    (torch.cat(ys) * dy).sum().backward()
    zg = z.float().to(DEV)
    stat = torch.empty(groups, 4, C, device=DEV)
    for i in range(groups):
        blk = z[i * rows_g:(i + 1) * rows_g]
        mean, var = blk.mean(0), blk.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + spec.BN_EPS)
        stat[i, 0], stat[i, 1] = mean.float().to(DEV), inv.float().to(DEV)
        stat[i, 2], stat[i, 3] = (gamma.detach() * inv).float().to(DEV), (beta.detach() - mean * gamma.detach() * inv).float().to(DEV)
    dz = torch.empty(rows, C, device=DEV); dgam = torch.zeros(C, device=DEV); dbet = torch.zeros(C, device=DEV)
    ops.bn_bwd(dy.float().to(DEV), C, zg, C, stat, C, rows, groups, True, dz, C, dgam, dbet, fused=fused)
    close(dz.cpu(), zr.grad, rtol=2e-4, atol=2e-5, what="dz")
    close(dgam.cpu(), gamma.grad, rtol=2e-4, atol=2e-3, what="dgamma")
    close(dbet.cpu(), beta.grad, rtol=2e-4, atol=2e-3, what="dbeta")
    dg2 = dgam.clone(); db2 = dbet.clone()
    ops.bn_bwd(dy.float().to(DEV), C, zg, C, stat, C, rows, groups, True, dz, C, dg2, db2, accum=True, fused=fused)
    assert torch.allclose(dg2, 2 * dgam, rtol=1e-6, atol=1e-6) and torch.allclose(db2, 2 * dbet, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("npg,C", [(6272, 64), (1568, 192), (513, 40), (100, 64)])
def test_bn_finalize_two_stage_matches_one_stage(npg, C):
    """Layers with thousands of BatchNorm partial rows (the stem: 6272 per group) are finalized in two stages (32 slices per 16 channels
    and group, then their double-precision rows); both forms sum in a fixed order in fp64, so statistics and running statistics agree to
    the last float bit or one ulp, and repeated runs are bit-identical."""
    g = torch.Generator().manual_seed(npg + C)
    groups, count = 2, 12345
    part = torch.rand(groups * npg, 2, C, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV); beta = torch.randn(C, generator=g).to(DEV)
    outs = []
    for two in (False, True, True):
        rm = torch.zeros(C, device=DEV); rv = torch.ones(C, device=DEV)
        st = ops.bn_finalize(part, npg, groups, C, count, gamma, beta, spec.BN_EPS, spec.BN_MOMENTUM, rm, rv, two_stage=two)
        outs.append((st.cpu(), rm.cpu(), rv.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a, b, rtol=3e-7, atol=1e-9), (a - b).abs().max()
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
    ref_mean = part.view(groups, npg, 2, C)[:, :, 0].double().sum(1).cpu() / count
    assert torch.allclose(outs[1][0][:, 0].double(), ref_mean, rtol=1e-6)


@pytest.mark.parametrize("tag", ["p133", "p333s2", "p333s1", "p133odd"])
def test_maxpool_golden(golden_dir, tag):
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    x = torch.from_numpy(G[tag + "_x"]); k = [int(v) for v in G[tag + "_k"]]; s = [int(v) for v in G[tag + "_s"]]
    N, C_, T, H, W = x.shape
    othw = [spec.same_out((T, H, W)[i], k[i], s[i]) for i in range(3)]
    padf = [spec.same_pad((T, H, W)[i], k[i], s[i])[0] for i in range(3)]
    d = desc.pool(N, (T, H, W), C_, C_, othw, C_, k, s, padf)
    y = torch.empty(N, *othw, C_, device=DEV); am = torch.empty(N, *othw, C_, device=DEV, dtype=torch.uint8)
    ops.maxpool_fwd(d, cl(x), y, am)
    close(uncl(y), torch.from_numpy(G[tag + "_y"]), atol=0, what="pool fwd")
    dx = torch.empty(N, T, H, W, C_, device=DEV)
    ops.maxpool_bwd(d, cl(torch.from_numpy(G[tag + "_dy"])), am, dx)
    close(uncl(dx), torch.from_numpy(G[tag + "_dx"]), atol=1e-6, what="pool bwd")


def test_elementwise_helpers():
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 8, 6, 10, generator=g, dtype=torch.float64)
    a = ops.to_ndhwc(x.to(DEV), 4)
    assert torch.equal(a[..., :3].cpu(), x.float().permute(0, 2, 3, 4, 1)) and (a[..., 3] == 0).all()
    b = ops.to_ndhwc(x.to(DEV), 4, flipw=True)
    assert torch.equal(b[..., :3].cpu(), x.float().flip(4).permute(0, 2, 3, 4, 1))
    assert torch.equal(ops.to_ncdhw(a, 3).cpu(), x.float())
    t = torch.randn(2, 5, 1, 4, 8, generator=g).to(DEV); sc = torch.rand(2, 8, generator=g).to(DEV)
    y = torch.empty_like(t)
    ops.channel_scale(t, 8, sc, 2, 5 * 4, 8, y, 8)
    assert torch.allclose(y, t * sc.view(2, 1, 1, 1, 8))
    src = torch.randn(3, 37, 50, generator=g).to(DEV); dst = torch.zeros(3, 50, 37, device=DEV)
    ops.transpose_batched(src, 3, 37, 50, 37 * 50, 50, dst, 50 * 37, 37)
    assert torch.equal(dst, src.transpose(1, 2))
    ops.transpose_batched(src, 3, 37, 50, 37 * 50, 50, dst, 50 * 37, 37, accum=True)
    assert torch.allclose(dst, 2 * src.transpose(1, 2))
    # activation backward + bias grad
    yv = torch.rand(70, 24, generator=g).to(DEV); dyv = torch.randn(70, 24, generator=g).to(DEV)
    for act, f in ((capi.ACT_RELU, lambda v: (v > 0.5).float()), (capi.ACT_SIGMOID, lambda v: v * (1 - v)), (capi.ACT_NONE, lambda v: torch.ones_like(v))):
        yy = (yv - 0.5).clamp(min=0) if act == capi.ACT_RELU else yv
        dz = torch.empty_like(dyv); db = torch.zeros(24, device=DEV)
        ops.act_bwd(dyv, 24, yy, 24, act, 24, 70, dz, 24, db)
        ref = dyv * (f(yv) if act != capi.ACT_RELU else (yy > 0).float())
        assert torch.allclose(dz, ref, atol=1e-6) and torch.allclose(db, ref.sum(0), atol=1e-4)
    p = torch.randn(1000, generator=g).to(DEV); gr = torch.randn(1000, generator=g).to(DEV)
    m = torch.zeros(1000, device=DEV); v = torch.zeros(1000, device=DEV)
    pr = p.clone().cpu().requires_grad_(True); opt = torch.optim.Adam([pr], lr=1e-3, eps=1e-6)
    for stp in (1, 2, 3):
        pr.grad = gr.cpu().clone(); opt.step()
        ops.adam_step(p, gr, m, v, 1e-3, stp)
    assert torch.allclose(p.cpu(), pr.detach(), atol=1e-6)


def _em_oracle(x, W, bu, ba, dout, npos, B, C_, dtype):
    x = x.detach().to(dtype).requires_grad_(True); W = W.detach().to(dtype).requires_grad_(True)
    bu = bu.detach().to(dtype).requires_grad_(True); ba = ba.detach().to(dtype).requires_grad_(True)
    v = ocaps.votes(x[:, :B * 16].reshape(npos, B, 16), W)
    mu, a_out = ocaps.em_routing(v, x[:, B * 16:].reshape(npos, B, 1), bu, ba)
    out = torch.cat([mu.reshape(npos, C_ * 16), a_out.reshape(npos, C_)], 1)
    out.backward(dout.to(dtype))
    return [t.detach().double() for t in (out[:, :C_ * 16], out[:, C_ * 16:], x.grad, W.grad[0], bu.grad, ba.grad)]


@pytest.mark.parametrize("C_,npos,pscale", [(24, 37, 1.0), (21, 20, 1.0), (24, 64, 0.3)])
def test_em_routing_fwd_bwd(C_, npos, pscale):
    """EM routing is ill-conditioned in fp32 (SURVEY finding 4): judge the HIP kernel against an
    fp64 run of the oracle, allowing a small multiple of the fp32 oracle's own distance to it."""
    g = torch.Generator().manual_seed(10)
    B = 32
    x = torch.cat([torch.randn(npos, B * 16, generator=g) * pscale, torch.rand(npos, B, generator=g)], 1)
    W = torch.randn(1, B, C_, 4, 4, generator=g) * 0.5
    bu = torch.randn(C_, 16, generator=g); ba = torch.randn(C_, generator=g)
    dout = torch.randn(npos, C_ * 17, generator=g)
    ref64 = _em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float64)
    ref32 = _em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float32)
    xg = x.to(DEV); Wg = W[0].contiguous().to(DEV)
    # the training path: the forward leaves its routing state for the backward; without it the backward recomputes the forward --
    # same gradients either way, up to the rounding of the one-wave / four-wave reductions
    state = ops.em_state(npos, DEV)
    og = ops.em_fwd(xg, Wg, bu.to(DEV), ba.to(DEV), npos, B, C_, state=state)
    assert torch.equal(og, ops.em_fwd(xg, Wg, bu.to(DEV), ba.to(DEV), npos, B, C_))
    dW = torch.zeros(B, C_, 4, 4, device=DEV); dbu = torch.zeros(C_, 16, device=DEV); dba = torch.zeros(C_, device=DEV)
    dx = ops.em_bwd(xg, Wg, bu.to(DEV), ba.to(DEV), dout.to(DEV), npos, B, C_, dW, dbu, dba, state=state)
    dW2 = torch.zeros_like(dW); dbu2 = torch.zeros_like(dbu); dba2 = torch.zeros_like(dba)
    dx2 = ops.em_bwd(xg, Wg, bu.to(DEV), ba.to(DEV), dout.to(DEV), npos, B, C_, dW2, dbu2, dba2)
    for a_, b_ in ((dx, dx2), (dW, dW2), (dbu, dbu2), (dba, dba2)):
        assert ((a_ - b_).norm() / (b_.norm() + 1e-30)).item() < 1e-4
    got = [og[:, :C_ * 16], og[:, C_ * 16:], dx, dW, dbu, dba]
    names = ["mu", "a_out", "dx", "dW", "dbeta_u", "dbeta_a"]
    report = []
    for n, gt, r32, r64 in zip(names, got, ref32, ref64):
        scale = r64.abs().max().item() + 1e-30
        e_gpu = (gt.detach().cpu().double() - r64).abs().max().item() / scale
        e_cpu = (r32 - r64).abs().max().item() / scale
        report.append((n, e_gpu, e_cpu))
    bad = [(n, eg, ec) for n, eg, ec in report if eg > max(5 * ec, 2e-5)]
    assert not bad, "rel err (gpu vs fp64, cpu-fp32 vs fp64): %s" % report


@pytest.mark.parametrize("seeds", ["act", "both"])
def test_em_routing_bwd_at_reference_init_scale(seeds):
    """The reference's own initialisation (PrimaryCaps weights N(0, 0.1), ConvCaps.weights randn) gives poses of std ~13 and
    saturated activations: assignments become one-hot, single input capsules own a class (v - mu -> 0, 1 / sigma^2 large) and
    the gradient that reaches the poses through a_out -> cost -> sigma^2 (the spread-loss path, which dominates there) only
    comes out right if d sigma^2 / d mu is taken with the forward's own rounded differences (csrc/caps.hip em_forward `sd`).
    With the gradient seeded on the activations only, that path is not hidden under the pose-seed path; rel-L2 per output
    against an fp64 run of the oracle, next to the fp32 oracle's own distance."""
    g = torch.Generator().manual_seed(12)
    B, C_, npos = 32, 24, 96
    x = torch.cat([torch.randn(npos, B * 16, generator=g) * 13.0, torch.sigmoid(torch.randn(npos, B, generator=g) * 13.0)], 1)
    W = torch.randn(1, B, C_, 4, 4, generator=g)
    bu = torch.randn(C_, 16, generator=g); ba = torch.randn(C_, generator=g)
    dout = torch.randn(npos, C_ * 17, generator=g)
    if seeds == "act":
        dout[:, :C_ * 16] = 0
    ref64 = _em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float64)
    ref32 = _em_oracle(x, W, bu, ba, dout, npos, B, C_, torch.float32)
    xg = x.to(DEV); Wg = W[0].contiguous().to(DEV)
    og = ops.em_fwd(xg, Wg, bu.to(DEV), ba.to(DEV), npos, B, C_)
    dW = torch.zeros(B, C_, 4, 4, device=DEV); dbu = torch.zeros(C_, 16, device=DEV); dba = torch.zeros(C_, device=DEV)
    dx = ops.em_bwd(xg, Wg, bu.to(DEV), ba.to(DEV), dout.to(DEV), npos, B, C_, dW, dbu, dba)
    split = lambda t: [t[:, :B * 16], t[:, B * 16:]]
    got = [og[:, :C_ * 16], og[:, C_ * 16:]] + split(dx) + [dW, dbu, dba]
    r32 = ref32[:2] + split(ref32[2]) + ref32[3:]; r64 = ref64[:2] + split(ref64[2]) + ref64[3:]
    report = []
    for n, gt, a32, a64 in zip(["mu", "a_out", "dpose", "da_in", "dW", "dbeta_u", "dbeta_a"], got, r32, r64):
        den = a64.norm().item() + 1e-300
        report.append((n, (gt.detach().cpu().double() - a64).norm().item() / den, (a32 - a64).norm().item() / den))
    bad = [(n, eg, ec) for n, eg, ec in report if eg > max(2 * ec, 2e-4)]
    assert not bad, "rel-L2 (hip vs fp64, fp32 oracle vs fp64): %s" % report


def test_em_routing_golden(golden_dir):
    """Reference ConvCaps output (fp32) on a full-size capsule layer.  The reference's a_out carries a few
    1e-3 of its own rounding noise through the sum-then-square stdv (SURVEY finding 4), so the HIP result is
    anchored on an fp64 run of the oracle: it must be no further from it than the reference itself is."""
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    pc = torch.from_numpy(G["capF_pc"]); b, h, w, _ = pc.shape
    out = ops.em_fwd(pc.reshape(-1, 32 * 17).to(DEV), torch.from_numpy(G["capF_W"])[0].contiguous().to(DEV),
                     torch.from_numpy(G["capF_beta_u"]).to(DEV), torch.from_numpy(G["capF_beta_a"]).to(DEV), b * h * w, 32, 24)
    out = out.reshape(b, h, w, -1).cpu().double()
    ref = torch.from_numpy(G["capF_out"]).double()
    x64 = pc.double().reshape(-1, 32 * 17)
    v = ocaps.votes(x64[:, :512].reshape(-1, 32, 16), torch.from_numpy(G["capF_W"]).double())
    mu, a = ocaps.em_routing(v, x64[:, 512:].reshape(-1, 32, 1), torch.from_numpy(G["capF_beta_u"]).double(), torch.from_numpy(G["capF_beta_a"]).double())
    o64 = torch.cat([mu.reshape(-1, 24 * 16), a.reshape(-1, 24)], 1).reshape(b, h, w, -1)
    e_ref = (ref - o64).abs().max().item(); e_gpu = (out - o64).abs().max().item()
    assert e_gpu <= max(1.5 * e_ref, 5e-5), (e_gpu, e_ref)
    assert (out - ref).abs().max().item() <= 2.5 * max(e_ref, 5e-5)         # and within the reference's own noise band of it
    close(out[..., :384], ref[..., :384], rtol=2e-4, what="poses vs reference")


def test_class_mask_and_tapsum():
    g = torch.Generator().manual_seed(11)
    Bn, npos, C_ = 4, 25, 24
    caps = torch.rand(Bn, npos, C_ * 17, generator=g)
    cls = torch.tensor([3., 7., 0., 23.]); lab = torch.tensor([1, 0, 1, 0], dtype=torch.int32)
    for mode in (0, 1, 2):
        pred, mask, masked = ops.class_mask_fwd(caps.to(DEV), Bn, npos, C_, cls.to(DEV), lab.to(DEV), mode)
        ap = caps[..., C_ * 16:].mean(1)
        m = ocaps.class_mask(ap, cls.view(-1, 1), lab, 1 if mode == 0 else 99, 11, mode != 2)
        close(pred, ap, atol=1e-6, what="actor_prediction")
        assert torch.equal(mask.cpu(), m)
        ref = (caps[..., :C_ * 16].view(Bn, npos, C_, 16) * m.view(Bn, 1, C_, 1)).reshape(Bn, npos, -1)
        close(masked, ref, atol=0, what="masked poses")
        dm = torch.randn(Bn, npos, C_ * 16, generator=g); dp = torch.randn(Bn, C_, generator=g)
        dc = ops.class_mask_bwd(dm.to(DEV), dp.to(DEV), mask, Bn, npos, C_).cpu()
        close(dc[..., :C_ * 16], (dm.view(Bn, npos, C_, 16) * m.view(Bn, 1, C_, 1)).reshape(Bn, npos, -1), atol=0)
        close(dc[..., C_ * 16:], (dp / npos).view(Bn, 1, C_).expand(Bn, npos, C_), atol=1e-7)
    # smooth = 27-tap projection (conv kernel, Co=27->32) + tap sum
    N, T, H, W = 2, 4, 9, 10
    x = torch.randn(N, 128, T, H, W, generator=g, requires_grad=True)
    w = (torch.randn(128, 1, 3, 3, 3, generator=g) * 0.05).requires_grad_(True); b = torch.randn(1, generator=g)
    y = F.conv_transpose3d(x, w, b, padding=1)
    dy = torch.randn(y.shape, generator=g); y.backward(dy)
    wproj = torch.zeros(32, 1, 128); wproj[:27, 0] = w.detach().reshape(128, 27).t()
    proj = torch.empty(N, T, H, W, 32, device=DEV)
    ops.conv_fwd(desc.conv_fwd(N, (T, H, W), 128, 128, 32, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (T, H, W)), cl(x), wproj.to(DEV), proj)
    out = ops.tapsum_fwd(proj, b.to(DEV))
    close(out.cpu(), y[:, 0], what="smooth fwd")
    dproj = ops.tapsum_bwd(dy[:, 0].contiguous().to(DEV))
    dxg = torch.empty(N, T, H, W, 128, device=DEV)
    wT = torch.zeros(128, 1, 32); wT[:, 0, :27] = w.detach().reshape(128, 27)
    ops.conv_fwd(desc.conv_fwd(N, (T, H, W), 32, 32, 128, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (T, H, W)), dproj, wT.to(DEV), dxg)
    close(uncl(dxg), x.grad, what="smooth dgrad")


LOSS_CASES = [
    dict(bv=True, n_frames=5, wt_ramp=0.3, wt_cons=0.1),
    dict(bv=True, n_frames=3, predict_maps=True, wt_ramp=0.0074),
    dict(gv=True, wt_cons=0.5),
    dict(gv=True, lower=0.2, upper=0.85),
    dict(bv=True, gv=True, n_frames=5, wt_ramp=0.6, bv_wt=0.3, gv_wt=0.7),
    dict(),
    dict(bv=True, gv=True, jhmdb=True, n_frames=3, wt_ramp=0.5),
]


@pytest.mark.parametrize("kw", LOSS_CASES)
@pytest.mark.parametrize("hw", [24, 224])
def test_consistency_loss(kw, hw):
    from oracle.step import default_args
    g = torch.Generator().manual_seed(12)
    B = 4
    O = (torch.randn(B, 1, 8, hw, hw, generator=g) * 2).requires_grad_(True)
    Fl = (O.detach().flip(4) * 0.8 + torch.randn(B, 1, 8, hw, hw, generator=g)).requires_grad_(True)
    seg = (torch.rand(B, 1, 8, hw, hw, generator=g) < 0.3).float()
    lab = torch.tensor([1, 0, 1, 0], dtype=torch.int32)
    a = default_args(bv=kw.get("bv", False), gv=kw.get("gv", False), n_frames=kw.get("n_frames", 3),
                     predict_maps=kw.get("predict_maps", False), lower_thresh=kw.get("lower"), upper_thresh=kw.get("upper"),
                     bv_wt=kw.get("bv_wt", 0.5), gv_wt=kw.get("gv_wt", 0.5))
    ramp = kw.get("wt_ramp", 0.0); wt_cons = kw.get("wt_cons", 1.0); wt_loc = 0.7
    idx = torch.where(lab == 1)[0]
    loc = olosses.bce_logits(O[idx], seg[idx]) + olosses.dice_loss(O[idx], seg[idx])
    fp = torch.flip(Fl, [4])
    l2 = olosses.weighted_mse(fp, O, torch.ones_like(O))
    c1 = c2 = None
    if a.bv:
        vc = olosses.var_mask(O, torch.flip(fp, [2]), a.n_frames, a.predict_maps).float()
        va = olosses.var_mask(torch.flip(O, [2]), fp, a.n_frames, a.predict_maps).float()
        c1 = ramp * (olosses.weighted_mse(fp, O, vc) + olosses.weighted_mse(fp, O, torch.flip(va, [2]))) + (1 - ramp) * l2
    if a.gv:
        c2 = olosses.weighted_mse(fp, O, olosses.grad_mask(O, a.lower_thresh, a.upper_thresh).float())
    if kw.get("jhmdb"):
        cons = c2 if a.gv else (c1 if a.bv else l2)
    elif a.bv and a.gv:
        cons = a.bv_wt * c1 + a.gv_wt * c2
    else:
        cons = c2 if a.gv else (c1 if a.bv else l2)
    (wt_loc * loc + wt_cons * cons).backward()
    d = ops.loss_desc(B, 8, hw, hw, bv=a.bv, gv=a.gv, n_frames=a.n_frames, predict_maps=a.predict_maps, jhmdb=kw.get("jhmdb", False),
                      lower=a.lower_thresh, upper=a.upper_thresh, bv_wt=a.bv_wt, gv_wt=a.gv_wt, wt_loc=wt_loc, wt_cons=wt_cons, wt_ramp=ramp)
    scal, dO, dF, mb, mg = ops.consistency_loss(d, O.detach().to(DEV), Fl.detach().to(DEV), seg.to(DEV), lab.to(DEV), want_masks=True)
    scal = scal.cpu()
    assert abs(scal[0].item() - loc.item()) <= 1e-4, ("loc", scal[0].item(), loc.item())        # north_star: loss scalars 1e-4
    assert abs(scal[1].item() - cons.item()) <= 1e-4 * max(1.0, abs(cons.item())), ("cons", scal[1].item(), cons.item())
    close(dO, O.grad, rtol=2e-4, what="d_output")
    close(dF, Fl.grad, rtol=2e-4, what="d_flip_op")
    if a.bv:
        close(mb, vc, atol=2e-5, what="bv mask")
    if a.gv:
        close(mg[:, 0], olosses.grad_mask(O, a.lower_thresh, a.upper_thresh).float(), atol=2e-5, what="gv mask")


def test_masks_standalone_golden(golden_dir):
    """utils.helpers drop-ins vs the reference's own outputs (tests/golden/masks.npz)."""
    M = np.load(os.path.join(golden_dir, "masks.npz"))
    g = np.random.default_rng(23)
    pred = g.normal(0, 2, (2, 1, 8, 224, 224)).astype(np.float32)
    flip = (pred[:, :, ::-1] * 0.7 + g.normal(0, 1, pred.shape)).astype(np.float32)
    P, Fp = torch.from_numpy(pred).to(DEV), torch.from_numpy(flip).to(DEV)
    for nf in (3, 5):
        for sig in (False, True):
            tag = "var%d%s" % (nf, "s" if sig else "")
            m = ops.var_mask(P, Fp, nf, sig).cpu().numpy()
            assert np.abs(m[..., ::7, ::7] - M[tag + "_sample"]).max() <= 2e-5, tag
            assert np.abs(m.astype(np.float64).sum(axis=(-1, -2)) - M[tag + "_sum"]).max() <= 0.05, tag
    for tag, lo, up in (("grad", None, None), ("grad_thr", 0.2, 0.85)):
        m = ops.grad_mask(P, lo, up).cpu().numpy()
        assert np.abs(m[..., ::7, ::7] - M[tag + "_sample"]).max() <= 2e-5, tag


def test_spread_loss_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    x = torch.from_numpy(G["spread_x"]).to(DEV); t = torch.from_numpy(G["spread_t"]).to(DEV)
    lab = torch.ones(x.shape[0], dtype=torch.int32, device=DEV)
    dx = torch.zeros_like(x)
    out = ops.spread_loss(x, t.view(-1), lab, 0.2, 1.0, dx).cpu()
    assert abs(out[0].item() - float(G["spread_loss"])) < 1e-6 and abs(out[1].item() - float(G["spread_abs"])) < 1e-5
    close(dx, torch.from_numpy(G["spread_dx"]), atol=1e-7, what="spread dx")
    # unlabeled rows are ignored
    lab2 = lab.clone(); lab2[1] = 0
    xr = torch.from_numpy(G["spread_x"]).requires_grad_(True); keep = [0, 2, 3, 4]
    l, _ = olosses.spread_loss(xr[keep], torch.from_numpy(G["spread_t"])[keep]); l.backward()
    dx2 = torch.zeros_like(x)
    out2 = ops.spread_loss(x, t.view(-1), lab2, 0.2, 2.0, dx2).cpu()
    assert abs(out2[0].item() - l.item()) < 1e-6
    close(dx2, 2.0 * xr.grad, atol=1e-7)


def test_errors_are_loud():
    with pytest.raises(RuntimeError):
        ops.conv_fwd(desc.conv_fwd(1, (1, 4, 4), 3, 3, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 4, 4)),
                     torch.zeros(1, 1, 4, 4, 3, device=DEV), torch.zeros(8, 1, 3, device=DEV), torch.zeros(1, 1, 4, 4, 8, device=DEV))
    with pytest.raises(RuntimeError):
        ops.conv_fwd(desc.conv_fwd(1, (1, 4, 4), 4, 4, 8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 4, 4)),
                     torch.zeros(1, 1, 4, 4, 4), torch.zeros(8, 1, 4), torch.zeros(1, 1, 4, 4, 8))   # CPU tensors


def test_collapsed_tail_kernels_vs_torch():
    """pc_tail_combine / pc_tail_colsum / pc_tail_grads against einsum, and the algebraic identity
    smooth(drop(upsample4(x))) == tapsum(convT(x, Wc[n]) + bc[n]) on a small case (capsules_ucf101.py:504-509)."""
    import ctypes as C
    g = torch.Generator().manual_seed(13)
    N, Ci, Co, taps, J = 2, 8, 12, 27, 27
    W4 = (torch.randn(Ci, Co, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b4 = torch.randn(Co, generator=g).requires_grad_(True)
    Wp = (torch.randn(Co, 1, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    bp = torch.randn(1, generator=g).requires_grad_(True)
    cs = (torch.rand(N, Co, generator=g) < 0.5).float() * 2
    x = torch.randn(N, Ci, 2, 3, 4, generator=g)
    u = F.conv_transpose3d(x, W4, b4, stride=2, padding=1, output_padding=1) * cs.view(N, Co, 1, 1, 1)
    out = F.conv_transpose3d(u, Wp, bp, padding=1)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    dev = lambda t: t.detach().contiguous().to(DEV)
    Wt = torch.empty(N, Ci, taps, 32, device=DEV); Wf = torch.empty(N, 32, taps, Ci, device=DEV); bc = torch.empty(N, 32, device=DEV)
    W4g, b4g, csg, Wpg, bpg = dev(W4), dev(b4), dev(cs), dev(Wp), dev(bp)      # keep the device copies alive across the raw-pointer calls
    capi.call("pc_tail_combine", ops.ptr(W4g), ops.ptr(b4g), ops.ptr(csg), ops.ptr(Wpg), N, Ci, Co, taps, J,
              ops.ptr(Wt), ops.ptr(Wf), ops.ptr(bc), ops.stream())
    Wc = torch.einsum("iot,no,oj->nitj", W4.detach().reshape(Ci, Co, taps), cs, Wp.detach().reshape(Co, J))
    close(Wt[..., :27], Wc, what="Wt"); close(Wf[:, :27].permute(0, 3, 2, 1), Wc, what="Wf")
    assert (Wt[..., 27:] == 0).all()
    close(bc[:, :27], torch.einsum("o,no,oj->nj", b4.detach(), cs, Wp.detach().reshape(Co, J)), what="bc")
    # forward identity through the conv kernel with per-sample weights + tap sum
    othw = tuple(out.shape[2:])
    proj = torch.zeros(N, *othw, 32, device=DEV)
    Cp = 8
    for dd in desc.transposed_classes(N, (2, 3, 4), Cp, Cp, othw, 32, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), flags=capi.F_BIAS, groups=N):
        dd["wgstride"] = 32 * taps * Ci; dd["bgstride"] = 32
        ops.conv_fwd(dd, cl(x), Wf, proj, bias=bc)
    o = ops.tapsum_fwd(proj, bpg)
    close(o.cpu(), out[:, 0], what="collapsed forward")
    # backward pieces
    dproj = ops.tapsum_bwd(dev(dout[:, 0]))
    sums = torch.empty(N, 32, device=DEV)
    per_n = int(np.prod(othw))
    capi.call("pc_tail_colsum", ops.ptr(dproj), N, per_n, ops.ptr(sums), ops.stream())
    close(sums, dproj.reshape(N, per_n, 32).sum(1), what="colsum")
    G = torch.zeros(N, Ci, taps, 32, device=DEV)
    xg = cl(x)
    for n in range(N):
        ops.conv_wgrad(desc.wgrad(1, (2, 3, 4), Ci, Ci, othw, 32, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), xg[n], dproj[n], G[n])
    dW4 = torch.zeros(Ci, Co, taps, device=DEV); db4 = torch.zeros(Co, device=DEV); dWp = torch.zeros(Co, J, device=DEV); dbp = torch.zeros(1, device=DEV)
    capi.call("pc_tail_grads", ops.ptr(G), ops.ptr(sums), ops.ptr(W4g), ops.ptr(b4g), ops.ptr(csg), ops.ptr(Wpg), N, Ci, Co, taps, J, 13,
              ops.ptr(dW4), ops.ptr(db4), ops.ptr(dWp), ops.ptr(dbp), 0, ops.stream())
    close(dW4.cpu(), W4.grad.reshape(Ci, Co, taps), what="dW4"); close(db4.cpu(), b4.grad, what="db4")
    close(dWp.cpu(), Wp.grad.reshape(Co, J), what="dWp"); close(dbp.cpu(), bp.grad, what="dbp")
    dx = torch.empty(N, 2, 3, 4, Ci, device=DEV)
    dd = desc.conv_fwd(N, othw, 32, 32, Ci, Ci, (3, 3, 3), (2, 2, 2), (1, 1, 1), (2, 3, 4), groups=N, ldw=32)
    dd["wgstride"] = Ci * taps * 32
    ops.conv_fwd(dd, dproj, Wt, dx)
    xr = x.clone().requires_grad_(True)
    F.conv_transpose3d(F.conv_transpose3d(xr, W4.detach(), b4.detach(), stride=2, padding=1, output_padding=1) * cs.view(N, Co, 1, 1, 1),
                       Wp.detach(), bp.detach(), padding=1).backward(dout)
    close(uncl(dx), xr.grad, what="collapsed dgrad")


def test_merged_tail_vs_torch():
    """upsample4 -> Dropout3d -> smooth as ONE five-tap stride-2 transposed conv with a single output channel
    (csrc/tail6.hip): 125-column GEMMs over the 8 position classes + gather forward; scatter + two GEMMs + the map back onto the combined weights
    backward; output, input gradient and all four parameter gradients against torch (capsules_ucf101.py:504-509)."""
    g = torch.Generator().manual_seed(14)
    N, Ci, Co, taps, J = 3, 8, 12, 27, 27
    I = (2, 3, 5)
    W4 = (torch.randn(Ci, Co, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b4 = torch.randn(Co, generator=g).requires_grad_(True)
    Wp = (torch.randn(Co, 1, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    bp = torch.randn(1, generator=g).requires_grad_(True)
    cs = (torch.rand(N, Co, generator=g) < 0.5).float() * 2
    x = torch.randn(N, Ci, *I, generator=g, requires_grad=True)
    u = F.conv_transpose3d(x, W4, b4, stride=2, padding=1, output_padding=1) * cs.view(N, Co, 1, 1, 1)
    out = F.conv_transpose3d(u, Wp, bp, padding=1)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    dev = lambda t: t.detach().contiguous().to(DEV)
    Wt = torch.empty(N, Ci, taps, 32, device=DEV); Wf = torch.empty(N, 32, taps, Ci, device=DEV); bc = torch.empty(N, 32, device=DEV)
    W4g, b4g, csg, Wpg, bpg = dev(W4), dev(b4), dev(cs), dev(Wp), dev(bp)
    capi.call("pc_tail_combine", ops.ptr(W4g), ops.ptr(b4g), ops.ptr(csg), ops.ptr(Wpg), N, Ci, Co, taps, J,
              ops.ptr(Wt), ops.ptr(Wf), ops.ptr(bc), ops.stream())
    from picons_amd import tail6
    SP = tail6.SP
    W5f = torch.empty(N, 8, SP, Ci, device=DEV); W5t = torch.empty(N, 8, Ci, SP, device=DEV)
    ops.tail6_weights(Wf, N, Ci, W5f, W5t)
    xg = cl(x)
    cols = torch.full((N, *I, SP), 9.0, device=DEV)
    for z, d in tail6.conv_descs(N, I, Ci, Ci):
        ops.conv_fwd(d, xg, W5f.view(-1)[z * SP * Ci:], cols)
    o = torch.empty(N, *[2 * v for v in I], device=DEV)
    ops.tail6_gather(cols, bc, bpg, N, *I, o)
    close(o.cpu(), out[:, 0], what="merged forward")
    # backward
    doutg = dev(dout[:, 0])
    dcols = torch.empty(N, *I, SP, device=DEV)
    ops.tail6_scatter(doutg, N, *I, dcols)
    dx = torch.full((N, *I, Ci), 9.0, device=DEV)
    for z, d in tail6.dgrad_descs(N, I, Ci, Ci, False):
        ops.conv_fwd(d, dcols, W5t.view(-1)[z * Ci * SP:], dx)
    close(uncl(dx), x.grad, what="merged dgrad")
    dW5 = torch.zeros(N, 8, Ci, SP, device=DEV)
    for z, d in tail6.wgrad_descs(N, I, Ci, Ci):
        ops.conv_wgrad(d, xg, dcols, dW5.view(-1)[z * Ci * SP:])
    Gc = torch.empty(N, Ci, taps, 32, device=DEV)
    ops.tail6_wgrad_map(dW5, N, Ci, Gc)
    sums = torch.empty(N, 32, device=DEV)
    ops.tail6_bias_sums(doutg, N, *I, sums)
    dproj = ops.tapsum_bwd(doutg)
    close(sums[:, :27], dproj.reshape(N, -1, 32).sum(1)[:, :27], what="bias sums")
    dW4 = torch.zeros(Ci, Co, taps, device=DEV); db4 = torch.zeros(Co, device=DEV); dWp = torch.zeros(Co, J, device=DEV); dbp = torch.zeros(1, device=DEV)
    capi.call("pc_tail_grads", ops.ptr(Gc), ops.ptr(sums), ops.ptr(W4g), ops.ptr(b4g), ops.ptr(csg), ops.ptr(Wpg), N, Ci, Co, taps, J, 13,
              ops.ptr(dW4), ops.ptr(db4), ops.ptr(dWp), ops.ptr(dbp), 0, ops.stream())
    close(dW4.cpu(), W4.grad.reshape(Ci, Co, taps), what="dW4"); close(db4.cpu(), b4.grad, what="db4")
    close(dWp.cpu(), Wp.grad.reshape(Co, J), what="dWp"); close(dbp.cpu(), bp.grad, what="dbp")


def test_full_correlation_as_gemm_plus_col2im():
    """PrimaryCaps dgrad form: cols = dY x W^T (1x1 conv with Co = taps*Ci) then pc_col2im == conv2d dgrad."""
    g = torch.Generator().manual_seed(14)
    N, Ci, Co, K, Ho = 3, 32, 64, 5, 6
    x = torch.randn(N, Ci, Ho + K - 1, Ho + K - 1, generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, K, K, generator=g) * 0.1
    y = F.conv2d(x, w)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # weights [tap][ci][co]
    wt = w.reshape(Co, Ci, K * K).permute(2, 1, 0).contiguous().to(DEV)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(DEV).view(N, 1, Ho, Ho, Co)
    cols = torch.empty(N, 1, Ho, Ho, K * K * Ci, device=DEV)
    ops.conv_fwd(desc.conv_fwd(N, (1, Ho, Ho), Co, Co, K * K * Ci, K * K * Ci, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, Ho, Ho), ldw=Co), dyg, wt, cols)
    dx = torch.full((N, Ho + K - 1, Ho + K - 1, Ci), 1.5, device=DEV)
    capi.call("pc_col2im", ops.ptr(cols), N, Ho, Ho, K, K, Ci, ops.ptr(dx), Ci, 0, ops.stream())
    close(dx.permute(0, 3, 1, 2).cpu(), x.grad, what="col2im dgrad")
    capi.call("pc_col2im", ops.ptr(cols), N, Ho, Ho, K, K, Ci, ops.ptr(dx), Ci, 1, ops.stream())
    close(dx.permute(0, 3, 1, 2).cpu(), 2 * x.grad, what="col2im accumulate")


def test_wgrad_multi_matches_individual_launches():
    """pc_conv_wgrad_multi: the wgrads of an Inception-module-like group (1x1x1 and 1x3x3 problems of different channel counts, one
    of them a row-segment shape that keeps its own launch, one batched sub-lattice problem like the merged tail's classes) in one
    call against pc_conv_wgrad per problem: same gradients up to the order of the fp32 atomic sums."""
    g = torch.Generator().manual_seed(21)
    N, thw = 4, (1, 28, 28)
    jobs, refs = [], []
    for Cd, Cs, k in [(96, 48, (1, 1, 1)), (72, 16, (1, 3, 3)), (208, 96, (1, 3, 3)), (64, 112, (1, 1, 1)), (40, 24, (1, 3, 3))]:
        pad = tuple(x // 2 for x in k)
        Dt = torch.randn(N, *thw, Cd, generator=g).to(DEV); St = torch.randn(N, *thw, Cs, generator=g).to(DEV)
        d = desc.trim_wgrad(desc.wgrad(N, thw, Cd, Cd, thw, Cs, Cs, k, (1, 1, 1), pad))
        g1 = torch.zeros(Cd, k[0] * k[1] * k[2], Cs, device=DEV); g2 = torch.zeros_like(g1)
        ops.conv_wgrad(d, Dt, St, g1)
        jobs.append((d, Dt, St, g2)); refs.append(g1)
    # a batched sub-lattice problem (nbatch = N, D = positions (1.., 1..) of a larger tensor), as the merged tail's classes are
    Cd, Cs, I = 32, 128, (2, 6, 6)
    X = torch.randn(N, *I, Cd, generator=g).to(DEV); Y = torch.randn(N, *I, Cs, generator=g).to(DEV)
    per = I[0] * I[1] * I[2]
    d = desc.wgrad(1, (1, 5, 5), Cd, Cd, I, Cs, Cs, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d.update(ioff0=[1, 1, 1], Td=I[0], Hd=I[1], Wd=I[2], doff=[1, 1, 1], nbatch=N, dbstride=per * Cd, sbstride=per * Cs, gbstride=Cd * Cs)
    g1 = torch.zeros(N, Cd, 1, Cs, device=DEV); g2 = torch.zeros_like(g1)
    ops.conv_wgrad(d, X, Y, g1)
    jobs.append((d, X, Y, g2)); refs.append(g1)
    ops.conv_wgrad_multi(jobs)
    for (d, _a, _b, got), want in zip(jobs, refs):
        assert want.abs().max().item() > 0
        rel = ((got - want).norm() / want.norm()).item()
        assert rel < 2e-6, (d["Cd"], d["Cs"], rel)
    # and against torch for one of them
    d, Dt, St, got = jobs[2]
    x = St.permute(0, 4, 1, 2, 3).contiguous().requires_grad_(False)
    w = torch.zeros(208, 96, 1, 3, 3, device=DEV, requires_grad=True)
    y = torch.nn.functional.conv3d(x, w, padding=(0, 1, 1))
    y.backward(Dt.permute(0, 4, 1, 2, 3).contiguous())
    want = w.grad.permute(0, 2, 3, 4, 1).reshape(208, 9, 96)
    assert ((got - want).norm() / want.norm()).item() < 1e-5
