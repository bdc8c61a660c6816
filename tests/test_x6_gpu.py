"""fp32 convolutions multiplied on the bf16 matrix cores (csrc/conv_x6.hip: three bf16 terms per operand, six products, fp32 accumulate),
through the C-ABI.  The bar: against an fp64 torch reference the kernel's error is no larger than the native fp32-MFMA
kernel's on the same launch -- anything looser would be narrower arithmetic than the reference's fp32 (/root/reference/models/pytorch_i3d.py:
116-119, models/capsules_ucf101.py:43-49,384,501 run in fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import capi, desc, ops, spec
from tests.test_kernels_gpu import cl, uncl, w_iko, w_oki

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_split_planes_is_exact():
    """h + m + l == x bit for bit (three bf16 terms hold 24 significant bits), zeros stay zeros, the terms are ordered by magnitude.
    Below 2^-110 the lower terms are subnormal fp32 differences, which the vector ALU flushes: there the sum is off by less than 2^-126."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1 << 16, generator=g) * torch.exp(4 * torch.randn(1 << 16, generator=g))
    x[:64] = 0.0
    x[64:128] = torch.tensor([1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 2.0 ** -20, 1 + 2.0 ** -23, 1 - 2.0 ** -24] * 8)
    pl = ops.split_planes(x.to(DEV)).cpu()
    terms = (pl.to(torch.int32) << 16).view(torch.float32)          # a bf16 pattern is the top half of the fp32 with the same value
    s = terms[0].double() + terms[1].double() + terms[2].double()
    bad = (s != x.double()).nonzero().flatten()
    assert bad.numel() == 0, "the three bf16 terms do not sum to the fp32 value at %s: x = %s, terms = %s" % (bad[:4].tolist(), x[bad[:4]].tolist(), terms[:, bad[:4]].tolist())
    assert torch.all(terms[:, :64] == 0)
    tiny = torch.tensor([1.2e-38, -3.0e-37, 2.0 ** -120, 1.5e-36], device=DEV)
    tt = (ops.split_planes(tiny).cpu().to(torch.int32) << 16).view(torch.float32).double().sum(0)
    assert (tt - tiny.cpu().double()).abs().max().item() < 2.0 ** -126
    assert torch.all(terms[1].abs() <= terms[0].abs() * 2.0 ** -8 + 1e-45) and torch.all(terms[2].abs() <= terms[0].abs() * 2.0 ** -16 + 1e-45)


def _rel(a, ref):
    return ((a.double() - ref).abs().sum() / ref.abs().sum()).item()


# (Ci, Co, k, stride, thw, N): every tile of pc_x6_tile and the epilogue variants
X6_CASES = [
    (64, 64, (3, 3, 3), (1, 1, 1), (2, 16, 32), 2),        # 128 x 64
    (128, 128, (1, 1, 1), (1, 1, 1), (4, 64, 64), 8),      # 256 x 128 on 8 waves (131 072 rows: two rounds of one block per CU), one tap
    (64, 240, (1, 3, 3), (1, 1, 1), (1, 64, 64), 16),      # 256 x 128 with a ragged last column tile
    (64, 192, (3, 3, 3), (2, 1, 1), (4, 28, 56), 4),       # temporal stride 2 (two parity classes in the input gradient)
    (96, 32, (1, 3, 3), (1, 1, 1), (1, 28, 28), 4),        # 128 x 32
    (160, 320, (3, 3, 3), (1, 1, 1), (1, 28, 28), 2),      # 128 x 64, 5 channel chunks, one frame (temporal taps trimmed by the tile's tap box)
    (32, 200, (3, 3, 3), (1, 1, 1), (2, 14, 14), 2),       # Co not a multiple of the tile
    (832, 64, (1, 3, 3), (1, 1, 1), (1, 28, 28), 16),      # conv28: K = 7 488
]


@pytest.mark.parametrize("Ci,Co,k,s,thw,N", X6_CASES)
def test_x6_conv_fwd_dgrad_vs_fp64_and_native(Ci, Co, k, s, thw, N):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g))
    x = torch.relu(x)                                        # what the layers see: half the operand is exact zeros
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y64 = F.conv3d(xp, w.double(), None, s)
    othw = tuple(y64.shape[2:]); pf = [p[0] for p in pads]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, s, pf, othw)
    xg, wk = cl(x), w_oki(w)
    nat = ops.conv_fwd(d, xg, wk, torch.empty(N, *othw, Co, device=DEV))
    got = ops.conv_fwd_x6(d, xg, ops.split_planes(wk), torch.empty(N, *othw, Co, device=DEV))
    e_nat, e_x6 = _rel(uncl(nat), y64), _rel(uncl(got), y64)
    assert e_x6 <= 1.05 * e_nat + 1e-9, "forward: bf16-split error %.3e vs native fp32 MFMA %.3e (against fp64)" % (e_x6, e_nat)
    assert e_x6 < 2e-6
    # input gradient: transposed weights, one launch per output-parity class, accumulate flag on the second visit
    dy = torch.randn(y64.shape, generator=g)
    dx64 = torch.autograd.grad(F.conv3d(xp.requires_grad_(True), w.double(), None, s), xp, dy.double())[0]
    dx64 = dx64[:, :, pads[0][0]:pads[0][0] + thw[0], pads[1][0]:pads[1][0] + thw[1], pads[2][0]:pads[2][0] + thw[2]]
    dyg, wt = cl(dy), w_iko(w)
    if Co % 32:
        return                                              # dgrad's K is Co: not a whole number of 32-channel chunks
    wtp = ops.split_planes(wt)
    dn = torch.zeros(N, *thw, Ci, device=DEV); dg = torch.zeros(N, *thw, Ci, device=DEV)
    for dd in desc.transposed_classes(N, othw, Co, Co, thw, Ci, Ci, k, s, pf, ldw=Co):
        ops.conv_fwd(dd, dyg, wt, dn)
        ops.conv_fwd_x6(dd, dyg, wtp, dg)
    e_nat, e_x6 = _rel(uncl(dn), dx64), _rel(uncl(dg), dx64)
    assert e_x6 <= 1.05 * e_nat + 1e-9, "dgrad: bf16-split error %.3e vs native %.3e" % (e_x6, e_nat)


def test_x6_epilogues_bias_relu_cscale_accum_bnpart_groups():
    """Every epilogue of the kernel against the native one on the same launch (they share the code; what differs is the tile, hence the
    BatchNorm partial rows): bias + ReLU from a channel on, Dropout3d scale, accumulate, BN partial sums per batch group."""
    g = torch.Generator().manual_seed(3)
    N, Ci, Co, thw, k = 4, 64, 96, (2, 28, 28), (3, 3, 3)
    x = torch.randn(N, *thw, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, 27, Ci, generator=g) / 40).to(DEV)
    bias = torch.randn(Co, generator=g).to(DEV)
    cs = ((torch.rand(N, Co, generator=g) < 0.5).float() * 2).to(DEV)
    wp = ops.split_planes(w)
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), (1, 1, 1), thw, act=capi.ACT_RELU, flags=capi.F_BIAS | capi.F_CSCALE | capi.F_ACCUM)
    d["act_c0"] = 32
    base = torch.randn(N, *thw, Co, generator=g).to(DEV)
    a = ops.conv_fwd(d, x, w, base.clone(), bias=bias, cscale=cs)
    b = ops.conv_fwd_x6(d, x, wp, base.clone(), bias=bias, cscale=cs)
    assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item()
    # BatchNorm partials, two batch groups: column sums of every group's rows must agree with the output itself
    d2 = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), (1, 1, 1), thw, flags=capi.F_BNPART, groups=2)
    dx6 = dict(d2, flags=d2["flags"] | capi.F_X6)
    rows = ops.conv_bnpart_rows(dx6)
    part = torch.zeros(rows, 2, Co, device=DEV)
    out = ops.conv_fwd_x6(d2, x, wp, torch.empty(N, *thw, Co, device=DEV), bnpart=part)
    per = rows // 2
    for gi in range(2):
        o = out[gi * 2:(gi + 1) * 2].reshape(-1, Co).double()
        s1 = part[gi * per:(gi + 1) * per, 0].double().sum(0)
        s2 = part[gi * per:(gi + 1) * per, 1].double().sum(0)
        assert (s1 - o.sum(0)).abs().max().item() <= 1e-3 * o.abs().sum(0).max().item()
        assert (s2 - (o * o).sum(0)).abs().max().item() <= 1e-4 * (o * o).sum(0).max().item()


def test_x6_grouped_weights_nfast_and_channel_major_output():
    """Per-group weights (wgstride) with n-fastest rows -- the spectral PrimaryCaps GEMM's launch shape (41 groups x 320 rows: 64-row
    tiles) -- and the channel-major store of the merged tail's column GEMMs."""
    g = torch.Generator().manual_seed(7)
    G, n, H, Ci, Co, KY = 5, 16, 28, 64, 160, 9
    OH = H - KY + 1
    x = torch.randn(G * n, 1, H, 1, Ci, generator=g).to(DEV)              # [g * n][T=1][H][W=1][Ci]
    w = (torch.randn(G, Co, KY, Ci, generator=g) / 24).to(DEV)
    d = desc.conv_fwd(G * n, (1, H, 1), Ci, Ci, Co, Co, (1, KY, 1), (1, 1, 1), (0, 0, 0), (1, OH, 1), flags=capi.F_NFAST, groups=G)
    d["wgstride"] = Co * KY * Ci
    a = ops.conv_fwd(d, x, w, torch.empty(G * n, 1, OH, 1, Co, device=DEV))
    b = ops.conv_fwd_x6(d, x, ops.split_planes(w), torch.empty(G * n, 1, OH, 1, Co, device=DEV))
    ref = torch.einsum("gnhkc,gokc->gnho", x.view(G, n, H, Ci).double().unfold(2, KY, 1).permute(0, 1, 2, 4, 3), w.double()).reshape(a.shape)
    assert _rel(b, ref) <= 1.05 * _rel(a, ref) + 1e-9
    # channel-major output
    N, thw, Ci2, Co2 = 2, (2, 14, 28), 128, 128
    x2 = torch.randn(N, *thw, Ci2, generator=g).to(DEV)
    w2 = (torch.randn(Co2, 1, Ci2, generator=g) / 11).to(DEV)
    d2 = desc.conv_fwd(N, thw, Ci2, Ci2, Co2, Co2, (1, 1, 1), (1, 1, 1), (0, 0, 0), thw, flags=capi.F_TOUT)
    a2 = ops.conv_fwd(d2, x2, w2, torch.empty(N, Co2, *thw, device=DEV))
    b2 = ops.conv_fwd_x6(d2, x2, ops.split_planes(w2), torch.empty(N, Co2, *thw, device=DEV))
    assert (a2 - b2).abs().max().item() <= 2e-5 * a2.abs().max().item()


def test_x6_refuses_what_it_cannot_run():
    x = torch.zeros(1, 1, 8, 8, 24, device=DEV)
    w = torch.zeros(32, 1, 24, device=DEV)
    d = desc.conv_fwd(1, (1, 8, 8), 24, 24, 32, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 8, 8))
    with pytest.raises(RuntimeError, match="bf16-split"):
        ops.conv_fwd_x6(d, x, ops.split_planes(w), torch.empty(1, 1, 8, 8, 32, device=DEV))
    with pytest.raises(RuntimeError, match="PC_F_X6"):
        ops.conv_fwd(dict(d, flags=capi.F_X6), x, w, torch.empty(1, 1, 8, 8, 32, device=DEV))


# (Ci, Co, thw, N, stride_t): the row-segment weight gradient's four bf16-split configurations (64- and 32-channel source blocks, 64- and
# 32-position chunks) incl. a row that is not a whole number of chunks (W = 112: 64 + 48) and the temporal stride 2 of Conv3d_2c
WG_CASES = [(64, 64, (2, 8, 112), 2, 1), (64, 192, (4, 6, 56), 2, 2), (96, 128, (2, 5, 28), 2, 1), (128, 256, (1, 28, 28), 4, 1), (192, 64, (2, 6, 56), 2, 1)]


@pytest.mark.parametrize("Ci,Co,thw,N,st", WG_CASES)
def test_x6_row_segment_wgrad_vs_fp64_and_native(Ci, Co, thw, N, st):
    """PC_WG_X6: the 3x3x3 weight gradient with both operands split in registers (csrc/conv.hip wgrad3_x6_kernel), against an fp64 torch
    gradient and against the native fp32-MFMA kernel on the same launch: its error must be no larger (operands as the step has them: a
    gradient of mixed sign and a ReLU output)."""
    g = torch.Generator().manual_seed(21)
    k, s = (3, 3, 3), (st, 1, 1)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g)))
    w = (torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * 27)).double().requires_grad_(True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, s)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = w.grad.reshape(Co, Ci, 27).permute(0, 2, 1)
    othw = tuple(y.shape[2:]); pf = [p[0] for p in pads]
    xg, dyg = cl(x), cl(dy)
    wd = desc.wgrad(N, othw, Co, Co, thw, Ci, Ci, k, s, pf)
    nat = ops.conv_wgrad(wd, dyg, xg, torch.zeros(Co, 27, Ci, device=DEV))
    got = ops.conv_wgrad(dict(wd, flags=capi.WG_X6), dyg, xg, torch.zeros(Co, 27, Ci, device=DEV))
    e_nat, e_x6 = _rel(nat.cpu(), ref), _rel(got.cpu(), ref)
    assert e_x6 <= 1.05 * e_nat + 1e-9, "wgrad: bf16-split error %.3e vs native fp32 MFMA %.3e (against fp64)" % (e_x6, e_nat)
    assert e_x6 < 5e-6


@pytest.mark.parametrize("Co,thw,N", [(64, (8, 56, 56), 2), (64, (6, 224, 224), 1), (24, (5, 18, 112), 1)])
def test_x6_stem_wgrad_vs_fp64_and_native(Co, thw, N):
    """PC_WG_X6 | PC_WG_CS3 on the stem's 7x7x7 stride-2 weight gradient (round 6: csrc/conv.hip wgrad4_x6_kernel; autograd of Conv3d_1a_7x7,
    /root/reference/models/pytorch_i3d.py:221-225): chunks of two 16-position sub-chunks incl. rows that are not whole sub-chunks (W = 28 = 16 + 12),
    the real 112-wide rows, fewer than 64 output channels, odd extents along t / h.  Bar as for every converted kernel: no further from fp64 than
    the native fp32-MFMA kernel (wgrad4_kernel) on the same launch; operands as the step has them (a clip in [0, 1], a gradient of mixed sign)."""
    g = torch.Generator().manual_seed(23)
    k, s = (7, 7, 7), (2, 2, 2)
    x3 = torch.rand(N, 3, *thw, generator=g)
    w = torch.zeros(Co, 3, *k, dtype=torch.float64, requires_grad=True)
    pads = [spec.same_pad(thw[i], k[i], s[i]) for i in range(3)]
    xp = F.pad(x3.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, s)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = w.grad.reshape(Co, 3, 343).permute(0, 2, 1)
    othw = tuple(y.shape[2:]); pf = [p[0] for p in pads]
    x4 = cl(torch.cat([x3, torch.zeros(N, 1, *thw)], 1))
    dyg = cl(dy)
    wd = dict(desc.wgrad(N, othw, Co, Co, thw, 4, 4, k, s, pf), flags=capi.WG_CS3)
    assert capi.lib().pc_wgrad_uses_x6(ops._fill_struct(capi.WgradDesc(), wd)) == 0
    assert capi.lib().pc_wgrad_uses_x6(ops._fill_struct(capi.WgradDesc(), dict(wd, flags=capi.WG_CS3 | capi.WG_X6))) == 1
    nat = ops.conv_wgrad(wd, dyg, x4, torch.zeros(Co, 343, 4, device=DEV))
    got = ops.conv_wgrad(dict(wd, flags=capi.WG_CS3 | capi.WG_X6), dyg, x4, torch.zeros(Co, 343, 4, device=DEV))
    assert torch.all(got[:, :, 3] == 0), "padding column of g was written"
    e_nat, e_x6 = _rel(nat[:, :, :3].cpu(), ref), _rel(got[:, :, :3].cpu(), ref)
    assert e_x6 <= 1.05 * e_nat + 1e-9, "stem wgrad: bf16-split error %.3e vs native fp32 MFMA %.3e (against fp64)" % (e_x6, e_nat)
    assert e_x6 < 5e-6


@pytest.mark.parametrize("Ci,Co,k,thw,N", [(256, 288, (1, 1, 1), (1, 28, 28), 16), (528, 128, (1, 1, 1), (1, 28, 28), 8), (64, 96, (1, 3, 3), (2, 14, 30), 4)])
def test_x6_generic_wgrad_vs_fp64_and_native(Ci, Co, k, thw, N):
    """PC_WG_X6 on the generic split-K weight-gradient kernel (the 1x1x1 layers, widths that are not multiples of 28): 64- and 128-row
    tiles, fp32 atomics between the K slices."""
    g = torch.Generator().manual_seed(31)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g)))
    w = (torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))).double().requires_grad_(True)
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(xp, w, None, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    taps = int(np.prod(k))
    ref = w.grad.reshape(Co, Ci, taps).permute(0, 2, 1)
    pf = [p[0] for p in pads]
    xg, dyg = cl(x), cl(dy)
    wd = desc.wgrad(N, thw, Co, Co, thw, Ci, Ci, k, (1, 1, 1), pf)
    nat = ops.conv_wgrad(wd, dyg, xg, torch.zeros(Co, taps, Ci, device=DEV))
    got = ops.conv_wgrad(dict(wd, flags=capi.WG_X6), dyg, xg, torch.zeros(Co, taps, Ci, device=DEV))
    e_nat, e_x6 = _rel(nat.cpu(), ref), _rel(got.cpu(), ref)
    assert e_x6 <= 1.05 * e_nat + 1e-9, "generic wgrad: bf16-split error %.3e vs native fp32 MFMA %.3e (against fp64)" % (e_x6, e_nat)


# (Ci, Co, k, thw, N): launches whose last round of resident blocks is partly filled -- 196 tiles of 128 x 64 on 512 slots (every tile split in
# two K slices), and 3 x 512 + 32 tiles (the last 32 split eight ways)
TAIL_CASES = [(64, 128, (3, 3, 3), (2, 28, 28), 8), (64, 128, (3, 3, 3), (4, 28, 28), 32)]


@pytest.mark.parametrize("Ci,Co,k,thw,N", TAIL_CASES)
def test_x6_tail_split_matches_and_is_deterministic(Ci, Co, k, thw, N):
    """pc_conv_fwd_x6_ws: the tiles of the last, partly filled round run as K slices whose partial sums meet in the workspace; the block that
    arrives last adds them in slice order and runs the epilogue.  Against fp64 the result is held to the same bar as the unsplit launch
    (no further than the fp32-MFMA kernel), two runs are bit-identical (the adding order does not depend on which block comes last), the
    tile counters are back at zero, and the BatchNorm partial sums / ReLU of the epilogue see the complete sums."""
    g = torch.Generator().manual_seed(41)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g) * torch.exp(torch.randn(N, Ci, 1, 1, 1, generator=g)))
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    pf = [p[0] for p in pads]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw)
    n_ws = ops.conv_x6_ws_floats(d)
    assert n_ws > 0, "this shape should split its tail"
    xg, wk = cl(x), w_oki(w)
    wp = ops.split_planes(wk)
    ws = torch.zeros(n_ws, device=DEV)
    plain = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV))
    a = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV), ws=ws)
    torch.cuda.synchronize()
    tiles_split = n_ws - (n_ws // (128 * 64)) * (128 * 64)              # the counters sit behind whole 128 x 64 slices
    assert tiles_split > 0 and torch.all(ws[-tiles_split:] == 0), "tile counters must be left at zero"
    b = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV), ws=ws)
    assert torch.equal(a, b), "two runs of the split launch differ"
    assert not torch.equal(a, plain)                                     # it did split (another summation order) ...
    assert (a - plain).abs().max().item() <= 2e-6 * plain.abs().max().item()      # ... of the same sums
    if N <= 8:                                                           # fp64 reference on the CPU for the smaller case
        xp = F.pad(x.double(), (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
        y64 = F.conv3d(xp, w.double(), None, 1)
        nat = ops.conv_fwd(d, xg, wk, torch.empty(N, *thw, Co, device=DEV))
        e_nat, e_sp = _rel(uncl(nat), y64), _rel(uncl(a), y64)
        assert e_sp <= 1.05 * e_nat + 1e-9, "tail-split error %.3e vs native fp32 MFMA %.3e (against fp64)" % (e_sp, e_nat)
    # epilogue on complete sums: BatchNorm partials and ReLU (+ bias), grouped as the step runs them
    bias = torch.randn(Co, generator=g).to(DEV)
    d2 = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw, flags=capi.F_BNPART, groups=2)
    assert ops.conv_x6_ws_floats(d2) > 0
    rows = ops.conv_bnpart_rows(dict(d2, flags=d2["flags"] | capi.F_X6))
    p0 = torch.zeros(rows, 2, Co, device=DEV); p1 = torch.zeros(rows, 2, Co, device=DEV)
    ws2 = torch.zeros(ops.conv_x6_ws_floats(d2), device=DEV)
    o0 = ops.conv_fwd_x6(d2, xg, wp, torch.empty(N, *thw, Co, device=DEV), bnpart=p0)
    o1 = ops.conv_fwd_x6(d2, xg, wp, torch.empty(N, *thw, Co, device=DEV), bnpart=p1, ws=ws2)
    assert torch.allclose(o0, o1, rtol=0, atol=2e-6 * o0.abs().max().item())
    assert torch.allclose(p0.sum(0), p1.sum(0), rtol=1e-5, atol=1e-4)
    d3 = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), pf, thw, act=capi.ACT_RELU, flags=capi.F_BIAS)
    r0 = ops.conv_fwd_x6(d3, xg, wp, torch.empty(N, *thw, Co, device=DEV), bias=bias)
    r1 = ops.conv_fwd_x6(d3, xg, wp, torch.empty(N, *thw, Co, device=DEV), bias=bias, ws=torch.zeros(ops.conv_x6_ws_floats(d3), device=DEV))
    assert torch.allclose(r0, r1, rtol=0, atol=2e-6 * r0.abs().max().item()) and (r1 >= 0).all()


@pytest.mark.parametrize("Ci,Co,k,thw,N,iters", [(64, 128, (3, 3, 3), (2, 28, 28), 8, 100), (832, 544, (1, 9, 1), (1, 28, 1), 80, 40)])
def test_x6_tail_split_stress_beside_a_memory_bound_stream(Ci, Co, k, thw, N, iters):
    """The tail split's hand-off (slices stored / loaded sc0 sc1, one relaxed counter per tile, no fences; csrc/conv_x6.hip) under the
    conditions that break an invalid hand-off: many back-to-back launches, consumer caches warm, and a second stream that keeps HBM / L2 busy
    (uneven load).  Every launch must be bit-identical to the first and leave its tile counters at zero.  (tools/stress_tail_split.py is the
    500-launch version of this; VERDICT r4 #8 / ADVICE r4.)"""
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g))
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), [p[0] for p in pads], thw)
    n_ws = ops.conv_x6_ws_floats(d)
    assert n_ws > 0, "this shape should split its tail"
    xg, wk = cl(x), w_oki(w)
    wp = ops.split_planes(wk)
    ws = torch.zeros(n_ws, device=DEV)
    side = torch.cuda.Stream()
    junk_a = torch.randn(32 << 20, device=DEV)
    junk_b = torch.empty_like(junk_a)
    first = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV), ws=ws).clone()
    plain = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV))
    assert (first - plain).abs().max().item() <= 2e-6 * plain.abs().max().item()
    out = torch.empty_like(first)
    side.wait_stream(torch.cuda.current_stream())
    differ = 0
    for _ in range(iters):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a)                                   # 256 MB of HBM traffic beside the launch
        ops.conv_fwd_x6(d, xg, wp, out, ws=ws)
        differ += int(not torch.equal(out, first))
    torch.cuda.synchronize()
    assert differ == 0, "%d of %d tail-split launches differ from the first" % (differ, iters)
    nctr = n_ws % (128 * 64)                                       # the counters sit behind whole 128 x 64 slices
    assert nctr > 0 and torch.all(ws[-nctr:] == 0), "tile counters must be left at zero"


@pytest.mark.parametrize("Ci,Co,k,thw,N", [(64, 128, (3, 3, 3), (4, 28, 28), 32), (832, 544, (1, 9, 1), (1, 28, 1), 80)])
def test_x6_launches_from_an_idle_chip_are_bit_identical(Ci, Co, k, thw, N):
    """Round 5 regression: with the weight planes fetched a whole chunk ahead, hipcc dropped the vmcnt(0) of the K loop's barrier and a wave could
    read an LDS tile before its LDS-DMA had landed -- only when the transfers were slow, i.e. on a launch that starts from an idle chip with cold
    caches (126 of 150 such launches had 1 - 10 k wrong elements; back-to-back launches never did).  The barrier now carries the wait as inline asm
    (PC_SYNC_DMA, csrc/common.h); this test keeps the launch pattern that exposed it: host pause, ONE launch, compare."""
    import time
    g = torch.Generator().manual_seed(29)
    x = torch.relu(torch.randn(N, Ci, *thw, generator=g))
    w = torch.randn(Co, Ci, *k, generator=g) / np.sqrt(Ci * np.prod(k))
    pads = [spec.same_pad(thw[i], k[i], 1) for i in range(3)]
    d = desc.conv_fwd(N, thw, Ci, Ci, Co, Co, k, (1, 1, 1), [p[0] for p in pads], thw, act=capi.ACT_RELU, flags=capi.F_BIAS)
    xg, wk = cl(x), w_oki(w)
    wp = ops.split_planes(wk)
    bias = torch.randn(Co, generator=g).to(DEV)
    junk = torch.empty(96 << 20, device=DEV)                          # 384 MB: flushes L2 and the Infinity Cache between launches
    first = ops.conv_fwd_x6(d, xg, wp, torch.empty(N, *thw, Co, device=DEV), bias=bias).clone()
    out = torch.empty_like(first)
    differ = 0
    for it in range(40):
        junk.fill_(float(it))
        torch.cuda.synchronize()
        time.sleep(0.002)
        ops.conv_fwd_x6(d, xg, wp, out, bias=bias)
        differ += int(not torch.equal(out, first))
    assert differ == 0, "%d of 40 launches from an idle chip differ from the first" % differ
