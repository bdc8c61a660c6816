"""GPU: the Winograd F(2x2, 3x3) convolution (csrc/wino.hip) against plain fp32/fp64 torch convolutions of the same op --
forward with bias / ReLU / BatchNorm partial sums, the input gradient through mirrored + transposed weights (flip), accumulate,
ragged tile blocks (tiles per side not a multiple of the block rectangle), channel counts that are not multiples of 64 -- and,
at the real conv112 / Conv3d_2c shapes, against the gather-GEMM kernel it replaces."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import capi, desc as D, ops

pytestmark = pytest.mark.gpu


def _ref(x, w, b=None, kt=3):
    """x (N,T,H,W,Ci) fp64 cpu, w (Co,Ci,kt,3,3) -> (N,T,H,W,Co)."""
    y = F.conv3d(x.permute(0, 4, 1, 2, 3), w, b, padding=(kt // 2, 1, 1))
    return y.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("N,T,H,W,Ci,Co,KT", [(2, 3, 12, 20, 16, 40, 3), (1, 1, 8, 8, 8, 4, 1), (3, 2, 28, 28, 24, 96, 3), (2, 4, 18, 34, 32, 64, 3)])
def test_wino_forward_matches_torch(N, T, H, W, Ci, Co, KT):
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, T, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, KT, 3, 3, generator=g) * (1.0 / np.sqrt(Ci * KT * 9))
    b = torch.randn(Co, generator=g)
    ref = _ref(x.double(), w.double(), b.double(), KT)
    xd, wd, bd = x.cuda(), w.cuda().contiguous(), b.cuda()
    U = ops.wino_weights(wd, Co, Ci, KT)
    out = torch.full((N, T, H, W, Co), float("nan"), device="cuda")
    d = ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, act=capi.ACT_NONE, flags=capi.F_BIAS)
    ops.wino_conv(d, xd, U, out, bias=bd)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    # ReLU + accumulate into an existing tensor (the input-gradient epilogue)
    base = torch.randn(N, T, H, W, Co, generator=g)
    out2 = base.cuda().clone()
    d2 = ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, act=capi.ACT_NONE, flags=capi.F_ACCUM)
    ops.wino_conv(d2, xd, U, out2)
    ref2 = base.double() + _ref(x.double(), w.double(), None, KT)
    assert (out2.cpu().double() - ref2).abs().max().item() <= 2e-5 * max(1.0, ref2.abs().max().item())
    d3 = ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, act=capi.ACT_RELU, flags=capi.F_BIAS)
    ops.wino_conv(d3, xd, U, out, bias=bd)
    assert (out.cpu().double() - ref.clamp_min(0)).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_wino_channel_slices_and_bn_partials():
    """Reads a channel slice of a wider tensor (ldi > Ci), writes a slice of a wider one (ldo > Co), and leaves BatchNorm partial
    sums whose totals are the column sums / sums of squares of the output (per sample: rows of a sample are consecutive)."""
    N, T, H, W, Ci, Co, ldi, ldo = 4, 2, 12, 12, 16, 32, 40, 48
    g = torch.Generator().manual_seed(5)
    xw = torch.randn(N, T, H, W, ldi, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1
    x = xw[..., 8:8 + Ci].contiguous()
    ref = _ref(x.double(), w.double(), None, 3)
    xd = xw.cuda()
    U = ops.wino_weights(w.cuda().contiguous(), Co, Ci, 3)
    outw = torch.zeros(N, T, H, W, ldo, device="cuda")
    d = ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, 3, flags=capi.F_BNPART)
    rows = capi.lib().pc_wino_bnpart_rows(d)
    part = torch.zeros(rows, 2, Co, device="cuda")
    ops.wino_conv(d, xd[..., 8:], U, outw[..., 4:], bnpart=part)
    assert (outw[..., 4:4 + Co].cpu().double() - ref).abs().max().item() <= 2e-5
    assert outw[..., :4].abs().max().item() == 0 and outw[..., 4 + Co:].abs().max().item() == 0
    # a slice that is not 16-byte aligned (channel offset 2): the kernel falls back from its row-contiguous 16-byte stores to 4-byte ones
    out2 = torch.zeros(N, T, H, W, ldo, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, 3), xd[..., 8:], U, out2[..., 2:])
    assert torch.equal(out2[..., 2:2 + Co], outw[..., 4:4 + Co]) and out2[..., :2].abs().max().item() == 0 and out2[..., 2 + Co:].abs().max().item() == 0
    per = rows // N
    for n in range(N):
        s = part[n * per:(n + 1) * per].sum(0).cpu().double()
        assert (s[0] - ref[n].sum(dim=(0, 1, 2))).abs().max().item() <= 1e-3
        assert (s[1] - (ref[n] ** 2).sum(dim=(0, 1, 2))).abs().max().item() <= 1e-3 * max(1.0, (ref[n] ** 2).sum(dim=(0, 1, 2)).max().item())


@pytest.mark.parametrize("Co", [6, 66])
def test_wino_output_slice_with_ragged_channel_count_leaves_neighbours_untouched(Co):
    """Co % 4 != 0 inside a wider tensor (ldo > Co, 16-byte aligned slice): the row-contiguous 16-byte epilogue must not be taken -- its last
    chunk would run over columns Co .. Co + 3, which belong to the neighbouring channel slice."""
    N, T, H, W, Ci, ldo = 2, 2, 12, 12, 16, 80
    g = torch.Generator().manual_seed(Co)
    x = torch.randn(N, T, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    ref = _ref(x.double(), w.double(), b.double(), 3)
    U = ops.wino_weights(w.cuda().contiguous(), Co, Ci, 3)
    outw = torch.full((N, T, H, W, ldo), 7.25, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, Ci, Co, ldo, 3, flags=capi.F_BIAS), x.cuda(), U, outw[..., 4:], bias=b.cuda())
    assert (outw[..., 4:4 + Co].cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.all(outw[..., :4] == 7.25) and torch.all(outw[..., 4 + Co:] == 7.25)


@pytest.mark.parametrize("KT", [3, 1])
def test_wino_input_gradient_through_flipped_weights(KT):
    """d(x) of y = conv(x, w) is the same correlation of d(y) with the taps mirrored and the channel roles exchanged:
    pc_wino_weights(flip=1) reads the master OIDHW weights through strides (O' = Ci, I' = Co)."""
    N, T, H, W, Ci, Co = 2, 3, 10, 14, 24, 16
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, T, H, W, Ci, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, KT, 3, 3, generator=g, dtype=torch.float64) * 0.1)
    dy = torch.randn(N, T, H, W, Co, generator=g, dtype=torch.float64)
    y = _ref(x, w, None, KT)
    (y * dy).sum().backward()
    wd = w.float().cuda().contiguous()
    U = ops.wino_weights(wd, Ci, Co, KT, flip=True, strides=(KT * 9, 1, Ci * KT * 9))
    dx = torch.empty(N, T, H, W, Ci, device="cuda")
    d = ops.wino_desc(N, T, H, W, Co, Co, Ci, Ci, KT)
    ops.wino_conv(d, dy.float().cuda(), U, dx)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("T", [4, 5])
def test_wino_temporal_stride_two_forward_and_input_gradient(T):
    """Conv3d_2c (pytorch_i3d.py:236-238 as this model strides it): 3x3x3, stride (2, 1, 1), TF-SAME padding along t (front = pad // 2,
    Unit3D.compute_pad :82-109).  Forward frame t reads input frames 2t - front + kt; the input gradient is the transposed map
    (only every other (frame, tap) pair exists)."""
    N, H, W, Ci, Co, s = 2, 8, 12, 16, 24, 2
    pad = max(3 - s, 0) if T % s == 0 else max(3 - T % s, 0)
    front, back = pad // 2, pad - pad // 2
    To = (T + pad - 3) // s + 1
    g = torch.Generator().manual_seed(11 + T)
    x = torch.randn(N, T, H, W, Ci, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(N, To, H, W, Co, generator=g, dtype=torch.float64)
    xp = F.pad(x.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, front, back))
    y = F.conv3d(xp, w, stride=(s, 1, 1)).permute(0, 2, 3, 4, 1)
    assert y.shape[1] == To
    (y * dy).sum().backward()
    wd = w.float().cuda().contiguous()
    U = ops.wino_weights(wd, Co, Ci, 3)
    out = torch.empty(N, To, H, W, Co, device="cuda")
    ops.wino_conv(ops.wino_desc(N, To, H, W, Ci, Ci, Co, Co, 3, Ti=T, ta=s, tc=-front, tden=1), x.detach().float().cuda(), U, out)
    assert (out.cpu().double() - y.detach()).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    Ut = ops.wino_weights(wd, Ci, Co, 3, flip=True, strides=(27, 1, Ci * 27))
    dx = torch.empty(N, T, H, W, Ci, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Co, Co, Ci, Ci, 3, Ti=To, ta=1, tc=front - 2, tden=s), dy.float().cuda(), Ut, dx)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("thw,Ci,Co", [((4, 112, 112), 64, 64), ((2, 56, 56), 64, 192), ((2, 56, 56), 192, 64)])
def test_wino_real_shapes_vs_gather_gemm(thw, Ci, Co):
    N = 16
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, *thw, Ci, device="cuda", generator=g).clamp_min(0)           # post-ReLU activations, as in the step
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) * (1.0 / np.sqrt(27 * Ci))
    b = torch.randn(Co, device="cuda", generator=g) * 0.1
    wk = w.permute(0, 2, 3, 4, 1).reshape(Co, 27, Ci).contiguous()
    dd = D.trim_conv(D.conv_fwd(N, thw, Ci, Ci, Co, Co, (3, 3, 3), (1, 1, 1), (1, 1, 1), thw, act=capi.ACT_RELU, flags=capi.F_BIAS, groups=2))
    ref = torch.empty(N, *thw, Co, device="cuda")
    ops.conv_fwd(dd, x, wk, ref, bias=b)
    U = ops.wino_weights(w, Co, Ci, 3)
    out = torch.empty_like(ref)
    ops.wino_conv(ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, act=capi.ACT_RELU, flags=capi.F_BIAS), x, U, out, bias=b)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    assert err <= 5e-5 * max(1.0, ref.abs().max().item()), err


# ---- strip mode (round 6, PC_F_STRIPS): frames whose tile grid is a multiple of 14 wide run as blocks of two 2 x 14-tile strips taken from a PAIR of
# planes (56 of 64 tile slots instead of 49): csrc/wino.hip.  Asked for per launch; the planner leaves it off (measured neutral).
ST = capi.F_STRIPS
@pytest.mark.parametrize("N,T,HW,Ci,Co,KT", [(4, 1, 28, 32, 96, 3), (4, 2, 28, 24, 64, 3), (8, 2, 56, 16, 64, 3), (4, 1, 28, 40, 70, 1), (12, 3, 28, 8, 8, 3)])
def test_wino_strip_blocks_match_torch_and_rectangle_blocks(N, T, HW, Ci, Co, KT):
    """Forward with bias + BatchNorm partial sums into a channel slice of a wider tensor, then accumulate + ReLU-free second pass, against an fp64
    torch convolution; the first two samples of the same input through a launch with N = 2 (rectangle blocks: N % 4 != 0) give bit-identical outputs
    (a tile's arithmetic does not depend on how tiles are grouped into blocks), and so does the whole batch without the flag; the BatchNorm partial rows of each sample pair sum to the column
    sums of its outputs (a block's rows belong to one plane pair)."""
    H = W = HW
    ldi, ldo = Ci + 8, Co + 12
    g = torch.Generator().manual_seed(N * 10 + T + HW)
    xw = torch.randn(N, T, H, W, ldi, generator=g)
    w = torch.randn(Co, Ci, KT, 3, 3, generator=g) * (1.0 / np.sqrt(Ci * KT * 9))
    b = torch.randn(Co, generator=g)
    x = xw[..., 4:4 + Ci].contiguous()
    ref = _ref(x.double(), w.double(), b.double(), KT)
    xd, bd = xw.cuda(), b.cuda()
    U = ops.wino_weights(w.cuda().contiguous(), Co, Ci, KT)
    d = ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, KT, flags=capi.F_BIAS | capi.F_BNPART | ST)
    rows = capi.lib().pc_wino_bnpart_rows(d)
    assert rows == (N // 2) * T * (H // 4) * (W // 28) * 2            # strips per plane = H / 4 blocks per plane pair, two partial rows per block
    part = torch.zeros(rows, 2, Co, device="cuda")
    outw = torch.zeros(N, T, H, W, ldo, device="cuda")
    ops.wino_conv(d, xd[..., 4:], U, outw[..., 8:], bias=bd, bnpart=part)
    got = outw[..., 8:8 + Co]
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert (got.cpu().double() - ref).abs().max().item() <= tol
    assert outw[..., :8].abs().max().item() == 0 and outw[..., 8 + Co:].abs().max().item() == 0
    per = rows // (N // 2)
    for q in range(N // 2):
        sm = part[q * per:(q + 1) * per].sum(0).cpu().double()
        r2 = ref[2 * q:2 * q + 2]
        assert (sm[0] - r2.sum(dim=(0, 1, 2, 3))).abs().max().item() <= 2e-3 * max(1.0, r2.abs().sum(dim=(0, 1, 2, 3)).max().item() * 1e-3)
        assert (sm[1] - (r2 ** 2).sum(dim=(0, 1, 2, 3))).abs().max().item() <= 1e-3 * max(1.0, (r2 ** 2).sum(dim=(0, 1, 2, 3)).max().item())
    # rectangle blocks (no flag) on the whole batch and on the first two samples: bit-identical
    o3 = torch.zeros(N, T, H, W, ldo, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, KT, flags=capi.F_BIAS), xd[..., 4:], U, o3[..., 8:], bias=bd)
    assert torch.equal(o3, outw)
    o2 = torch.zeros(2, T, H, W, ldo, device="cuda")
    ops.wino_conv(ops.wino_desc(2, T, H, W, Ci, ldi, Co, ldo, KT, flags=capi.F_BIAS), xd[:2, ..., 4:], U, o2[..., 8:], bias=bd)
    assert torch.equal(o2[..., 8:8 + Co], got[:2])
    # accumulate into an existing tensor
    base = torch.randn(N, T, H, W, Co, generator=g)
    acc = base.cuda().clone()
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, ldi, Co, Co, KT, flags=capi.F_ACCUM | ST), xd[..., 4:], U, acc)
    ref2 = base.double() + _ref(x.double(), w.double(), None, KT)
    assert (acc.cpu().double() - ref2).abs().max().item() <= 2e-5 * max(1.0, ref2.abs().max().item())


def test_wino_strip_blocks_temporal_stride_two_and_input_gradient():
    """Conv3d_2c's shape class in strip mode (56 x 56 frames: 14 strips per plane): forward with temporal stride 2 and its transposed input gradient."""
    N, T, H, W, Ci, Co, s = 4, 4, 56, 56, 16, 64, 2
    pad = max(3 - s, 0)
    front, back = pad // 2, pad - pad // 2
    To = (T + pad - 3) // s + 1
    g = torch.Generator().manual_seed(19)
    x = torch.randn(N, T, H, W, Ci, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(N, To, H, W, Co, generator=g, dtype=torch.float64)
    xp = F.pad(x.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, front, back))
    y = F.conv3d(xp, w, stride=(s, 1, 1)).permute(0, 2, 3, 4, 1)
    (y * dy).sum().backward()
    wd = w.float().cuda().contiguous()
    out = torch.empty(N, To, H, W, Co, device="cuda")
    ops.wino_conv(ops.wino_desc(N, To, H, W, Ci, Ci, Co, Co, 3, Ti=T, ta=s, tc=-front, tden=1, flags=ST), x.detach().float().cuda(), ops.wino_weights(wd, Co, Ci, 3), out)
    assert (out.cpu().double() - y.detach()).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    Ut = ops.wino_weights(wd, Ci, Co, 3, flip=True, strides=(27, 1, Ci * 27))
    dx = torch.empty(N, T, H, W, Ci, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Co, Co, Ci, Ci, 3, Ti=To, ta=1, tc=front - 2, tden=s, flags=ST), dy.float().cuda(), Ut, dx)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())
