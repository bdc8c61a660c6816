"""GPU: the Winograd F(4x4, 3x3) convolution (csrc/wino4.hip, pc_wino_desc.m = 4) against fp64 torch convolutions of the same op -- the cases
of tests/test_wino_gpu.py at H, W that are multiples of 4: forward with bias / ReLU / accumulate, channel slices + BatchNorm partial sums,
ragged channel counts, the input gradient through mirrored + transposed weights, the temporal stride-two map, and the real 112 x 112 /
56 x 56 / 28 x 28 shapes against the F(2x2, 3x3) kernel.  Tolerance: with its interpolation points at +-1/sqrt(2), +-sqrt(2) the kernel's
rms error is about 3x F(2x2, 3x3)'s (1.0e-6 against 3.5e-7 of the output rms on post-ReLU inputs, maximum 1.3e-5 against 3e-6:
tools/bench_wino4.py prints both); the bars are test_wino_gpu.py's own 2e-5 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from picons_amd import capi, ops

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _ref(x, w, b=None, kt=3):
    y = F.conv3d(x.permute(0, 4, 1, 2, 3), w, b, padding=(kt // 2, 1, 1))
    return y.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("N,T,H,W,Ci,Co,KT", [(2, 3, 12, 20, 16, 40, 3), (1, 1, 8, 8, 8, 4, 1), (3, 2, 28, 28, 24, 96, 3), (2, 4, 16, 36, 32, 64, 3),
                                               (1, 2, 4, 4, 8, 8, 3), (1, 2, 56, 56, 24, 72, 3), (2, 1, 40, 132, 8, 16, 3)])
def test_wino4_forward_matches_torch(N, T, H, W, Ci, Co, KT):
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, T, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, KT, 3, 3, generator=g) * (1.0 / np.sqrt(Ci * KT * 9))
    b = torch.randn(Co, generator=g)
    ref = _ref(x.double(), w.double(), b.double(), KT)
    xd, wd, bd = x.cuda(), w.cuda().contiguous(), b.cuda()
    U = ops.wino_weights(wd, Co, Ci, KT, m=4)
    out = torch.full((N, T, H, W, Co), float("nan"), device="cuda")
    d = ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, act=capi.ACT_NONE, flags=capi.F_BIAS, m=4)
    ops.wino_conv(d, xd, U, out, bias=bd)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), err
    base = torch.randn(N, T, H, W, Co, generator=g)
    out2 = base.cuda().clone()
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, flags=capi.F_ACCUM, m=4), xd, U, out2)
    ref2 = base.double() + _ref(x.double(), w.double(), None, KT)
    assert (out2.cpu().double() - ref2).abs().max().item() <= TOL * max(1.0, ref2.abs().max().item())
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, Ci, Co, Co, KT, act=capi.ACT_RELU, flags=capi.F_BIAS, m=4), xd, U, out, bias=bd)
    assert (out.cpu().double() - ref.clamp_min(0)).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())


def test_wino4_channel_slices_and_bn_partials():
    N, T, H, W, Ci, Co, ldi, ldo = 4, 2, 12, 12, 16, 32, 40, 48
    g = torch.Generator().manual_seed(5)
    xw = torch.randn(N, T, H, W, ldi, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1
    x = xw[..., 8:8 + Ci].contiguous()
    ref = _ref(x.double(), w.double(), None, 3)
    xd = xw.cuda()
    U = ops.wino_weights(w.cuda().contiguous(), Co, Ci, 3, m=4)
    outw = torch.zeros(N, T, H, W, ldo, device="cuda")
    d = ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, 3, flags=capi.F_BNPART, m=4)
    rows = capi.lib().pc_wino_bnpart_rows(d)
    assert rows > 0 and rows % N == 0
    part = torch.zeros(rows, 2, Co, device="cuda")
    ops.wino_conv(d, xd[..., 8:], U, outw[..., 4:], bnpart=part)
    assert (outw[..., 4:4 + Co].cpu().double() - ref).abs().max().item() <= TOL
    assert outw[..., :4].abs().max().item() == 0 and outw[..., 4 + Co:].abs().max().item() == 0
    out2 = torch.zeros(N, T, H, W, ldo, device="cuda")           # slice that is not 16-byte aligned: 4-byte stores
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, ldi, Co, ldo, 3, m=4), xd[..., 8:], U, out2[..., 2:])
    assert torch.equal(out2[..., 2:2 + Co], outw[..., 4:4 + Co]) and out2[..., :2].abs().max().item() == 0 and out2[..., 2 + Co:].abs().max().item() == 0
    per = rows // N
    for n in range(N):
        s = part[n * per:(n + 1) * per].sum(0).cpu().double()
        assert (s[0] - ref[n].sum(dim=(0, 1, 2))).abs().max().item() <= 1e-3
        assert (s[1] - (ref[n] ** 2).sum(dim=(0, 1, 2))).abs().max().item() <= 1e-3 * max(1.0, (ref[n] ** 2).sum(dim=(0, 1, 2)).max().item())


@pytest.mark.parametrize("Co", [6, 66])
def test_wino4_ragged_channel_count_leaves_neighbours_untouched(Co):
    N, T, H, W, Ci, ldo = 2, 2, 12, 12, 16, 80
    g = torch.Generator().manual_seed(Co)
    x = torch.randn(N, T, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    ref = _ref(x.double(), w.double(), b.double(), 3)
    U = ops.wino_weights(w.cuda().contiguous(), Co, Ci, 3, m=4)
    outw = torch.full((N, T, H, W, ldo), 7.25, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Ci, Ci, Co, ldo, 3, flags=capi.F_BIAS, m=4), x.cuda(), U, outw[..., 4:], bias=b.cuda())
    assert (outw[..., 4:4 + Co].cpu().double() - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert torch.all(outw[..., :4] == 7.25) and torch.all(outw[..., 4 + Co:] == 7.25)


@pytest.mark.parametrize("KT", [3, 1])
def test_wino4_input_gradient_through_flipped_weights(KT):
    N, T, H, W, Ci, Co = 2, 3, 12, 16, 24, 16
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, T, H, W, Ci, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, KT, 3, 3, generator=g, dtype=torch.float64) * 0.1)
    dy = torch.randn(N, T, H, W, Co, generator=g, dtype=torch.float64)
    y = _ref(x, w, None, KT)
    (y * dy).sum().backward()
    wd = w.float().cuda().contiguous()
    U = ops.wino_weights(wd, Ci, Co, KT, flip=True, strides=(KT * 9, 1, Ci * KT * 9), m=4)
    dx = torch.empty(N, T, H, W, Ci, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Co, Co, Ci, Ci, KT, m=4), dy.float().cuda(), U, dx)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= TOL * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("T", [4, 5])
def test_wino4_temporal_stride_two_forward_and_input_gradient(T):
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
    (y * dy).sum().backward()
    wd = w.float().cuda().contiguous()
    U = ops.wino_weights(wd, Co, Ci, 3, m=4)
    out = torch.empty(N, To, H, W, Co, device="cuda")
    ops.wino_conv(ops.wino_desc(N, To, H, W, Ci, Ci, Co, Co, 3, Ti=T, ta=s, tc=-front, tden=1, m=4), x.detach().float().cuda(), U, out)
    assert (out.cpu().double() - y.detach()).abs().max().item() <= TOL * max(1.0, y.abs().max().item())
    Ut = ops.wino_weights(wd, Ci, Co, 3, flip=True, strides=(27, 1, Ci * 27), m=4)
    dx = torch.empty(N, T, H, W, Ci, device="cuda")
    ops.wino_conv(ops.wino_desc(N, T, H, W, Co, Co, Ci, Ci, 3, Ti=To, ta=1, tc=front - 2, tden=s, m=4), dy.float().cuda(), Ut, dx)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= TOL * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("N,thw,Ci,Co", [(8, (4, 112, 112), 64, 64), (1, (2, 224, 224), 64, 64), (8, (2, 56, 56), 64, 192), (8, (2, 56, 56), 192, 64), (8, (2, 28, 28), 96, 128)])
def test_wino4_real_shapes_vs_f23_kernel(N, thw, Ci, Co):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, *thw, Ci, device="cuda", generator=g).clamp_min(0)           # post-ReLU activations, as in the step
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) * (1.0 / np.sqrt(27 * Ci))
    b = torch.randn(Co, device="cuda", generator=g) * 0.1
    outs = {}
    for m in (2, 4):
        U = ops.wino_weights(w, Co, Ci, 3, m=m)
        outs[m] = torch.empty(N, *thw, Co, device="cuda")
        ops.wino_conv(ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, act=capi.ACT_RELU, flags=capi.F_BIAS, m=m), x, U, outs[m], bias=b)
    torch.cuda.synchronize()
    err = (outs[4] - outs[2]).abs().max().item()
    assert err <= TOL * max(1.0, outs[2].abs().max().item()), err
    # bit-identical from launch to launch (no atomics, fixed reduction order)
    again = torch.empty_like(outs[4])
    ops.wino_conv(ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, act=capi.ACT_RELU, flags=capi.F_BIAS, m=4), x, ops.wino_weights(w, Co, Ci, 3, m=4), again, bias=b)
    assert torch.equal(again, outs[4])


def test_wino4_real_shape_vs_fp64_torch():
    """F(4x4, 3x3) at a real shape of the step (conv56's forward: 8 x (2, 56, 56), 64 -> 192, bias + ReLU) against an fp64 torch
    convolution directly -- the small shapes above are, the real ones were only compared with the F(2x2, 3x3) kernel."""
    N, thw, Ci, Co = 8, (2, 56, 56), 64, 192
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, *thw, Ci, generator=g).clamp_min(0)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * (1.0 / np.sqrt(27 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    ref = torch.relu(F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), b.double(), padding=1)).permute(0, 2, 3, 4, 1)
    xd, wd = x.cuda(), w.cuda().contiguous()
    out = torch.empty(N, *thw, Co, device="cuda")
    ops.wino_conv(ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, act=capi.ACT_RELU, flags=capi.F_BIAS, m=4), xd, ops.wino_weights(wd, Co, Ci, 3, m=4), out, bias=b.cuda())
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), err
    rms = ((out.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rms <= 2e-6, rms          # measured 1.0e-6 of the output rms with the points (0, +-1/sqrt2, +-sqrt2, inf)


def test_wino4_refuses_shapes_it_cannot_tile():
    d = ops.wino_desc(1, 1, 14, 14, 8, 8, 8, 8, 3, m=4)
    assert capi.lib().pc_wino_bnpart_rows(d) == -1
    with pytest.raises(Exception):
        ops.wino_conv(d, torch.zeros(1, 1, 14, 14, 8, device="cuda"), torch.zeros(1 << 16, device="cuda"), torch.zeros(1, 1, 14, 14, 8, device="cuda"))
    d.m = 3
    with pytest.raises(Exception):
        ops.wino_conv(d, torch.zeros(1, 1, 14, 14, 8, device="cuda"), torch.zeros(1 << 16, device="cuda"), torch.zeros(1, 1, 14, 14, 8, device="cuda"))


def test_wino4_launches_from_an_idle_chip_are_bit_identical():
    """The launch pattern that exposed a missing wait in conv_x6.hip this round (tests/test_x6_gpu.py): host pause, caches flushed, ONE launch.
    This kernel's weight fragments come straight from global memory a chunk ahead and its raw patch by LDS-DMA two chunks ahead; slow transfers
    (cold caches, idle clocks) are when a missing wait shows."""
    import time
    g = torch.Generator(device="cuda").manual_seed(9)
    N, thw, Ci, Co = 4, (2, 56, 56), 64, 192
    x = torch.randn(N, *thw, Ci, device="cuda", generator=g).clamp_min(0)
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) * (1.0 / np.sqrt(27 * Ci))
    U = ops.wino_weights(w, Co, Ci, 3, m=4)
    d = ops.wino_desc(N, *thw, Ci, Ci, Co, Co, 3, m=4)
    first = ops.wino_conv(d, x, U, torch.empty(N, *thw, Co, device="cuda")).clone()
    junk = torch.empty(96 << 20, device="cuda")
    out = torch.empty_like(first)
    differ = 0
    for it in range(30):
        junk.fill_(float(it))
        torch.cuda.synchronize()
        time.sleep(0.002)
        ops.wino_conv(d, x, U, out)
        differ += int(not torch.equal(out, first))
    assert differ == 0, "%d of 30 launches from an idle chip differ from the first" % differ
