"""Thin torch-tensor wrappers over the C-ABI (one per entry point).  PyTorch is used for device
memory and streams only; every wrapper launches HIP kernels from libpicons.so on torch's current
stream and raises if the tensor is not on a GPU - there is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import capi, desc as D


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("picons ops need CUDA/HIP tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _fill_struct(st, d):
    for name, ctype in st._fields_:
        v = d[name]
        if isinstance(v, (list, tuple)):
            setattr(st, name, (C.c_int32 * 3)(*[int(x) for x in v]))
        else:
            setattr(st, name, ctype(v))
    return st


def conv_desc(d):
    return _fill_struct(capi.ConvDesc(), d)


def conv_bnpart_rows(d):
    return capi.lib().pc_conv_bnpart_rows(C.byref(conv_desc(d)))


def conv_fwd(d, x, w, out, bias=None, cscale=None, bnpart=None):
    capi.call("pc_conv_fwd", C.byref(conv_desc(d)), ptr(x), ptr(w), ptr(bias), ptr(cscale), ptr(out), ptr(bnpart), stream())
    return out


def split_planes(w):
    """fp32 device tensor (numel % 4 == 0) -> int16 tensor [3][numel]: the bf16 terms h, m, l of every element (pc_split_planes)."""
    n = w.numel()
    planes = torch.empty(3, n, device=w.device, dtype=torch.int16)
    capi.call("pc_split_planes", ptr(w), ptr(planes), n, n, stream())
    return planes


def conv_x6_ws_floats(d):
    """Floats of workspace pc_conv_fwd_x6_ws would use to split the tiles of the launch's last round into K slices (0: it would not)."""
    dd = dict(d)
    dd["flags"] = int(dd.get("flags", 0)) | capi.F_X6
    return int(capi.lib().pc_conv_x6_ws_floats(C.byref(conv_desc(dd))))


def conv_fwd_x6(d, x, wplanes, out, bias=None, cscale=None, bnpart=None, ws=None):
    """pc_conv_fwd_x6: d as for conv_fwd (PC_F_X6 is added here), wplanes = split_planes(w) of the [Co][taps][ldw] weights.
    ws: a ZEROED float32 device tensor of conv_x6_ws_floats(d) elements (or more) lets the launch split its tail tiles along K."""
    dd = dict(d)
    dd["flags"] = int(dd.get("flags", 0)) | capi.F_X6
    if ws is None:
        capi.call("pc_conv_fwd_x6", C.byref(conv_desc(dd)), ptr(x), ptr(wplanes), wplanes.shape[1], ptr(bias), ptr(cscale), ptr(out), ptr(bnpart), stream())
    else:
        capi.call("pc_conv_fwd_x6_ws", C.byref(conv_desc(dd)), ptr(x), ptr(wplanes), wplanes.shape[1], ptr(bias), ptr(cscale), ptr(out), ptr(bnpart),
                  ptr(ws), ws.numel(), stream())
    return out


def wino_desc(N, T, H, W, Ci, ldi, Co, ldo, KT=3, act=0, flags=0, Ti=None, ta=1, tc=None, tden=1, m=2):
    """pc_wino_desc; defaults = temporal stride 1 with padding KT // 2 (see include/picons.h for (ta, tc, tden)); m = 2: F(2x2, 3x3),
    4: F(4x4, 3x3) (U from wino_weights(..., m=4))."""
    st = capi.WinoDesc()
    st.m = m
    st.N, st.T, st.H, st.W, st.Ci, st.ldi, st.Co, st.ldo, st.KT, st.act, st.flags = N, T, H, W, Ci, ldi, Co, ldo, KT, act, flags
    st.Ti, st.ta, st.tc, st.tden = (T if Ti is None else Ti), ta, (-(KT // 2) if tc is None else tc), tden
    return st


def wino_weights(w, O, I, KT=3, flip=False, strides=None, out=None, m=2):
    """Transform-domain weights of pc_wino_conv from a weight tensor addressed as w[o*sO + tap*sT + i*sI] (default: contiguous
    OIDHW, i.e. (O, I, KT, 3, 3))."""
    sO, sT, sI = strides if strides is not None else (I * KT * 9, 1, KT * 9)
    fam = "pc_wino4" if m == 4 else "pc_wino"
    n = getattr(capi.lib(), fam + "_u_floats")(O, I, KT)
    U = out if out is not None else torch.empty(n, device=w.device, dtype=torch.float32)
    capi.call(fam + "_weights", ptr(w), int(sO), int(sT), int(sI), O, I, KT, int(flip), ptr(U), stream())
    return U


def wino_conv(d, x, U, out, bias=None, bnpart=None):
    capi.call("pc_wino_conv", C.byref(d), ptr(x), ptr(U), ptr(bias), ptr(out), ptr(bnpart), stream())
    return out


def conv_wgrad(d, Dt, St, g):
    capi.call("pc_conv_wgrad", C.byref(_fill_struct(capi.WgradDesc(), d)), ptr(Dt), ptr(St), ptr(g), stream())
    return g


def wgrad_slices(d):
    """K slices pc_conv_wgrad makes for descriptor d (host-only): the workspace images an ordered (atomic-free) launch needs."""
    n = capi.lib().pc_wgrad_slices(C.byref(_fill_struct(capi.WgradDesc(), dict(d, ws_slices=0))))
    if n < 1:
        raise RuntimeError("pc_wgrad_slices: %s" % capi.lib().pc_last_error().decode())
    return int(n)


def conv_wgrad_ordered(d, Dt, St, ws=None):
    """The weight gradient without atomics: the K slices leave their partial sums in `ws` ([slices][Cd][taps][Cs], zero before its first
    use) and are added in slice order -> (g [Cd][taps][Cs], ws).  Bit-identical from run to run."""
    n = wgrad_slices(d)
    taps = d["KT"] * d["KH"] * d["KW"]
    image = d["gbstride"] * d["nbatch"] if d.get("nbatch", 0) > 1 else d["Cd"] * taps * d["Cs"]
    if ws is None:
        ws = torch.zeros(n * image, device=Dt.device, dtype=torch.float32)
    conv_wgrad(dict(d, ws_slices=n), Dt, St, ws)
    g = ws.view(n, image)[0].clone()
    for k in range(1, n):
        g += ws.view(n, image)[k]
    return g, ws


def wgrad_fold(ws, image_floats, nslices):
    """pc_wgrad_fold: the K-slice images of `ws` folded in place to groups of pc_wgrad_fold_group() (slice order)."""
    capi.call("pc_wgrad_fold", ptr(ws), int(image_floats), int(nslices), stream())
    return ws


def wgrad_job_table(jobs):
    """[(desc dict, D ptr, S ptr, g ptr)] with integer device addresses -> host array of pc_wgrad_job."""
    tab = np.zeros(len(jobs), dtype=capi.WJOB_DTYPE)
    for q, (d, Dp, Sp, gp) in enumerate(jobs):
        tab[q]["d"][:] = D.flatten(d, D.WGRAD_FIELDS)
        tab[q]["D"], tab[q]["S"], tab[q]["g"] = Dp, Sp, gp
    return tab


def conv_wgrad_multi(jobs):
    """jobs: [(desc dict, D tensor, S tensor, g tensor)]: pc_conv_wgrad_multi (one grid for the generic split-K problems)."""
    tab = wgrad_job_table([(d, Dt.data_ptr(), St.data_ptr(), g.data_ptr()) for d, Dt, St, g in jobs])
    for _d, Dt, St, g in jobs:
        ptr(Dt); ptr(St); ptr(g)
    capi.call("pc_conv_wgrad_multi", C.c_void_p(tab.ctypes.data), len(jobs), stream())


def bn_finalize(part, npg, groups, C_, count, gamma, beta, eps, momentum, rmean=None, rvar=None, two_stage=False):
    """two_stage: with the workspace of pc_bn_finalize_ws (the plan's form; a no-op below 512 partial rows per group)."""
    stat = torch.empty(groups, 4, C_, device=part.device, dtype=torch.float32)
    if two_stage:
        n = capi.lib().pc_bn_finalize_ws_floats(npg, groups, C_)
        ws = torch.empty(max(int(n), 1), device=part.device, dtype=torch.float32)
        capi.call("pc_bn_finalize_ws", ptr(part), npg, groups, C_, int(count), ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar),
                  ptr(stat), ptr(ws) if n > 0 else None, stream())
    else:
        capi.call("pc_bn_finalize", ptr(part), npg, groups, C_, int(count), ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar),
                  ptr(stat), stream())
    return stat


def bn_finalize_apply(part, npg, groups, C_, count, gamma, beta, eps, momentum, rmean, rvar, z, ldz, rows, y, ldy, relu=True):
    """pc_bn_finalize_apply: the statistics and the normalised output in one launch (npg <= 256 partial rows per group) -> stat."""
    stat = torch.empty(groups, 4, C_, device=part.device, dtype=torch.float32)
    capi.call("pc_bn_finalize_apply", ptr(part), npg, groups, C_, int(count), ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar), ptr(stat),
              ptr(z), ldz, int(rows), ptr(y), ldy, int(relu), stream())
    return stat


def bn_apply(z, ldz, stat, C_, rows, groups, y, ldy, relu=True):
    capi.call("pc_bn_apply", ptr(z), ldz, ptr(stat), C_, int(rows), groups, ptr(y), ldy, int(relu), stream())
    return y


def bn_eval_stat(gamma, beta, rm, rv, eps):
    stat = torch.empty(1, 4, gamma.numel(), device=gamma.device, dtype=torch.float32)
    capi.call("pc_bn_eval_stat", ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, gamma.numel(), ptr(stat), stream())
    return stat


def bn_bwd(dy, lddy, z, ldz, stat, C_, rows, groups, relu, dz, lddz, dgamma, dbeta, accum=False, fused=False):
    """fused: the finalize folded into the apply kernel (bit 1 of pc_bn_bwd's `relu`)."""
    ws = torch.empty(capi.lib().pc_bn_bwd_ws_floats(int(rows), C_, groups), device=dy.device, dtype=torch.float32)
    capi.call("pc_bn_bwd", ptr(dy), lddy, ptr(z), ldz, ptr(stat), C_, int(rows), groups, int(bool(relu)) | (2 if fused else 0), ptr(dz), lddz, ptr(dgamma), ptr(dbeta),
              int(accum), ptr(ws), stream())
    return dz


def maxpool_fwd(d, x, y, argmax):
    capi.call("pc_maxpool_fwd", C.byref(_fill_struct(capi.PoolDesc(), d)), ptr(x), ptr(y), ptr(argmax), stream())


def maxpool_bwd(d, dy, argmax, dx, accum=False):
    capi.call("pc_maxpool_bwd", C.byref(_fill_struct(capi.PoolDesc(), d)), ptr(dy), ptr(argmax), ptr(dx), int(accum), stream())


def channel_scale(x, ldx, scale, N, pos_per_n, C_, y, ldy, accum=False):
    capi.call("pc_channel_scale", ptr(x), ldx, ptr(scale), N, int(pos_per_n), C_, ptr(y), ldy, int(accum), stream())


def act_bwd(dy, lddy, y, ldy, act, C_, rows, dz, lddz, dbias=None, accum=False):
    ws = torch.empty(capi.lib().pc_act_bwd_ws_floats(int(rows), C_), device=dy.device, dtype=torch.float32)
    capi.call("pc_act_bwd", ptr(dy), lddy, ptr(y), ldy, act, C_, int(rows), ptr(dz), lddz, ptr(dbias), int(accum), ptr(ws), stream())


def to_ndhwc(src, Cpad=None, flipw=False):
    """src (N,C,T,H,W) fp32/fp64 contiguous -> (N,T,H,W,Cpad) fp32."""
    N, Cc, T, H, W = src.shape
    Cpad = Cpad or Cc
    dst = torch.empty(N, T, H, W, Cpad, device=src.device, dtype=torch.float32)
    capi.call("pc_ncdhw_to_ndhwc", ptr(src.contiguous()), int(src.dtype == torch.float64), N, Cc, T * H * W, W, Cpad, int(flipw), ptr(dst), stream())
    return dst


def to_ncdhw(src, C_=None):
    N, T, H, W, ld = src.shape
    C_ = C_ or ld
    dst = torch.empty(N, C_, T, H, W, device=src.device, dtype=torch.float32)
    capi.call("pc_ndhwc_to_ncdhw", ptr(src), ld, N, C_, T * H * W, ptr(dst), stream())
    return dst


def transpose_batched(src, batch, R, Cc, sbs, sld, dst, dbs, dld, accum=False):
    capi.call("pc_transpose_batched", ptr(src), batch, R, Cc, int(sbs), sld, ptr(dst), int(dbs), dld, int(accum), stream())


def em_fwd(x, W, bu, ba, npos, B, C_, state=None):
    """state: optional float32 device tensor of pc_em_state_floats(npos) floats the forward fills for em_bwd."""
    out = torch.empty(npos, C_ * 17, device=x.device, dtype=torch.float32)
    capi.call("pc_em_routing_fwd", ptr(x), ptr(W), ptr(bu), ptr(ba), npos, B, C_, ptr(out), ptr(state), stream())
    return out


def em_state(npos, device):
    return torch.empty(capi.lib().pc_em_state_floats(int(npos)), device=device, dtype=torch.float32)


def em_bwd(x, W, bu, ba, dout, npos, B, C_, dW, dbu, dba, state=None):
    dx = torch.empty(npos, B * 17, device=x.device, dtype=torch.float32)
    ws = torch.empty(capi.lib().pc_em_ws_floats(npos, B, C_), device=x.device, dtype=torch.float32)
    capi.call("pc_em_routing_bwd", ptr(x), ptr(W), ptr(bu), ptr(ba), ptr(dout), npos, B, C_, ptr(dx), ptr(dW), ptr(dbu), ptr(dba), ptr(ws), ptr(state), stream())
    return dx


def class_mask_fwd(caps, Bn, npos, C_, cls, labeled, mode):
    dev = caps.device
    pred = torch.empty(Bn, C_, device=dev); mask = torch.empty(Bn, C_, device=dev)
    masked = torch.empty(Bn, npos, C_ * 16, device=dev)
    capi.call("pc_class_mask_fwd", ptr(caps), Bn, npos, C_, ptr(cls), ptr(labeled), mode, ptr(pred), ptr(mask), ptr(masked), stream())
    return pred, mask, masked


def class_mask_bwd(dmasked, dpred, mask, Bn, npos, C_):
    dcaps = torch.empty(Bn, npos, C_ * 17, device=dmasked.device)
    capi.call("pc_class_mask_bwd", ptr(dmasked), ptr(dpred), ptr(mask), Bn, npos, C_, ptr(dcaps), stream())
    return dcaps


def tapsum_fwd(proj, bias):
    N, T, H, W, _ = proj.shape
    out = torch.empty(N, T, H, W, device=proj.device)
    capi.call("pc_tapsum_fwd", ptr(proj), N, T, H, W, ptr(bias), ptr(out), stream())
    return out


def tapsum_bwd(dout):
    N, T, H, W = dout.shape
    dproj = torch.empty(N, T, H, W, 32, device=dout.device)
    capi.call("pc_tapsum_bwd", ptr(dout), N, T, H, W, ptr(dproj), stream())
    return dproj


def loss_desc(B, T, H, W, bv=False, gv=False, n_frames=3, predict_maps=False, jhmdb=False, lower=None, upper=None,
              bv_wt=0.5, gv_wt=0.5, wt_loc=1.0, wt_cons=1.0, wt_ramp=0.0):
    d = capi.LossDesc()
    d.B, d.T, d.H, d.W = B, T, H, W
    d.bv, d.gv, d.n_frames, d.predict_maps, d.jhmdb = int(bv), int(gv), n_frames, int(predict_maps), int(jhmdb)
    d.lower_thresh = -1.0 if lower is None else lower
    d.upper_thresh = -1.0 if upper is None else upper
    d.bv_wt, d.gv_wt, d.wt_loc, d.wt_cons, d.wt_ramp = bv_wt, gv_wt, wt_loc, wt_cons, wt_ramp
    return d


def consistency_loss(d, output, flip_op, seg, labeled, want_masks=False):
    """output, flip_op, seg: (B,1,8,H,W) fp32 contiguous; labeled int32 (B,).  Returns
    (scalars[8], d_output, d_flip_op, mask_bv, mask_gv)."""
    dev = output.device
    scal = torch.zeros(8, device=dev)
    dO = torch.empty_like(output); dF = torch.empty_like(flip_op)
    mb = torch.zeros_like(output) if want_masks else None
    mg = torch.zeros_like(output) if want_masks else None
    ws = torch.empty(capi.lib().pc_loss_ws_floats(C.byref(d)), device=dev, dtype=torch.float32)
    capi.call("pc_consistency_loss", C.byref(d), ptr(output), ptr(flip_op), ptr(seg), ptr(labeled), ptr(scal), ptr(dO), ptr(dF),
              ptr(mb), ptr(mg), ptr(ws), stream())
    return scal, dO, dF, mb, mg


def var_mask(pred, flip_pred, n_frames=5, use_sig=False):
    B, _, T, H, W = pred.shape
    d = loss_desc(B, T, H, W)
    ws = torch.empty(capi.lib().pc_loss_ws_floats(C.byref(d)), device=pred.device, dtype=torch.float32)
    m = torch.empty(B, 1, T, H, W, device=pred.device)
    capi.call("pc_var_mask", ptr(pred.contiguous()), ptr(flip_pred.contiguous()), B, T, H, W, n_frames, int(use_sig), ptr(m), ptr(ws), stream())
    return m


def grad_mask(pred, lower=None, upper=None):
    B, _, T, H, W = pred.shape
    d = loss_desc(B, T, H, W)
    ws = torch.empty(capi.lib().pc_loss_ws_floats(C.byref(d)), device=pred.device, dtype=torch.float32)
    m = torch.empty(B, T, H, W, device=pred.device)
    capi.call("pc_grad_mask", ptr(pred.contiguous()), B, T, H, W, -1.0 if lower is None else lower, -1.0 if upper is None else upper,
              ptr(m), ptr(ws), stream())
    return m


def spread_loss(x, cls, labeled, m=0.2, wt=1.0, dx=None):
    out = torch.empty(2, device=x.device)
    capi.call("pc_spread_loss", ptr(x), ptr(cls), ptr(labeled), x.shape[0], x.shape[1], m, wt, ptr(out), ptr(dx), stream())
    return out


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-6, gscale=1.0):
    capi.call("pc_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, b1, b2, eps, step, gscale, stream())


def axis_linear(d, x, M, out, bias=None):
    """d: dict with capi.AXIS_FIELDS (see include/picons.h pc_axis_desc)."""
    st = capi.AxisDesc(*[int(d[k]) for k in capi.AXIS_FIELDS])
    capi.call("pc_axis_linear", C.byref(st), ptr(x), ptr(M), ptr(bias), ptr(out), stream())
    return out


def wspec_fwd(w, tw, A, B, KY, KX, U, Ur, out):
    capi.call("pc_wspec_fwd", ptr(w), ptr(tw), A, B, KY, KX, U, Ur, ptr(out), stream())
    return out


def wspec_bwd(dV, tw, A, B, KY, KX, U, Ur, kg):
    capi.call("pc_wspec_bwd", ptr(dV), ptr(tw), A, B, KY, KX, U, Ur, ptr(kg), stream())
    return kg


def wspec_master_fwd(w, tw, Acnt, a0, Atot, B, KY, KX, U, Ur, out_f, out_t):
    capi.call("pc_wspec_master_fwd", ptr(w), ptr(tw), Acnt, a0, Atot, B, KY, KX, U, Ur, ptr(out_f), ptr(out_t), stream())


def wspec_master_bwd(dV, tw, Acnt, a0, Atot, B, KY, KX, U, Ur, dw, accum=False):
    capi.call("pc_wspec_master_bwd", ptr(dV), ptr(tw), Acnt, a0, Atot, B, KY, KX, U, Ur, ptr(dw), int(accum), stream())
    return dw


def tail6_weights(wf, N, Ci, W6f, W6t):
    capi.call("pc_tail6_weights", ptr(wf), N, Ci, ptr(W6f), ptr(W6t), stream())


def tail6_gather(cols, bc, bsm, N, It, Ih, Iw, out):
    capi.call("pc_tail6_gather", ptr(cols), ptr(bc), ptr(bsm), N, It, Ih, Iw, ptr(out), stream())
    return out


def tail6_scatter(dout, N, It, Ih, Iw, dcols):
    capi.call("pc_tail6_scatter", ptr(dout), N, It, Ih, Iw, ptr(dcols), stream())
    return dcols


def tail6_wgrad_map(dW6, N, Ci, Gc):
    capi.call("pc_tail6_wgrad_map", ptr(dW6), N, Ci, ptr(Gc), stream())
    return Gc


def tail6_bias_sums(dout, N, It, Ih, Iw, sums, ws=None):
    """ws (tail6_bias_sums_ws_floats floats): per-block partial rows added in block order instead of fp32 atomics."""
    if ws is None:
        capi.call("pc_tail6_bias_sums", ptr(dout), N, It, Ih, Iw, ptr(sums), stream())
    else:
        assert ws.numel() >= tail6_bias_sums_ws_floats(N, It, Ih, Iw)
        capi.call("pc_tail6_bias_sums_ws", ptr(dout), N, It, Ih, Iw, ptr(sums), ptr(ws), stream())
    return sums


def tail6_bias_sums_ws_floats(N, It, Ih, Iw):
    return int(capi.lib().pc_tail6_bias_sums_ws_floats(N, It, Ih, Iw))


def tail6_wgrad_map_slices(ws, nslices8, N, Ci, Gc):
    """ws: the classes' K-slice workspaces back to back ([z][k < nslices8[z]][N][Ci][128], pc_wgrad_desc.ws_slices); a class's image 0 holds the
    class's in-order sum afterwards."""
    import ctypes
    import numpy as np
    ns = np.ascontiguousarray(nslices8, dtype=np.int32)
    assert ns.shape == (8,) and ws.numel() >= int(ns.sum()) * N * Ci * 128
    capi.call("pc_tail6_wgrad_map_slices", ptr(ws), ns.ctypes.data_as(ctypes.c_void_p), N, Ci, ptr(Gc), stream())
    return Gc


def transpose_multi(jobs):
    """jobs: list of (src, dst, batch, R, C, src_batch_stride, src_ld, dst_batch_stride, dst_ld, accum) with torch tensors
    for src / dst: dst[b][c][r] (+)= src[b][r][c] for all of them in one launch."""
    import numpy as np
    tab = np.zeros(len(jobs), dtype=capi.TJOB_DTYPE)
    for q, job in enumerate(jobs):
        src, dst, batch, R, Cc, sbs, sld, dbs, dld, accum = job[:10]
        nslices, sst = (job[10], job[11]) if len(job) > 10 else (0, 0)       # K-slice images of a weight gradient, added in slice order on the way
        if not (src.is_cuda and dst.is_cuda):
            raise RuntimeError("picons ops need CUDA/HIP tensors (no CPU fallback)")
        tab[q] = (src.data_ptr(), dst.data_ptr(), sbs, dbs, batch, R, Cc, sld, dld, int(accum), int(nslices), 0, int(sst))
    capi.call("pc_transpose_multi", C.c_void_p(tab.ctypes.data), len(jobs), stream())


def _lane_array(side):
    """(c_void_p array, n): torch's current stream as lane 0 + the caller's side streams."""
    hs = [torch.cuda.current_stream().cuda_stream] + [st.cuda_stream for st in (side or ())]
    return (C.c_void_p * len(hs))(*hs), len(hs)


def streams_fanin(target, side=None):
    """`target` (a torch stream) waits for everything enqueued so far on torch's current stream and on the side streams."""
    arr, nl = _lane_array(side)
    capi.call("pc_streams_fanin", C.c_void_p(target.cuda_stream), arr, nl)


def run_ops(ops_np, n=None, side=None):
    """ops_np: numpy array of capi.OP_DTYPE (host memory); replayed by the library.  side: extra
    torch.cuda.Stream objects for the plan's lanes 1.. (None -> everything on the current stream)."""
    n = len(ops_np) if n is None else n
    if not side:
        capi.call("pc_run_ops", C.c_void_p(ops_np.ctypes.data), n, stream())
    else:
        arr, nl = _lane_array(side)
        capi.call("pc_run_ops_lanes", C.c_void_p(ops_np.ctypes.data), n, arr, nl)


def run_ops_timed(ops_np, kind, side=None, defer=False):
    """defer: record the event pairs but do not synchronise; timed_collect() reads them later."""
    ms = C.c_float(0); cnt = C.c_int32(0)
    arr, nl = _lane_array(side)
    capi.call("pc_run_ops_timed", C.c_void_p(ops_np.ctypes.data), len(ops_np), kind, None if defer else C.byref(ms), C.byref(cnt), arr, nl)
    return ms.value, cnt.value


def timed_collect():
    ms = C.c_float(0); cnt = C.c_int32(0)
    capi.call("pc_run_ops_timed_collect", C.byref(ms), C.byref(cnt))
    return ms.value, cnt.value


def seg_frame_counts(logits, gt):
    """logits / gt: contiguous float32 device tensors with the same number of elements, frame-major ([..., H, W] with any
    leading dims flattened to frames).  -> int32 [nframes, 3] = (intersection, union, truth pixels) per frame."""
    if logits.dtype != torch.float32 or gt.dtype != torch.float32 or not logits.is_contiguous() or not gt.is_contiguous():
        raise ValueError("seg_frame_counts: contiguous float32 tensors")
    pix = logits.shape[-1] * logits.shape[-2]
    nframes = logits.numel() // pix
    if gt.numel() != logits.numel():
        raise ValueError("seg_frame_counts: %d logits vs %d truth pixels" % (logits.numel(), gt.numel()))
    counts = torch.empty(nframes, 3, dtype=torch.int32, device=logits.device)
    capi.call("pc_seg_frame_counts", ptr(logits), ptr(gt), nframes, pix, ptr(counts), stream())
    return counts


def map_accumulate(counts, label, frame_hits, video_hits, n_frames, n_vids):
    capi.call("pc_map_accumulate", ptr(counts), counts.shape[0], int(label), frame_hits.shape[0], ptr(frame_hits), ptr(video_hits),
              ptr(n_frames), ptr(n_vids), stream())


def clip_from_u8(video, span, h0, w0, rects, S=224, out=None, ndhwc4=False):
    """video: uint8 device tensor [F,H,W,3]; span: 8 frame ids; rects: int32 device tensor [8,R,4] (x0,x1,y0,y1) or None.
    -> data, aug [3,8,S,S] float32, mask [8,S,S] float32 (pc_clip_from_u8).  out: (data, aug, mask) contiguous float32 device tensors of those
    sizes to write into -- a sample's place in a minibatch staging buffer -- instead of fresh ones.  ndhwc4: data / aug as [8,S,S,4] (r, g, b, 0),
    the layout the network's first conv reads (pc_clip_from_u8_ndhwc4)."""
    if video.dtype != torch.uint8 or video.dim() != 4 or video.shape[3] != 3 or not video.is_contiguous():
        raise ValueError("clip_from_u8: contiguous uint8 [F,H,W,3] frames")
    F, H, W, _ = video.shape
    R = 0 if rects is None else int(rects.shape[1])
    if rects is not None and (rects.dtype != torch.int32 or not rects.is_contiguous() or rects.shape[0] != 8 or rects.shape[2] != 4):
        raise ValueError("clip_from_u8: rects must be contiguous int32 [8,R,4]")
    nch = 4 if ndhwc4 else 3
    if out is None:
        data = torch.empty((8, S, S, 4) if ndhwc4 else (3, 8, S, S), device=video.device); aug = torch.empty_like(data); mask = torch.empty(8, S, S, device=video.device)
    else:
        data, aug, mask = out
        for t, n_ in ((data, nch * 8 * S * S), (aug, nch * 8 * S * S), (mask, 8 * S * S)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n_ or t.device != video.device:
                raise ValueError("clip_from_u8: out tensors must be contiguous float32 device tensors of %d*8*S*S / %d*8*S*S / 8*S*S elements" % (nch, nch))
    sp = (C.c_int32 * 8)(*[int(v) for v in span])
    capi.call("pc_clip_from_u8_ndhwc4" if ndhwc4 else "pc_clip_from_u8", ptr(video), F, H, W, sp, int(h0), int(w0), S, ptr(rects) if R else None, R, ptr(data), ptr(aug), ptr(mask), stream())
    return data, aug, mask


def clip_from_u8_masks(video, span, h0, w0, maskframes, valid, S=224):
    """JHMDB form (pc_clip_from_u8_masks): maskframes uint8 device [F,H,W]; valid: 8 flags.  -> data, aug, mask, mask_cls."""
    if video.dtype != torch.uint8 or video.dim() != 4 or video.shape[3] != 3 or not video.is_contiguous():
        raise ValueError("clip_from_u8_masks: contiguous uint8 [F,H,W,3] frames")
    F, H, W, _ = video.shape
    if maskframes.dtype != torch.uint8 or tuple(maskframes.shape) != (F, H, W) or not maskframes.is_contiguous():
        raise ValueError("clip_from_u8_masks: maskframes must be contiguous uint8 [F,H,W]")
    data = torch.empty(3, 8, S, S, device=video.device); aug = torch.empty_like(data)
    mask = torch.empty(8, S, S, device=video.device); mask_cls = torch.empty_like(mask)
    sp = (C.c_int32 * 8)(*[int(v) for v in span]); va = (C.c_int32 * 8)(*[int(bool(v)) for v in valid])
    capi.call("pc_clip_from_u8_masks", ptr(video), F, H, W, sp, int(h0), int(w0), S, ptr(maskframes), va, ptr(data), ptr(aug), ptr(mask), ptr(mask_cls), stream())
    return data, aug, mask, mask_cls


_RESIZE_TABS = {}


def resize_u8(src, Ho, Wo, interpolation, binarize=False):
    """cv2.resize(src, (Wo, Ho), interpolation) on uint8 device images (pc_resize_u8): src [n,H,W,C] or [n,H,W] contiguous;
    interpolation = cv2's flag value (0 nearest, 1 linear, 3 area).  The coordinate / coefficient table is computed on the
    host by the library (pc_resize_tables) and cached on the device per (interpolation, sizes, device)."""
    if src.dtype != torch.uint8 or not src.is_contiguous() or src.dim() not in (3, 4):
        raise ValueError("resize_u8: contiguous uint8 [n,H,W,C] or [n,H,W]")
    n, H, W = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    Cc = int(src.shape[3]) if src.dim() == 4 else 1
    key = (int(interpolation), H, W, int(Ho), int(Wo), str(src.device))
    tab = _RESIZE_TABS.get(key)
    if tab is None:
        need = capi.lib().pc_resize_tables(int(interpolation), H, W, int(Ho), int(Wo), None, 0)
        if need < 0:
            capi.check(int(need))
        host = np.zeros(int(need), np.int32)
        got = capi.lib().pc_resize_tables(int(interpolation), H, W, int(Ho), int(Wo), C.c_void_p(host.ctypes.data), int(need))
        assert got == need
        tab = _RESIZE_TABS[key] = torch.from_numpy(host).to(src.device)
    dst = torch.empty((n, int(Ho), int(Wo)) + ((Cc,) if src.dim() == 4 else ()), dtype=torch.uint8, device=src.device)
    capi.call("pc_resize_u8", ptr(src), n, H, W, Cc, int(Ho), int(Wo), ptr(tab), int(bool(binarize)), ptr(dst), stream())
    return dst
