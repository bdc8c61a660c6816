"""ctypes binding of libpicons.so (include/picons.h).  No CPU fallback: if the HIP library is
missing or a call fails, this raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PICONS_DIAG_LIB=1: the diagnostic build (make -C csrc diag), which also holds the ablation / stamp variants of the GEMM kernels -- some
# of them WRONG by design.  The product library is built without them and never reads their switches; setting one without the diagnostic
# library is refused here instead of being silently ignored (tools/ablate_conv.py, tools/ablate_wgrad.py, tools/probe_wino.py).
DIAG = os.environ.get("PICONS_DIAG_LIB", "0") not in ("", "0")
DIAG_SWITCHES = ("PICONS_CONV_ABLATE", "PICONS_WGRAD_ABLATE", "PICONS_WINO_VARIANT")
LIB_PATH = os.path.join(_HERE, "libpicons_diag.so" if DIAG else "libpicons.so")
# development only (tools/gpu/*.sh A/B runs): another BUILD of this same library, by file name inside the package directory -- e.g. the
# previous commit's build kept as libpicons_base.so -- so that two builds can be timed in one gpurun call on one box
if os.environ.get("PICONS_LIB_NAME"):
    LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ["PICONS_LIB_NAME"]))
_lib = None

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "Ti", "Hi", "Wi", "Ci", "ldi", "Tq", "Hq", "Wq", "To", "Ho", "Wo", "Co", "ldo")] + \
               [(n, i32 * 3) for n in ("ostr", "ooff", "istr", "ntap", "ioff0", "istep", "wk0", "wkstep")] + \
               [(n, i32) for n in ("KT", "KH", "KW", "ldw", "act", "flags", "wgstride", "bgstride", "act_c0", "groups")]


class WgradDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "Tq", "Hq", "Wq", "Cd", "ldd", "Ts", "Hs", "Ws", "Cs", "lds")] + \
               [(n, i32 * 3) for n in ("istr", "ntap", "ioff0", "istep", "wk0")] + [(n, i32) for n in ("KT", "KH", "KW", "splitk", "nbatch", "dbstride", "sbstride", "gbstride", "Td", "Hd", "Wd")] + [("doff", i32 * 3), ("flags", i32), ("ws_slices", i32)]


AXIS_FIELDS = ["R", "I", "O", "C", "in_split", "out_split", "act", "act_c0", "accum", "in_sr", "in_hi", "in_lo", "out_sr", "out_hi", "out_lo"]


class WinoDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "T", "H", "W", "Ci", "ldi", "Co", "ldo", "KT", "act", "flags", "Ti", "ta", "tc", "tden", "m")]


class AxisDesc(C.Structure):
    _fields_ = [(n, i32) for n in AXIS_FIELDS]


class PoolDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("N", "Ti", "Hi", "Wi", "C", "ldi", "To", "Ho", "Wo", "ldo")] + \
               [(n, i32 * 3) for n in ("k", "s", "padf")]


class LossDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "T", "H", "W", "bv", "gv", "n_frames", "predict_maps", "jhmdb")] + \
               [(n, f32) for n in ("lower_thresh", "upper_thresh", "bv_wt", "gv_wt", "wt_loc", "wt_cons", "wt_ramp")]


# numpy mirror of struct pc_op (kind, i[48], f[8], lane, p[12], l[4])
OP_DTYPE = np.dtype([("kind", np.int32), ("i", np.int32, 48), ("f", np.float32, 8), ("lane", np.int32),
                     ("p", np.uint64, 12), ("l", np.int64, 4)], align=False)

(OP_CONV, OP_WGRAD, OP_BN_FINALIZE, OP_BN_APPLY, OP_BN_EVAL_STAT, OP_BN_BWD, OP_POOL_FWD, OP_POOL_BWD, OP_CHSCALE,
 OP_ACT_BWD, OP_TO_NDHWC, OP_TO_NCDHW, OP_TRANSPOSE, OP_FILL, OP_AXPY, OP_EM_FWD, OP_EM_BWD, OP_CMASK_FWD, OP_CMASK_BWD,
 OP_TAPSUM_FWD, OP_TAPSUM_BWD, OP_LOSS, OP_SPREAD, OP_ADAM, OP_TAIL_COMBINE, OP_TAIL_COLSUM, OP_TAIL_GRADS, OP_COL2IM,
 OP_AXIS, OP_WSPEC_FWD, OP_WSPEC_BWD, OP_WSPEC_MASTER_FWD, OP_WSPEC_MASTER_BWD, OP_TAIL6_WEIGHTS, OP_TAIL6_GATHER, OP_TAIL6_SCATTER, OP_TAIL6_WGRAD_MAP, OP_TAIL6_BIAS_SUMS,
 OP_TRANSPOSE_MULTI, OP_FORK, OP_JOIN, OP_WGRAD_MULTI, OP_WINO_CONV, OP_WINO_WEIGHTS, OP_CONV_X6, OP_SPLIT_PLANES, OP_SPLIT_PLANES_MULTI, OP_WSPEC_MASTER_PLANES, OP_BN_FIN_APPLY, OP_WGRAD_FOLD) = range(1, 51)
MAX_LANES = 8

# numpy mirror of struct pc_wgrad_job (pc_wgrad_desc = 42 int32, then D, S, g)
WJOB_DTYPE = np.dtype([("d", np.int32, 42), ("D", np.uint64), ("S", np.uint64), ("g", np.uint64)], align=False)

# numpy mirror of struct pc_transpose_job
TJOB_DTYPE = np.dtype([("src", np.uint64), ("dst", np.uint64), ("sbs", np.int64), ("dbs", np.int64), ("batch", np.int32), ("R", np.int32),
                       ("C", np.int32), ("sld", np.int32), ("dld", np.int32), ("accum", np.int32), ("nslices", np.int32), ("reserved", np.int32),
                       ("sst", np.int64)], align=False)

# numpy mirror of struct pc_split_job
SJOB_DTYPE = np.dtype([("src", np.uint64), ("planes", np.uint64), ("n", np.int64), ("pstride", np.int64)], align=False)

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
F_ACCUM, F_BIAS, F_CSCALE, F_BNPART, F_NFAST, F_TOUT, F_CI3, F_X6, F_STRIPS = 1, 2, 4, 8, 16, 32, 64, 128, 256
WG_CS3, WG_X6 = 1, 2

ABI_VERSION = 102          # PC_VERSION of include/picons.h

_SIGS = {
    "pc_version": (i32, []),
    "pc_last_error": (C.c_char_p, []),
    "pc_conv_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]),
    "pc_conv_bnpart_rows": (i32, [C.POINTER(ConvDesc)]),
    "pc_conv_fwd_x6": (i32, [C.POINTER(ConvDesc), vp, vp, i64, vp, vp, vp, vp, vp]),
    "pc_conv_fwd_x6_ws": (i32, [C.POINTER(ConvDesc), vp, vp, i64, vp, vp, vp, vp, vp, i64, vp]),
    "pc_conv_x6_ws_floats": (i64, [C.POINTER(ConvDesc)]),
    "pc_conv_x6_ok": (i32, [C.POINTER(ConvDesc)]),
    "pc_split_planes": (i32, [vp, vp, i64, i64, vp]),
    "pc_split_planes_multi": (i32, [vp, i32, vp]),
    "pc_conv_work": (i32, [C.POINTER(ConvDesc), i32, i32, C.POINTER(C.c_double)]),
    "pc_conv_wgrad": (i32, [C.POINTER(WgradDesc), vp, vp, vp, vp]),
    "pc_conv_wgrad_multi": (i32, [vp, i32, vp]),
    "pc_wgrad_slices": (i32, [C.POINTER(WgradDesc)]),
    "pc_wgrad_uses_x6": (i32, [C.POINTER(WgradDesc)]),
    "pc_wgrad_fold_group": (i32, []),
    "pc_wgrad_fold": (i32, [vp, i64, i32, vp]),
    "pc_wgrad_work": (i32, [C.POINTER(WgradDesc), i32, i32, C.POINTER(C.c_double)]),
    "pc_wino_u_floats": (i64, [i32, i32, i32]),
    "pc_wino_weights": (i32, [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
    "pc_wino_conv": (i32, [C.POINTER(WinoDesc), vp, vp, vp, vp, vp, vp]),
    "pc_wino4_u_floats": (i64, [i32, i32, i32]),
    "pc_wino4_weights": (i32, [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
    "pc_wino_bnpart_rows": (i32, [C.POINTER(WinoDesc)]),
    "pc_wino_work": (i32, [C.POINTER(WinoDesc), C.POINTER(C.c_double)]),
    "pc_bn_finalize": (i32, [vp, i32, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp]),
    "pc_bn_finalize_ws": (i32, [vp, i32, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp]),
    "pc_bn_finalize_ws_floats": (i64, [i32, i32, i32]),
    "pc_bn_finalize_apply_ok": (i32, [i32, i32]),
    "pc_bn_finalize_apply": (i32, [vp, i32, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp, i32, i64, vp, i32, i32, vp]),
    "pc_bn_apply": (i32, [vp, i32, vp, i32, i64, i32, vp, i32, i32, vp]),
    "pc_bn_eval_stat": (i32, [vp, vp, vp, vp, f32, i32, vp, vp]),
    "pc_bn_bwd_ws_floats": (i64, [i64, i32, i32]),
    "pc_bn_bwd": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, i32, vp, i32, vp, vp, i32, vp, vp]),
    "pc_maxpool_fwd": (i32, [C.POINTER(PoolDesc), vp, vp, vp, vp]),
    "pc_maxpool_bwd": (i32, [C.POINTER(PoolDesc), vp, vp, vp, i32, vp]),
    "pc_channel_scale": (i32, [vp, i32, vp, i32, i64, i32, vp, i32, i32, vp]),
    "pc_act_bwd_ws_floats": (i64, [i64, i32]),
    "pc_act_bwd": (i32, [vp, i32, vp, i32, i32, i32, i64, vp, i32, vp, i32, vp, vp]),
    "pc_ncdhw_to_ndhwc": (i32, [vp, i32, i32, i32, i64, i32, i32, i32, vp, vp]),
    "pc_ndhwc_to_ncdhw": (i32, [vp, i32, i32, i32, i64, vp, vp]),
    "pc_transpose_batched": (i32, [vp, i32, i32, i32, i64, i32, vp, i64, i32, i32, vp]),
    "pc_col2im": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "pc_seg_frame_counts": (i32, [vp, vp, i64, i64, vp, vp]),
    "pc_map_accumulate": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, vp]),
    "pc_clip_from_u8": (i32, [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "pc_clip_from_u8_ndhwc4": (i32, [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "pc_clip_from_u8_masks": (i32, [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, i32, i32, vp, C.POINTER(C.c_int32), vp, vp, vp, vp, vp]),
    "pc_resize_tables": (i64, [i32, i32, i32, i32, i32, vp, i64]),
    "pc_resize_u8": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "pc_fill": (i32, [vp, i64, f32, vp]),
    "pc_axpy": (i32, [vp, vp, i64, f32, vp]),
    "pc_em_ws_floats": (i64, [i32, i32, i32]),
    "pc_em_state_floats": (i64, [i32]),
    "pc_em_routing_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "pc_em_routing_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "pc_class_mask_fwd": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp]),
    "pc_class_mask_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "pc_tapsum_fwd": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "pc_tapsum_bwd": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "pc_loss_ws_floats": (i64, [C.POINTER(LossDesc)]),
    "pc_consistency_loss": (i32, [C.POINTER(LossDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "pc_var_mask": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "pc_grad_mask": (i32, [vp, i32, i32, i32, i32, f32, f32, vp, vp, vp]),
    "pc_spread_loss": (i32, [vp, vp, vp, i32, i32, f32, f32, vp, vp, vp]),
    "pc_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp]),
    "pc_tail_combine": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "pc_tail_colsum": (i32, [vp, i32, i64, vp, vp]),
    "pc_tail_grads": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "pc_tail_grads_ws_floats": (i64, [i32, i32, i32]),
    "pc_tail_grads_ws": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "pc_axis_linear": (i32, [vp, vp, vp, vp, vp, vp]),
    "pc_wspec_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "pc_wspec_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "pc_wspec_master_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "pc_wspec_master_planes": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i64, vp]),
    "pc_wspec_master_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "pc_tail6_weights": (i32, [vp, i32, i32, vp, vp, vp]),
    "pc_tail6_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "pc_tail6_scatter": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "pc_tail6_wgrad_map": (i32, [vp, i32, i32, vp, vp]),
    "pc_tail6_bias_sums": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "pc_tail6_wgrad_map_slices": (i32, [vp, vp, i32, i32, vp, vp]),
    "pc_tail6_bias_sums_ws_floats": (i64, [i32, i32, i32, i32]),
    "pc_tail6_bias_sums_ws": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "pc_transpose_multi": (i32, [vp, i32, vp]),
    "pc_run_ops": (i32, [vp, i32, vp]),
    "pc_run_ops_lanes": (i32, [vp, i32, vp, i32]),
    "pc_streams_fanin": (i32, [vp, vp, i32]),
    "pc_release_thread_events": (i32, []),
    "pc_run_ops_timed": (i32, [vp, i32, i32, C.POINTER(f32), C.POINTER(i32), vp, i32]),
    "pc_run_ops_timed_collect": (i32, [C.POINTER(f32), C.POINTER(i32)]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Load libpicons.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        stray = [k for k in DIAG_SWITCHES if os.environ.get(k, "0") not in ("", "0")]
        if stray and not DIAG:
            raise RuntimeError("%s set, but these switches exist only in the diagnostic build of the library (results can be wrong by design): "
                               "`make -C pi-consistency-activity-detection_amd/csrc diag` and PICONS_DIAG_LIB=1, or unset them" % ", ".join(stray))
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libpicons.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C pi-consistency-activity-detection_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.pc_version() != ABI_VERSION:
            raise RuntimeError("libpicons.so is ABI version %d, this package mirrors include/picons.h version %d: rebuild it "
                               "(`make -C pi-consistency-activity-detection_amd/csrc`)" % (L.pc_version(), ABI_VERSION))
        assert C.sizeof(ConvDesc) == 48 * 4 and OP_DTYPE.itemsize == 4 + 192 + 32 + 4 + 96 + 32
        assert C.sizeof(WgradDesc) == 168 and WJOB_DTYPE.itemsize == 192
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libpicons error %d: %s" % (rc, lib().pc_last_error().decode()))


def call(name, *args):
    check(getattr(lib(), name)(*args))
