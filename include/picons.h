/* picons.h - C ABI of libpicons.so: the MI355X (gfx950) hot path of the semi-supervised
 * video action-detection train step (AKASH2907/pi-consistency-activity-detection).
 *
 * The reference has NO native/FFI layer: its boundary is the Python module
 * surface.  This ABI sits underneath that surface; each entry cites the reference call site
 * (an ATen op launched from Python) it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers + POD descriptors, no torch types.  All tensor pointers are
 *    DEVICE pointers to fp32 unless stated.  Activations are NDHWC (channels innermost) with
 *    an explicit channel stride `ld*` so a kernel can read/write a channel slice of a wider
 *    tensor (torch.cat becomes free); the pointer passed already includes the channel offset.
 *  - Ownership: the caller owns every buffer incl. workspaces; the library allocates nothing,
 *    keeps no pointers between calls, and only enqueues work on the hipStream_t it is given
 *    (never the null stream implicitly, no device sync).  Entries are re-entrant.
 *  - Errors: int return, 0 = ok, negative = PC_E_*; message via pc_last_error() (thread-local).
 *    Nothing throws across the ABI or calls exit().
 */
#ifndef PICONS_H
#define PICONS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pc_stream;            /* hipStream_t */

#define PC_OK            0
#define PC_E_ARG        -1          /* bad descriptor / unsupported shape */
#define PC_E_LAUNCH     -2          /* hip launch error */
#define PC_E_NODEVICE   -3

#define PC_ACT_NONE 0
#define PC_ACT_RELU 1
#define PC_ACT_SIGMOID 2

#define PC_F_ACCUM   1              /* out += result (dgrad into a tensor with several consumers) */
#define PC_F_BIAS    2
#define PC_F_CSCALE  4              /* multiply by cscale[n][co] (Dropout3d draw, capsules_ucf101.py:428,507) */
#define PC_F_BNPART  8              /* emit per-block BatchNorm partial sums (sum, sumsq) */
#define PC_F_NFAST   16             /* GEMM rows ordered (t,h,w,n) instead of (n,t,h,w): a 128-row tile is a narrow spatial
                                       patch of all samples, so taps that are padding for the whole tile are skipped
                                       (9x9 ConvTranspose / PrimaryCaps dgrad: 784 gathered vs 400 real positions); with groups the
                                       order holds inside each group */
#define PC_F_TOUT    32             /* channel-major output: out[(n * ldo + co) * (To*Ho*Wo) + position] instead of
                                       out[(n*To*Ho*Wo + position) * ldo + co].  Plain launches only (no bias / activation / cscale /
                                       accumulate / BN partials): the merged tail's column GEMMs, whose gather then streams every
                                       column entry once instead of pulling 4 bytes out of 27 different 512-byte rows */
#define PC_F_CI3     64             /* Ci == 4 whose 4th channel is padding (the RGB clip, 3 channels in 16-byte pieces): that
                                     * channel is taken as zero whatever it holds; the LDS-DMA stem kernel skips its MFMAs */

#define PC_F_STRIPS  256            /* pc_wino_conv, F(2x2, 3x3) only: blocks of two 2 x 14-tile strips taken from pairs of planes (n, n + 1) instead of one
                                     * rectangle of 64 tiles per block -- 56 instead of 49 of a block's 64 tile slots on 28 x 28 frames; needs a tile grid
                                     * a multiple of 14 wide, an even number of tile rows and N % 4 == 0, and is ignored otherwise.  Bit-identical
                                     * results; BatchNorm partial rows then number pc_wino_bnpart_rows(d) for THIS flag setting */
#define PC_F_X6      128            /* the launch multiplies on the bf16 matrix cores: fp32 operands as exact sums of three bf16 values, six
                                     * products, fp32 accumulate (pc_conv_fwd_x6; the caller holds the weights as bf16 planes) */

/* ABI version of this header: bumped whenever a struct in it grows or an op's operands change (101: pc_wino_desc.m, PC_OP_BN_FIN_APPLY,
 * pc_wgrad_desc.ws_slices, pc_transpose_job.nslices / slice_stride, pc_wgrad_slices; 102: the workspace operands of PC_OP_TAIL6_WGRAD_MAP / PC_OP_TAIL6_BIAS_SUMS /
 * PC_OP_TAIL_GRADS).  Descriptors must be zero-initialised by the caller:
 * fields added later read as "old behaviour" when 0.  pc_version() returns the value the library was built with; the Python host
 * (capi.lib()) refuses a library whose version differs from the header it mirrors. */
#define PC_VERSION 102
int         pc_version(void);
const char* pc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Generalised gather-GEMM convolution (fp32 MFMA v_mfma_f32_32x32x2_f32, LDS-tiled).
 *   out[n, q*ostr+ooff, co] = act( bias[co] + sum_{a,b,c} sum_ci
 *        in[n, q*istr + ioff0 + (a,b,c)*istep, ci] * w[co][wtap(a,b,c)][ci] ) * cscale[n][co]
 *   wtap = ((wk0_t + a*wkstep_t)*KH + wk0_h + b*wkstep_h)*KW + wk0_w + c*wkstep_w
 * Weight layout [Co][KT*KH*KW][ldw] (taps then channels innermost).  One descriptor covers
 *   nn.Conv3d/Conv2d forward incl. TF-SAME padding folded in (pytorch_i3d.py:112-115; capsules_ucf101.py:44-45,490,497,501),
 *   conv dgrad (stride 1: mirrored taps; stride s: one launch per output-parity class),
 *   nn.ConvTranspose2d/3d forward as sub-pixel parity classes (capsules_ucf101.py:486,495,499,504),
 *   and ConvTranspose dgrad (= strided conv).
 * Requires Ci % 4 == 0, ldi % 4 == 0, ldw % 4 == 0 (float4 loads). */
typedef struct pc_conv_desc {
    int32_t N;
    int32_t Ti, Hi, Wi, Ci, ldi;
    int32_t Tq, Hq, Wq;
    int32_t To, Ho, Wo, Co, ldo;
    int32_t ostr[3], ooff[3];
    int32_t istr[3], ntap[3], ioff0[3], istep[3];
    int32_t wk0[3], wkstep[3];
    int32_t KT, KH, KW, ldw;
    int32_t act, flags;
    int32_t wgstride, bgstride;     /* per-group weight / bias offsets (floats): group g uses w + g*wgstride, bias + g*bgstride
                                       (per-sample collapsed decoder-tail weights); 0 = shared */
    int32_t act_c0;                 /* activation applies to output channels >= act_c0 (merged pose|activation conv) */
    int32_t groups;                 /* >=1: rows are tiled per batch group (N/groups samples each) so BatchNorm
                                       partials never straddle the two forward passes of one step */
} pc_conv_desc;

/* bnpart: [pc_conv_bnpart_rows(d)][2][Co] partial (sum, sumsq) over output rows, or NULL.
 * Size limit: the LDS-DMA gather addresses `in` and `w` with 32-bit byte offsets from a wave-uniform base, so each must be smaller than
 * 4 GiB (0xff000000 bytes; PC_E_ARG otherwise -- the largest activation at bs = 8 is 0.8 GB).  Same for D / S of pc_conv_wgrad and for one
 * (H, W, ldi) frame of pc_wino_conv. */
int pc_conv_fwd(const pc_conv_desc* d, const float* in, const float* w, const float* bias,
                const float* cscale, float* out, float* bnpart, pc_stream s);
int pc_conv_bnpart_rows(const pc_conv_desc* d);

/* The same convolution with the fp32 multiplications carried out on the bf16 matrix cores (csrc/conv_x6.hip): every fp32 operand is split
 * exactly into three bf16 values h + m + l and the six products of weight >= 2^-16 are accumulated in fp32 -- closer to an fp64 result than
 * an fp32 FMA chain (tests/test_x6_gpu.py), 1.5 - 1.8x the rate of v_mfma_f32_32x32x2_f32.  `in` stays fp32 (split in registers); the weights
 * come as three bf16 planes, each in the layout of `w` above ([Co][KT*KH*KW][ldw], group g at + g * wgstride elements), plane p at
 * wplanes + p * plane_stride elements -- pc_split_planes makes them from the fp32 layout.  d->flags must carry PC_F_X6 and
 * pc_conv_x6_ok(d) must hold (Ci % 32 == 0, ldw % 8 == 0, <= 10 taps per dimension); everything else as pc_conv_fwd. */
int pc_conv_fwd_x6(const pc_conv_desc* d, const float* in, const uint16_t* wplanes, int64_t plane_stride, const float* bias,
                   const float* cscale, float* out, float* bnpart, pc_stream s);
/* With a workspace the launch may split the tiles of its last, partly filled round of resident blocks (or all tiles, when it does not fill
 * one round) into K slices: one block per slice leaves its partial sums in `ws`, the last block to arrive at a tile adds them in slice order
 * and runs the epilogue -- same results whichever block that is (run-to-run bit-identical), every epilogue supported.  pc_conv_x6_ws_floats(d)
 * = floats the launch wants (0: it would not split).  The LAST ceil(rem / 4) * 4 floats of that size are per-tile counters: they must be ZERO
 * before the first launch that uses the workspace, every launch leaves them zero.  ws == NULL: pc_conv_fwd_x6. */
int64_t pc_conv_x6_ws_floats(const pc_conv_desc* d);
int pc_conv_fwd_x6_ws(const pc_conv_desc* d, const float* in, const uint16_t* wplanes, int64_t plane_stride, const float* bias,
                      const float* cscale, float* out, float* bnpart, float* ws, int64_t ws_floats, pc_stream s);
int pc_conv_x6_ok(const pc_conv_desc* d);       /* host-only advice for a planner: 1 if the descriptor (with PC_F_X6) can take the bf16-split kernel AND is
                                                  * large enough to gain from it (small launches are faster on pc_conv_fwd) */
/* planes[p * plane_stride + i] = p-th bf16 term of src[i] (p = 0, 1, 2: h = bf16(x), m = bf16(x - h), l = x - h - m, round to nearest; h + m + l == src[i] exactly for 2^-110 <= |x| < 2^128);
 * n and plane_stride multiples of 4 */
int pc_split_planes(const float* src, uint16_t* planes, int64_t n, int64_t plane_stride, pc_stream s);
/* the same for many buffers in one launch (the per-step weight planes); `jobs` is HOST memory */
typedef struct pc_split_job {
    uint64_t src, planes;           /* device pointers */
    int64_t  n, plane_stride;
} pc_split_job;
int pc_split_planes_multi(const pc_split_job* jobs, int njobs, pc_stream s);
/* Host-only work accounting of one pc_conv_fwd launch (no GPU call; measurement support for bench.py / tools/launch_table.py,
 * no reference counterpart).  Walks the launch's tiles in the kernel's own row order and counts the K loop each block really
 * runs (taps that are padding for every row of a tile are skipped by the kernel).  out[7]:
 *   [0] multiply-accumulates ISSUED to the matrix cores (whole tiles: what an MFMA instruction counter sees),
 *   [1] EXECUTED on real outputs (real rows x real columns x the tile's K; padding taps inside the tile's tap box included),
 *   [2] VALID (taps that read inside the volume, real channels only),  [3] blocks,  [4] BM,  [5] BN,
 *   [6] 1 = LDS-DMA kernel with the per-tile tap box, 0 = register-staged kernel (flattened K).
 * ci_real / co_real: channels that are not padding (0 = Ci / Co). */
int pc_conv_work(const pc_conv_desc* d, int ci_real, int co_real, double* out);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) form of the stride-1, "same"-padded KT x 3 x 3 convolutions (KT = 3 with temporal padding 1, or 1):
 *   out[n,t,h,w,co] = act( bias[co] + sum_{kt,kh,kw,ci} in[n, t+kt-KT/2, h+kh-1, w+kw-1, ci] * g[co][kt][kh][kw][ci] )   (zero padding)
 * 2.25x fewer multiply-accumulates than pc_conv_fwd for the same result (equal in exact arithmetic; fp32 rounding differs at the
 * 1e-6 level).  Used for the decoder's skip convs and Conv3d_2c and for their input gradients (nn.Conv3d forward / ATen
 * convolution_backward's input branch: capsules_ucf101.py:497,501; pytorch_i3d.py:236-238).  H, W even; Ci % 8 == 0.
 * The weights are passed in the transform domain: pc_wino_weights builds U (pc_wino_u_floats(O, I, KT) floats) from any strided
 * weight tensor w[o*sO + ((kt*3+kh)*3+kw)*sT + i*sI] -- the master OIDHW tensor directly (sO = I*KT*9, sT = 1, sI = KT*9), or, with
 * flip = 1 and O / I exchanged (sO = KT*9, sI = Cout*KT*9), the mirrored / transposed weights of the input gradient.
 * flags: PC_F_BIAS, PC_F_ACCUM, PC_F_BNPART (partials [pc_wino_bnpart_rows(d)][2][Co]; rows of one sample are consecutive, samples
 * in order, so the rows of a batch group are one contiguous range). */
typedef struct pc_wino_desc {
    int32_t N, T, H, W;             /* output frames / positions (H, W: stride 1, same padding, so input = output size) */
    int32_t Ci, ldi, Co, ldo;
    int32_t KT;                     /* temporal taps: 3 or 1 */
    int32_t act, flags;             /* PC_ACT_NONE / PC_ACT_RELU */
    int32_t Ti;                     /* input frames */
    int32_t ta, tc, tden;           /* tap kt of output frame t reads input frame (t*ta + kt + tc) / tden when that is an integer in
                                     * [0, Ti): forward with temporal stride s and front padding p: (s, -p, 1); input gradient of that
                                     * layer (weights mirrored): (1, p - (KT-1), s).  Stride 1, padding KT/2: (1, -(KT/2), 1) both ways */
    int32_t m;                      /* output tile edge: 2 (or 0) = F(2x2, 3x3); 4 = F(4x4, 3x3): 4x fewer multiply-accumulates than the direct
                                     * form for ~3x the rounding error of F(2x2, 3x3); H, W multiples of 4, Ci % 8 == 0, and U
                                     * from pc_wino4_weights (pc_wino4_u_floats floats) */
} pc_wino_desc;
int64_t pc_wino_u_floats(int O, int I, int KT);
int pc_wino_weights(const float* w, int64_t sO, int64_t sT, int64_t sI, int O, int I, int KT, int flip, float* U, pc_stream s);
int pc_wino_conv(const pc_wino_desc* d, const float* in, const float* U, const float* bias, float* out, float* bnpart, pc_stream s);
/* transform-domain weights of the m = 4 form (same addressing of w as pc_wino_weights) */
int64_t pc_wino4_u_floats(int O, int I, int KT);
int pc_wino4_weights(const float* w, int64_t sO, int64_t sT, int64_t sI, int O, int I, int KT, int flip, float* U, pc_stream s);
int pc_wino_bnpart_rows(const pc_wino_desc* d);
/* host-only work accounting (see pc_conv_work): out[0] issued, out[1] executed multiply-accumulates, out[2] blocks */
int pc_wino_work(const pc_wino_desc* d, double* out);

/* Weight gradient:  g[m][ (a,b,c) , cs ] += sum_{n,q} D[n,q,m] * S[n, q*istr+ioff0+(a,b,c)*istep, cs]
 * D dense over the lattice (Tq,Hq,Wq), S gathered.  g layout [Cd][KT*KH*KW][Cs], tap (a,b,c) -> wk0+(a,b,c),
 * accumulated with fp32 atomics (g must be initialised).  Conv3d wgrad: D=dY, S=X;
 * ConvTranspose wgrad: D=X, S=dOut.  (replaces ATen convolution_backward's weight branch.) */
typedef struct pc_wgrad_desc {
    int32_t N;
    int32_t Tq, Hq, Wq, Cd, ldd;
    int32_t Ts, Hs, Ws, Cs, lds;
    int32_t istr[3], ntap[3], ioff0[3], istep[3];
    int32_t wk0[3];                 /* first weight tap per dim (taps that only ever see padding are trimmed by the host) */
    int32_t KT, KH, KW;             /* full weight tap extents: g is [Cd][KT*KH*KW][Cs] */
    int32_t splitk;                 /* 0 = choose; -1 = one slice written with plain stores (g need not be initialised;
                                     * only valid when no tap is trimmed: ntap == (KT,KH,KW)) */
    int32_t nbatch;                 /* 0/1 = one problem; > 1: that many independent problems in ONE launch (blockIdx.z =
                                     * problem * slices + slice), D, S and g advanced by the strides below (floats) */
    int32_t dbstride, sbstride, gbstride;
    int32_t Td, Hd, Wd, doff[3];    /* Td > 0: D is not dense but the sub-lattice (Tq,Hq,Wq) starting at doff of a
                                     * [N][Td][Hd][Wd][ldd] tensor (with nbatch > 1: per problem, N = 1 each) */
    int32_t flags;                  /* PC_WG_CS3: Cs == 4 whose 4th channel is padding -- g[..][3] may be left untouched */
    int32_t ws_slices;              /* 0: K slices are combined with fp32 atomics in g (arrival order).  n > 0: g is a WORKSPACE of n images of the
                                     * gradient, each Cd*KT*KH*KW*Cs floats (nbatch > 1: gbstride*nbatch) -- slice k leaves its partial sums in image k
                                     * with plain stores, n >= pc_wgrad_slices(d); the consumer adds the images in slice order
                                     * (pc_transpose_job.nslices), so the gradient is bit-identical from run to run.  The workspace must be
                                     * zero before its FIRST use (trimmed taps and empty slices are never written) and needs no fill after. */
} pc_wgrad_desc;
#define PC_WG_CS3    1
#define PC_WG_X6     2              /* the row-segment (3 taps along w, padding 1), generic split-K and -- together with PC_WG_CS3 -- stem kernels multiply
                                     * on the bf16 matrix cores: both operands split into three bf16 terms in registers, six products, fp32
                                     * accumulate (as pc_conv_fwd_x6); the 9-tap spectral route ignores it (pc_wgrad_uses_x6) */
int pc_conv_wgrad(const pc_wgrad_desc* d, const float* D, const float* S, float* g, pc_stream s);
/* K slices the launch(es) of pc_conv_wgrad make for this problem (host-only, no GPU call; ws_slices is ignored): the number of workspace
 * images a caller that sets ws_slices has to provide.  -1 on a bad descriptor. */
int pc_wgrad_slices(const pc_wgrad_desc* d);
/* Folds the K-slice images of a workspace in place, for launches with many slices: image g*G of every group of G = pc_wgrad_fold_group()
 * consecutive images becomes the sum of its group (slice order); the consumer then adds ceil(nslices / G) images G*image_floats apart. */
int pc_wgrad_fold_group(void);
int pc_wgrad_fold(float* ws, int64_t image_floats, int nslices, pc_stream s);
/* Host-only work accounting of one pc_conv_wgrad launch (no GPU call; see pc_conv_work).  out[5]: multiply-accumulates ISSUED to
 * the matrix cores, EXECUTED on real rows x columns, VALID (non-padding source positions), and the kernel family the problem is
 * routed to (0 stem, 1 row-segment with 3 taps, 2 row-segment with 9 taps, 3 generic split-K), kernel launches the call makes.
 * cd_real / cs_real: 0 = Cd / Cs. */
int pc_wgrad_work(const pc_wgrad_desc* d, int cd_real, int cs_real, double* out);
/* 1 if the problem's launch multiplies on the bf16 matrix cores (PC_WG_X6 on the row-segment, generic and -- with PC_WG_CS3 -- stem routes), 0 if on
 * fp32 MFMA (host-only; bench.py's two weight-gradient roofline legs and pc_run_ops_timed's sub-filter split the family by it) */
int pc_wgrad_uses_x6(const pc_wgrad_desc* d);
/* Several weight gradients in one call (the wgrads of one Inception module; the eight position classes of the merged tail):
 * the problems the generic split-K kernel would take share ONE grid, each with a range of blocks in proportion to its work; the
 * others (stem, row-segment, long-K shapes) get their usual launch.  Same results as njobs pc_conv_wgrad calls up to the order
 * of the fp32 atomic sums.  `jobs` is host memory. */
typedef struct pc_wgrad_job {
    pc_wgrad_desc d;
    const float* D; const float* S; float* g;
} pc_wgrad_job;
int pc_conv_wgrad_multi(const pc_wgrad_job* jobs, int njobs, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * BatchNorm3d(train) + ReLU (pytorch_i3d.py:116-119; eps 1e-3, momentum 0.01 at :80).
 * `groups`: the two forward passes of one step (main_ucf101.py:85-86) run as one batch whose
 * rows split into `groups` equal row ranges with separate batch statistics; running stats
 * are updated once per group, in order (as the reference's two sequential forwards do). */
/* partials [groups][nparts_per_group][2][C] (sum, sumsq from pc_conv_fwd) -> stat[groups][4][C] =
 * mean, invstd, scale=gamma*invstd, shift=beta-mean*scale; running stats updated if non-NULL. */
int pc_bn_finalize(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, float* stat, pc_stream s);
/* The same with a workspace of pc_bn_finalize_ws_floats(...) floats (0 for fewer than 512 partial rows per group): layers with
 * thousands of partial rows (the stem: 2 x 6272) are reduced in two stages -- 32 blocks per 16 channels and group, then the finalize
 * over their double-precision partial rows -- instead of by one block per 16 channels (85 -> 15 us on the dependency chain).  ws NULL
 * = the one-stage form.  Fixed summation order either way. */
int64_t pc_bn_finalize_ws_floats(int nparts_per_group, int groups, int C);
int pc_bn_finalize_ws(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group,
                      const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* stat, float* ws, pc_stream s);
/* y[r][c] = relu?(z[r][c]*scale[g(r)][c]+shift[g(r)][c]) */
int pc_bn_apply(const float* z, int ldz, const float* stat, int C, int64_t rows, int groups, float* y,
                int ldy, int relu, pc_stream s);
/* eval mode: stat[4][C] from running stats */
/* pc_bn_finalize + pc_bn_apply in ONE launch, for layers with at most 256 partial rows per batch group (pc_bn_finalize_apply_ok): every block
 * of the apply kernel reduces the partial rows of its own 64 channels (doubles, fixed order) and streams its rows; row block 0 of group 0 writes
 * `stat` and the running statistics (groups in order).  Round 5: one dispatch fewer per BatchNorm site on the step's dependency chain
 * (/root/reference/models/pytorch_i3d.py:116-119).  Arguments as the two calls it replaces; rows = all rows of z (every group). */
int pc_bn_finalize_apply_ok(int nparts_per_group, int C);
int pc_bn_finalize_apply(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group, const float* gamma, const float* beta,
                         float eps, float momentum, float* running_mean, float* running_var, float* stat, const float* z, int ldz, int64_t rows,
                         float* y, int ldy, int relu, pc_stream s);
int pc_bn_eval_stat(const float* gamma, const float* beta, const float* running_mean,
                    const float* running_var, float eps, int C, float* stat, pc_stream s);
/* backward: dy (grad after ReLU, row stride lddy), z -> dz; dgamma/dbeta (+)= if accum.
 * ws: >= pc_bn_bwd_ws_floats(rows,C,groups) floats.  relu: bit 0 = the layer has a ReLU; bit 1 = the finalize folded into the apply kernel
 * (two launches instead of three; measured: no gain, the planner never sets it by itself -- switches.py PICONS_BN_FUSED). */
int64_t pc_bn_bwd_ws_floats(int64_t rows, int C, int groups);
int pc_bn_bwd(const float* dy, int lddy, const float* z, int ldz, const float* stat, int C,
              int64_t rows, int groups, int relu, float* dz, int lddz, float* dgamma, float* dbeta,
              int accum, float* ws, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * MaxPool3d with TF-SAME zero padding (pytorch_i3d.py:21-45); argmax kept for the backward. */
typedef struct pc_pool_desc {
    int32_t N, Ti, Hi, Wi, C, ldi;
    int32_t To, Ho, Wo, ldo;
    int32_t k[3], s[3], padf[3];
} pc_pool_desc;
int pc_maxpool_fwd(const pc_pool_desc* d, const float* x, float* y, uint8_t* argmax, pc_stream s);
int pc_maxpool_bwd(const pc_pool_desc* d, const float* dy, const uint8_t* argmax, float* dx,
                   int accum, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Elementwise helpers */
/* y[n,pos,c] = x[n,pos,c] * scale[n][c]  (Dropout3d, capsules_ucf101.py:428; also its backward) */
int pc_channel_scale(const float* x, int ldx, const float* scale, int N, int64_t pos_per_n, int C,
                     float* y, int ldy, int accum, pc_stream s);
/* dz = dy * act'(y) ; dbias[c] (+)= sum dz   (bias+ReLU / bias+sigmoid / bias-only epilogues) */
int64_t pc_act_bwd_ws_floats(int64_t rows, int C);
int pc_act_bwd(const float* dy, int lddy, const float* y, int ldy, int act, int C, int64_t rows,
               float* dz, int lddz, float* dbias, int accum, float* ws, pc_stream s);
/* NCDHW (reference layout, main_ucf101.py:52-59 input clips) -> NDHWC with C padded to Cpad;
 * flipw mirrors W (torch.flip(.,[4]), main_ucf101.py:100). src may be fp32 or fp64. */
int pc_ncdhw_to_ndhwc(const void* src, int src_is_f64, int N, int C, int64_t thw, int W, int Cpad,
                      int flipw, float* dst, pc_stream s);
int pc_ndhwc_to_ncdhw(const float* src, int ld, int N, int C, int64_t thw, float* dst, pc_stream s);
/* dst[b][c][r] (+)= src[b][r][c] through 32x32 LDS tiles.  Every weight / weight-gradient
 * re-layout between the reference's OI(T)HW / IO(T)HW state_dict layout and the kernels'
 * [O][taps][I] / [I][taps][O] layouts is one call of this (see plan.py). */
int pc_transpose_batched(const float* src, int batch, int R, int Cc, int64_t src_batch_stride,
                         int src_ld, float* dst, int64_t dst_batch_stride, int dst_ld, int accum,
                         pc_stream s);
/* the same for many independent jobs in one launch (the per-step weight re-layouts); `jobs` is HOST memory */
typedef struct pc_transpose_job {
    uint64_t src, dst;              /* device pointers */
    int64_t  src_batch_stride, dst_batch_stride;
    int32_t  batch, R, C, src_ld, dst_ld, accum;
    int32_t  nslices, reserved;     /* nslices > 1: src is that many images slice_stride floats apart (the K-slice images of a weight gradient,
                                     * pc_wgrad_desc.ws_slices); they are added in slice order on the way -- dst (+)= sum_k src_k^T */
    int64_t  slice_stride;
} pc_transpose_job;
int pc_transpose_multi(const pc_transpose_job* jobs, int njobs, pc_stream s);
/* dx[n,h,w,c] (+)= sum over valid taps of cols[n, h-b, w-c'][(b*KW+c')*C + c]: gather half of an exact stride-1 'full'
 * correlation (PrimaryCaps dgrad, capsules_ucf101.py:44-45 backward) computed as GEMM + col2im, so only the
 * Ho*Wo real output positions are multiplied instead of the (Ho+KH-1)*(Wo+KW-1) gathered ones. */
int pc_col2im(const float* cols, int N, int Ho, int Wo, int KH, int KW, int C, float* dx, int lddx,
              int accum, pc_stream s);
/* ------------------------------------------------------------------------------------------
 * Evaluation metrics (evaluate_ucf101.py:142-183, evaluate_jhmdb.py same lines): what the reference does in numpy after
 * `segmentation.cpu()`.  pc_seg_frame_counts: per frame of `nframes` x `pix` logits (B,1,8,H,W contiguous = frame-major)
 * and truth masks, counts[frame] = { #(pred+gt == 2), #(pred+gt != 0), #(gt != 0) } with pred = sigmoid(logit) >= 0.5
 * (:128,:148-149,:160-161); the call zeroes `counts` itself.  pc_map_accumulate: one video -- frames with truth add to
 * n_frames[label] and, for every threshold k/20 (float32, :72) that the frame's inter/union (float64) reaches, to
 * frame_hits[label][k]; the video's summed counts do the same for video_hits; n_vids[label] += 1 (:153-183).
 * All tables int32, [ncls][20] / [ncls], owned and zero-initialised by the caller. */
int pc_seg_frame_counts(const float* logits, const float* gt, int64_t nframes, int64_t pix, int32_t* counts, pc_stream s);
int pc_map_accumulate(const int32_t* counts, int64_t nframes, int label, int ncls, int32_t* frame_hits, int32_t* video_hits,
                      int32_t* n_frames, int32_t* n_vids, pc_stream s);
/* ------------------------------------------------------------------------------------------
 * Input pipeline (datasets/ucf_dataloader.py:146-173 and the box rasterisation of load_video :204-221): from the decoded
 * uint8 frames `video` [F][H][W][3] in HBM, the 8 frames `span8` (host array), the S x S crop at (h0, w0): data =
 * frame / 255 (float64 division, then float32) as NCDHW [3][8][S][S], aug = its horizontal flip, mask [8][S][S] = 1 where
 * any of the frame's R boxes rects[t][r] = (x0, x1, y0, y1) (device int32, frame coordinates, half-open, already clipped
 * the way numpy clips `bbox[f, y:y+h, x:x+w]`) covers the pixel, else 0. */
int pc_clip_from_u8(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                    const int32_t* rects, int R, float* data, float* aug, float* mask, pc_stream s);
/* JHMDB form (datasets/jhmdb_dataloader.py:167-205): the truth is a per-pixel mask per frame, `maskframes` [F][H][W] uint8
 * (> 0 = foreground), and only the frames flagged in valid8 (host array; :187-194) carry it: mask = valid && maskframe > 0,
 * mask_cls [8][S][S] = valid. */
int pc_clip_from_u8_masks(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                          const uint8_t* maskframes, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls,
                          pc_stream s);
/* pc_clip_from_u8 with data / aug written as [8][S][S][4] float32 (r, g, b, 0) -- the NDHWC layout with the RGB clip padded to one 16-byte piece per
 * position that the network's first conv reads (Conv3d_1a_7x7, PC_F_CI3) -- so that a sample written into its place of the next step's minibatch
 * needs no layout conversion at all (StepEngine.sample_stager); mask as pc_clip_from_u8.  data / aug 16-byte aligned. */
int pc_clip_from_u8_ndhwc4(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                           const int32_t* rects, int R, float* data, float* aug, float* mask, pc_stream s);
/* cv2.resize on uint8 images [n][H][W][C] -> [n][Ho][Wo][C] (C <= 4), the calls of the reference's loaders:
 * datasets/jhmdb_dataloader.py:252 (frames, INTER_AREA 320x240 -> 256x256), :267,:281 (puppet masks, INTER_NEAREST),
 * :192,:208 and ucf_dataloader.py:165,171 (224 crop -> frame size, INTER_LINEAR; the identity at 224).  OpenCV's 8-bit
 * algorithm restated (csrc/inputpipe.hip has the rules).  pc_resize_tables is HOST arithmetic (no GPU needed): it writes the
 * int32 coordinate / coefficient table for one (interpolation = cv2 flag 0 nearest | 1 linear | 3 area, sizes) into `tab` if
 * `cap` words suffice and returns the word count; the caller uploads it once and passes the device copy to pc_resize_u8.
 * binarize != 0: the result is (value > 0) of resizing a non-negative mask in float, i.e. 1 where a tap with a positive
 * weight meets a positive sample (`cv2.resize(bbox_img, ...) > 0`, ucf_dataloader.py:170-172). */
int64_t pc_resize_tables(int interpolation, int H, int W, int Ho, int Wo, int32_t* tab, int64_t cap);
int pc_resize_u8(const uint8_t* src, int n, int H, int W, int C, int Ho, int Wo, const int32_t* dev_tab, int binarize,
                 uint8_t* dst, pc_stream s);
int pc_fill(float* p, int64_t n, float v, pc_stream s);
int pc_axpy(float* y, const float* x, int64_t n, float a, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Capsule head (capsules_ucf101.py:290-331, 108-211): votes + 3-iteration EM routing, one
 * wave per spatial position, votes recomputed from poses and W held in LDS.
 *  x      [npos][B*17]   primary caps output (poses B*16 then activations B)
 *  W      [B][C][4][4], beta_u [C][16], beta_a [C]
 *  out    [npos][C*17]   (mu C*16 then a_out C)
 * backward: hand-derived reverse of all iterations (autograd equivalent), per-block partial
 * parameter grads in ws then reduced; dW/dbeta_u/dbeta_a accumulated (+=).
 *  state  optional, pc_em_state_floats(npos) floats, 16-byte aligned: the forward leaves every iteration's routing state
 *         there (assignments, means, variances, ...: 21 KB per position) and a backward given the same buffer loads it
 *         instead of running the routing iterations again; NULL = no state kept / the backward recomputes the forward. */
int64_t pc_em_ws_floats(int npos, int B, int C);
int64_t pc_em_state_floats(int npos);
int pc_em_routing_fwd(const float* x, const float* W, const float* beta_u, const float* beta_a,
                      int npos, int B, int C, float* out, float* state, pc_stream s);
int pc_em_routing_bwd(const float* x, const float* W, const float* beta_u, const float* beta_a,
                      const float* dout, int npos, int B, int C, float* dx, float* dW,
                      float* dbeta_u, float* dbeta_a, float* ws, const float* state, pc_stream s);

/* class-capsule masking (capsules_ucf101.py:438-484):
 *  actor_prediction[b][c] = mean_pos act; mask row: labeled -> one-hot(cls), unlabeled ->
 *  ones (mode 0) or one-hot(argmax pred) (mode 1); eval (mode 2): argmax for all rows.
 *  masked[b,pos,c*16+h] = pose * mask[b][c];  mask is written out for the backward. */
int pc_class_mask_fwd(const float* caps, int Bn, int npos_per_b, int C, const float* cls,
                      const int32_t* labeled, int mode, float* actor_pred, float* mask,
                      float* masked, pc_stream s);
/* dcaps = [dmasked*mask , dactor_pred/npos_per_b] */
int pc_class_mask_bwd(const float* dmasked, const float* dactor_pred, const float* mask, int Bn,
                      int npos_per_b, int C, float* dcaps, pc_stream s);

/* `smooth` ConvTranspose3d(128->1,k3,p1) second stage (capsules_ucf101.py:373,509): the 27 tap
 * projections proj[n,t,h,w,32] (from pc_conv_fwd with Co=27) are summed at their offsets:
 *  out[n,t,h,w] = bias + sum_tap proj[n, (t,h,w)+pad-k(tap), tap]   and the adjoint. */
int pc_tapsum_fwd(const float* proj, int N, int T, int H, int W, const float* bias, float* out, pc_stream s);
int pc_tapsum_bwd(const float* dout, int N, int T, int H, int W, float* dproj, pc_stream s);

/* Collapsed decoder tail: upsample4 -> Dropout3d -> smooth (capsules_ucf101.py:504-509) are three linear ops, so
 * they run as ONE 128->27 transposed conv with per-sample combined weights
 *   Wc[n][ci][tap][j] = sum_co W4[ci][co][tap] * cs[n][co] * Wp[co][j],  bc[n][j] = sum_co b4[co]*cs[n][co]*Wp[co][j]
 * followed by pc_tapsum_fwd.  W4 = upsample4.weight (Ci,Co,taps), Wp = smooth.weight (Co,J=27), cs = Dropout3d scale
 * (N,Co) or NULL.  Wt [N][Ci][taps][32] is the dgrad layout, Wf [N][32][taps][Ci] the forward layout, bc [N][32]. */
int pc_tail_combine(const float* W4, const float* b4, const float* cs, const float* Wp, int N, int Ci,
                    int Co, int taps, int J, float* Wt, float* Wf, float* bc, pc_stream s);
/* sums[n][32] = per-sample column sums of dproj [N][rows_per_n][32] */
int pc_tail_colsum(const float* dproj, int N, int64_t rows_per_n, float* sums, pc_stream s);
/* G [N][Ci][taps][32] = per-sample dWc (from pc_conv_wgrad) -> upsample4.weight/bias and smooth.weight/bias grads (+)= */
int pc_tail_grads(const float* G, const float* sums, const float* W4, const float* b4, const float* cs,
                  const float* Wp, int N, int Ci, int Co, int taps, int J, int center, float* dW4,
                  float* db4, float* dWp, float* dbp, int accum, pc_stream s);
/* the same with the smooth-weight gradient's N * Ci/8 block partials stored in ws (pc_tail_grads_ws_floats floats, no initialisation needed) and
 * added in block order instead of fp32 atomics: bit-identical from run to run (round 6).  ws == NULL: pc_tail_grads. */
int64_t pc_tail_grads_ws_floats(int N, int Ci, int Co);
int pc_tail_grads_ws(const float* G, const float* sums, const float* W4, const float* b4, const float* cs,
                     const float* Wp, int N, int Ci, int Co, int taps, int J, int center, float* dW4,
                     float* db4, float* dWp, float* dbp, int accum, float* ws, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Fused loss: everything main_ucf101.py:89-148 computes from (output, flip_op, loc_msk):
 * BCEWithLogits + Dice on labeled rows, equal-weight L2, bv (utils/helpers.py:8-67) and gv
 * (:70-95) attentive masks with their weighted MSE, combined as the reference does, plus
 * d(total)/d(output), d(total)/d(flip_op).  Masks are detached, like the reference's numpy. */
typedef struct pc_loss_desc {
    int32_t B, T, H, W;             /* T must be 8 (utils/helpers.py:14) */
    int32_t bv, gv, n_frames, predict_maps, jhmdb;
    float   lower_thresh, upper_thresh;   /* <0 = None */
    float   bv_wt, gv_wt, wt_loc, wt_cons, wt_ramp;
} pc_loss_desc;
int64_t pc_loss_ws_floats(const pc_loss_desc* d);
/* output, flip_op: [B][T][H][W] logits (flip_op still W-mirrored, as the model returns it);
 * seg [B][T][H][W]; labeled[B] 0/1.  scalars[8] = {loc, cons, bce, dice, l2, lv1+lv2, lg, n_labeled};
 * total adds wt_cls*spread on the host side.  mask_bv/mask_gv optional outputs (may be NULL). */
int pc_consistency_loss(const pc_loss_desc* d, const float* output, const float* flip_op,
                        const float* seg, const int32_t* labeled, float* scalars, float* d_output,
                        float* d_flip_op, float* mask_bv, float* mask_gv, float* ws, pc_stream s);
/* standalone masks behind utils.helpers.measure_pixelwise_var_v2 / measure_pixelwise_gradient */
int pc_var_mask(const float* pred, const float* flip_pred, int B, int T, int H, int W, int n_frames,
                int use_sig, float* mask, float* ws, pc_stream s);
int pc_grad_mask(const float* pred, int B, int T, int H, int W, float lower, float upper,
                 float* mask, float* ws, pc_stream s);
/* SpreadLoss (utils/losses.py:14-37) on labeled rows: out[2] = {loss, absloss}; dx (+)= wt*dloss/dx */
int pc_spread_loss(const float* x, const float* cls, const int32_t* labeled, int Bn, int C, float m,
                   float wt, float* out, float* dx, pc_stream s);

/* fused Adam over a flat buffer (optim.Adam(lr, weight_decay=0, eps=1e-6), main_ucf101.py:416);
 * gscale folds the 1/world_size of the DP mean. */
int pc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                 float b2, float eps, int step, float gscale, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Row-spectral PrimaryCaps (capsules_ucf101.py:43-49: Conv2d 832 -> 512+32, 9x9, stride 1).  A real DFT
 * of length P = image width along the rows turns the kx taps into a product per frequency; what is left
 * is a 9x1 conv with complex channels per frequency u = 0..P/2.  The complex product is taken in its
 * three-multiplication form (X = Xr + i Xi, conj(W) = Wr - i Wi):
 *     t0 = (Xr + Xi) Wr,  t1 = Xi (Wr - Wi),  t2 = Xr (Wr + Wi);   Re = t0 - t1,  Im = t0 - t2,
 * so the whole layer is ONE grouped real conv for pc_conv_fwd / pc_conv_wgrad (3 groups per complex frequency,
 * 1 for DC / Nyquist; Ci -> Co, 9x1 taps): a quarter of the direct form's multiply-adds, equal to it in exact arithmetic.
 * The sums/differences of the operands and results are folded into the DFT matrices.  These three
 * HBM-bound helpers are the rest of it. */
/* out[r][o][c] = sum_i M[o][i] * in[r][i][c] (+ bias[c]; activation on channels >= act_c0; accum adds
 * the old value first): a small dense matrix along one tensor axis.  Element offsets (floats):
 * in:  r*in_sr  + (i / in_split)*in_hi   + (i % in_split)*in_lo  + c
 * out: r*out_sr + (o / out_split)*out_hi + (o % out_split)*out_lo + c */
typedef struct pc_axis_desc {
    int32_t R, I, O, C;
    int32_t in_split, out_split, act, act_c0, accum;
    int32_t in_sr, in_hi, in_lo, out_sr, out_hi, out_lo;
} pc_axis_desc;
int pc_axis_linear(const pc_axis_desc* d, const float* in, const float* M, const float* bias, float* out, pc_stream s);
/* weights in a kernel layout in[A][KY*KX][B] -> weight planes [A][KY][B]: three per complex frequency,
 *   V0 = Wr, V1 = Wr - Wi, V2 = Wr + Wi,  Wr = sum_kx in*tw[u][kx][0], Wi = sum_kx in*tw[u][kx][1]
 * (tw = cos, -sin of 2*pi*u*kx/P), followed by ONE plane (Wr) for each of the Ur trailing frequencies of tw whose spectrum
 * is real (DC and, for even P, Nyquist: X and W have no imaginary part there, so one real product is the whole term).
 * in = [Co][taps][Ci] gives the forward GEMM weights, in = [Ci][taps][Co] the dgrad GEMM weights. */
int pc_wspec_fwd(const float* in, const float* tw, int A, int B, int KY, int KX, int U, int Ur, float* out, pc_stream s);
/* adjoint: kg[a][ky*KX+kx][b] = sum_u tw[u][kx][0]*(d0+d1+d2) + tw[u][kx][1]*(d2-d1) over the complex frequencies plus
 * tw[u][kx][0]*d0 over the real ones; dV in the plane order of pc_wspec_fwd */
int pc_wspec_bwd(const float* dV, const float* tw, int A, int B, int KY, int KX, int U, int Ur, float* kg, pc_stream s);

/* The same two maps straight from / to the reference's master layout w[A][B][KY][KX] (OIHW, taps contiguous), so the
 * 36.7 M-element PrimaryCaps weight needs no kernel-layout copies: rows [a0, a0 + Acnt) of an Atot-row weight (pose and
 * activation capsules are two tensors).  out_f = planes [g][Atot][KY][B] (forward GEMM), out_t = [g][B][KY][Atot] (dgrad
 * GEMM); either may be NULL.  The adjoint reads plane gradients in the out_f layout and writes (accum: adds to) dw. */
int pc_wspec_master_fwd(const float* w, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                        float* out_f, float* out_t, pc_stream s);
/* pc_wspec_master_fwd with the planes written as the three bf16 terms of every value (pc_split_planes' format, term p at + p * plane_stride
 * elements) for pc_conv_fwd_x6: no fp32 copy of the 2 x 167 M-element PrimaryCaps weight planes exists in that mode */
int pc_wspec_master_planes(const float* w, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                           uint16_t* out_f, uint16_t* out_t, int64_t plane_stride, pc_stream s);
int pc_wspec_master_bwd(const float* dV, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                        float* dw, int accum, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Merged decoder tail (capsules_ucf101.py:504-509).  upsample4 -> Dropout3d -> smooth composes, per dimension, into one
 * stride-2 transposed conv with five taps k5 = k4 + ks (o = 2i - 2 + k5) and ONE output channel; the single term that
 * would pass through the cropped position m = -1 of upsample4's output (input index 0, k4 = 0, only in tap k5 = 2) is
 * left out of that tap's weight for the positions whose index is 0 in that dimension.  The positions therefore fall into
 * 8 classes z = 4*(it==0) + 2*(ih==0) + (iw==0), each with its own 128 x 125 weight matrix, and
 * cols[n][i][slot] = x[n][i][:] . W5[n][z(i)][:][slot], slot = (k5t*5 + k5h)*5 + k5w (125, padded to 128), is one grouped
 * 1x1 GEMM per class over its sub-lattice for pc_conv_fwd; these helpers are the rest. */
/* wf [N][32][27][Ci] (pc_tail_combine's forward layout) -> W5f [N][8][128][Ci] (forward GEMM), W5t [N][8][Ci][128] (dgrad GEMM) */
int pc_tail6_weights(const float* wf, int N, int Ci, float* W5f, float* W5t, pc_stream s);
/* out[n][o] = bsm[0] + sum_{ks: o+1-ks in grid} bc[n][ks] + sum of the <= 27 entries of cols [N][It][Ih][Iw][128] that
 * land on o = 2i - 2 + k5; out is [N][2It][2Ih][2Iw] */
int pc_tail6_gather(const float* cols, const float* bc, const float* bsm, int N, int It, int Ih, int Iw, float* out, pc_stream s);
/* the transpose: dcols[n][i][slot] = dout[n][2i - 2 + k5] (0 where that output does not exist) */
int pc_tail6_scatter(const float* dout, int N, int It, int Ih, int Iw, float* dcols, pc_stream s);
/* dW5 [N][8][Ci][128] -> gradient of the combined weights Gc [N][Ci][27][32] in the layout pc_tail_grads consumes */
int pc_tail6_wgrad_map(const float* dW5, int N, int Ci, float* Gc, pc_stream s);
/* the same from the classes' K-slice workspaces (pc_wgrad_desc.ws_slices, one launch per class with the N clip-passes as batched problems,
 * gbstride = Ci*128): ws = [z][k < nslices8[z]][N][Ci][128] back to back, nslices8[z] = 0 for an empty class.  A class's images are added in slice
 * order INTO ITS IMAGE 0 (ws is modified; the next pc_conv_wgrad into it rewrites image 0), the class sums in class order -- no fp32 atomics */
int pc_tail6_wgrad_map_slices(float* ws, const int32_t* nslices8, int N, int Ci, float* Gc, pc_stream s);
/* sums[n][ks] = sum of dout[n][o] over the outputs whose smooth tap ks reads an in-grid position (what pc_tail_colsum
 * gives on the 27-channel form) */
int pc_tail6_bias_sums(const float* dout, int N, int It, int Ih, int Iw, float* sums, pc_stream s);
/* the same without atomics: ws (pc_tail6_bias_sums_ws_floats floats, no initialisation needed) takes one partial row per block, a second
 * kernel adds them in block order.  ws == NULL: pc_tail6_bias_sums. */
int64_t pc_tail6_bias_sums_ws_floats(int N, int It, int Ih, int Iw);
int pc_tail6_bias_sums_ws(const float* dout, int N, int It, int Ih, int Iw, float* sums, float* ws, pc_stream s);

/* ------------------------------------------------------------------------------------------
 * Op-list runner: the host builds the step as a flat list of POD ops once (shape inference and
 * arena planning in Python) and the library replays it with no per-op host round trip. */
typedef struct pc_op {
    int32_t  kind;                  /* PC_OP_* */
    int32_t  i[48];
    float    f[8];
    int32_t  lane;                  /* which of the runner's streams the op is enqueued on (0 = main) */
    uint64_t p[12];                 /* device pointers */
    int64_t  l[4];
} pc_op;
enum {
    PC_OP_CONV = 1, PC_OP_WGRAD, PC_OP_BN_FINALIZE, PC_OP_BN_APPLY, PC_OP_BN_EVAL_STAT, PC_OP_BN_BWD,
    PC_OP_POOL_FWD, PC_OP_POOL_BWD, PC_OP_CHSCALE, PC_OP_ACT_BWD, PC_OP_TO_NDHWC, PC_OP_TO_NCDHW,
    PC_OP_TRANSPOSE, PC_OP_FILL, PC_OP_AXPY, PC_OP_EM_FWD, PC_OP_EM_BWD, PC_OP_CMASK_FWD,
    PC_OP_CMASK_BWD, PC_OP_TAPSUM_FWD, PC_OP_TAPSUM_BWD, PC_OP_LOSS, PC_OP_SPREAD, PC_OP_ADAM,
    PC_OP_TAIL_COMBINE, PC_OP_TAIL_COLSUM, PC_OP_TAIL_GRADS, PC_OP_COL2IM,
    PC_OP_AXIS,                     /* i[0..14] = pc_axis_desc; p = in, M, bias, out */
    PC_OP_WSPEC_FWD,                /* i = A, B, KY, KX, U, Ur; p = in, tw, out */
    PC_OP_WSPEC_BWD,                /* i = A, B, KY, KX, U, Ur; p = dV, tw, kg */
    PC_OP_WSPEC_MASTER_FWD,         /* i = Acnt, a0, Atot, B, KY, KX, U, Ur; p = w, tw, out_f, out_t */
    PC_OP_WSPEC_MASTER_BWD,         /* i = Acnt, a0, Atot, B, KY, KX, U, Ur, accum; p = dV, tw, dw */
    PC_OP_TAIL6_WEIGHTS,            /* i = N, Ci; p = wf, W5f, W5t */
    PC_OP_TAIL6_GATHER,             /* i = N, It, Ih, Iw; p = cols, bc, bsm, out */
    PC_OP_TAIL6_SCATTER,            /* i = N, It, Ih, Iw; p = dout, dcols */
    PC_OP_TAIL6_WGRAD_MAP,          /* i = N, Ci, sliced, nslices[8]; p = dW5 (sliced: the classes' K-slice workspaces), Gc */
    PC_OP_TAIL6_BIAS_SUMS,          /* i = N, It, Ih, Iw; p = dout, sums, ws (or 0) */
    PC_OP_TRANSPOSE_MULTI,          /* p[0] = HOST pointer to pc_transpose_job[i[0]] (kept alive by the owner of the list) */
    PC_OP_FORK,                     /* i[0] = lane bitmask: those lanes wait for everything enqueued so far on lane i[1] (0 by default) */
    PC_OP_JOIN,                     /* i[0] = lane bitmask: lane 0 waits for everything enqueued on those lanes */
    PC_OP_WGRAD_MULTI,              /* p[0] = HOST pointer to pc_wgrad_job[i[0]] (kept alive by the owner of the list): pc_conv_wgrad_multi */
    PC_OP_WINO_CONV,                /* i[0..15] = pc_wino_desc; p = in, U, bias, out, bnpart */
    PC_OP_WINO_WEIGHTS,             /* i = O, I, KT, flip, m (4: pc_wino4_weights); l = sO, sT, sI; p = w, U */
    PC_OP_CONV_X6,                  /* i = pc_conv_desc (flags with PC_F_X6); l[0] = plane stride, l[1] = workspace floats; p = in, wplanes, bias, cscale, out, bnpart, ws (0: none) */
    PC_OP_SPLIT_PLANES,             /* l = n, plane stride; p = src, planes */
    PC_OP_SPLIT_PLANES_MULTI,       /* p[0] = HOST pointer to pc_split_job[i[0]] (kept alive by the owner of the list) */
    PC_OP_WSPEC_MASTER_PLANES,      /* i = Acnt, a0, Atot, B, KY, KX, U, Ur; l[0] = plane stride; p = w, tw, out_f planes, out_t planes */
    PC_OP_BN_FIN_APPLY,             /* i = npg, groups, C, ldz, ldy, relu; l = count per group, rows; f = eps, momentum; p = part, gamma, beta, running_mean, running_var, stat, z, y */
    PC_OP_WGRAD_FOLD,               /* i[0] = nslices; l[0] = image floats; p = ws: pc_wgrad_fold */
    PC_OP__COUNT
};
#define PC_MAX_LANES 8
/* Replays the list on one stream (lane tags ignored: FORK/JOIN are no-ops, order = list order). */
int pc_run_ops(const pc_op* ops, int n, pc_stream s);
/* Replays the list over `nlanes` HIP streams: op k is enqueued on lanes[ops[k].lane]; the independent
 * branches of an Inception module (pytorch_i3d.py:149-154) and their backward run concurrently between
 * a FORK and the matching JOIN.  Lanes >= nlanes fold onto lane 0.  Every list must end joined. */
int pc_run_ops_lanes(const pc_op* ops, int n, const pc_stream* lanes, int nlanes);
/* `target` waits for everything enqueued so far on each of the nlanes streams (one persistent event per lane, no host sync):
 * how a gradient bucket's all-reduce stream is put behind every lane of a step replayed in segments (dist.GradReducer.launch;
 * the reference has no counterpart -- it is single-GPU, main_ucf101.py:171-184). */
int pc_streams_fanin(pc_stream target, const pc_stream* lanes, int nlanes);
/* Destroys the events the calling thread's replays created (FORK / JOIN, fan-in, the timing pool; VERDICT r1 weak 12: they are
 * thread-local and otherwise live until the process ends).  Callable at any time the streams are idle; later calls re-create them. */
int pc_release_thread_events(void);
/* same, with a hipEvent pair around every op of `kind` on the stream that op runs on (for PC_OP_CONV, whose ops are
 * one kernel each, the pair rides in the kernel's own dispatch and brackets exactly the kernel; other kinds are
 * bracketed by recorded events); returns elapsed ms summed over those ops in *ms and their count in *count (bench.py
 * roofline leg).  Synchronises all lanes before returning (unless ms == NULL, see below). */
int pc_run_ops_timed(const pc_op* ops, int n, int kind, float* ms, int* count, const pc_stream* lanes, int nlanes);
/* With ms == NULL pc_run_ops_timed does not synchronise: the event pairs stay pending (per host thread) until this call
 * waits for them and returns their summed elapsed ms and their number. */
int pc_run_ops_timed_collect(float* ms, int* count);

#ifdef __cplusplus
}
#endif
#endif
