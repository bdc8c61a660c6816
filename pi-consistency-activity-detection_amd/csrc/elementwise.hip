// HBM-bound members of the train step for gfx950: BatchNorm3d(train)+ReLU fwd/bwd with
// two-pass batch groups, MaxPool3d SAME fwd/bwd, Dropout3d channel scale, activation/bias
// backward, layout changes, batched transposes for weight re-layout, fill/axpy, fused Adam.
// All loads/stores are float4 along the channel (innermost NDHWC) dimension.
#include "common.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------ BN forward
// part [groups][npg][2][C]; stat [groups][4][C] = mean, invstd, scale, shift.
constexpr int FIN_RL = 16;   // row lanes of the partial-sum finalize kernels
constexpr int FIN_CL = 16;   // channels per block: C / 16 blocks, so that even the 64-channel layers spread over a few CUs
constexpr int FIN_T = FIN_RL * FIN_CL;
// 256-thread blocks (one wave per SIMD, <= 32 registers, 4 KB of LDS): these kernels are a handful of blocks on the step's dependency chain, and
// they run beside GEMM launches that fill every resident-block slot for hundreds of microseconds.  As 1024-thread blocks (four waves per SIMD and
// 128 registers per lane at once on ONE CU) they waited for a CU to drain: 35 us on average in the four-lane step, up to 290, against 5 alone.

// sum over partial rows [npg][2][C] for FIN_CL channels; result valid on threads with rl == 0
template <typename TP>
__device__ __forceinline__ void partial_colsum(const TP* __restrict__ part, int npg, int C, int c, int rl,
                                               double (*sh)[FIN_RL][FIN_CL], int cl, double& s, double& s2) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (c < C) {
        int i = rl;
        for (; i + 3 * FIN_RL < npg; i += 4 * FIN_RL) {          // four independent rows in flight per thread
            TP x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const TP* q = part + (size_t)(i + u * FIN_RL) * 2 * C + c; x[u] = q[0]; y[u] = q[C]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] += (double)x[u]; b[u] += (double)y[u]; }
        }
        for (; i < npg; i += FIN_RL) { const TP* q = part + (size_t)i * 2 * C + c; a[0] += (double)q[0]; b[0] += (double)q[C]; }
    }
    sh[0][rl][cl] = (a[0] + a[1]) + (a[2] + a[3]); sh[1][rl][cl] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    s = 0; s2 = 0;
    if (rl == 0)
        for (int r = 0; r < FIN_RL; ++r) { s += sh[0][r][cl]; s2 += sh[1][r][cl]; }
    __syncthreads();
}

// Two-stage form for layers with thousands of partial rows (the stem leaves 2 x 6272 per batch group: one block per 16 channels walked
// them in 85 us, alone on the dependency chain at the very start of a step): stage 1 spreads the rows of a group over FIN_S blocks per
// channel block and leaves FIN_S double-precision partial rows, stage 2 is the finalize kernel over those.  Fixed order: deterministic.
constexpr int FIN_S = 32;
__global__ __launch_bounds__(FIN_T) void bn_partial_reduce_kernel(const float* __restrict__ part, int npg, int C, double* __restrict__ ws) {
    __shared__ double sh[2][FIN_RL][FIN_CL];
    const int cl = threadIdx.x % FIN_CL, rl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl, sl = blockIdx.y, g = blockIdx.z;
    const int per = (npg + FIN_S - 1) / FIN_S, r0 = sl * per, r1 = min(npg, r0 + per);
    double s, s2;
    partial_colsum(part + ((size_t)g * npg + r0) * 2 * C, max(0, r1 - r0), C, c, rl, sh, cl, s, s2);
    if (rl == 0 && c < C) {
        double* o = ws + ((size_t)g * FIN_S + sl) * 2 * C;
        o[c] = s; o[C + c] = s2;
    }
}

template <typename TP>
__global__ __launch_bounds__(FIN_T) void bn_finalize_kernel(const TP* __restrict__ part, int npg, int groups,
                                                           int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float mom,
                                                           float* rmean, float* rvar, float* __restrict__ stat) {
    __shared__ double sh[2][FIN_RL][FIN_CL];
    const int cl = threadIdx.x % FIN_CL, rl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl;
    float rm = 0.f, rv = 0.f;
    const bool upd = (rmean != nullptr) && c < C && rl == 0;
    if (upd) { rm = rmean[c]; rv = rvar[c]; }
    for (int g = 0; g < groups; ++g) {
        double s, s2;
        partial_colsum(part + (size_t)g * npg * 2 * C, npg, C, c, rl, sh, cl, s, s2);
        if (rl == 0 && c < C) {
            const double mean = s / count;
            double var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * invstd;
            float* st = stat + (size_t)g * 4 * C;
            st[c] = (float)mean; st[C + c] = invstd; st[2 * C + c] = sc; st[3 * C + c] = beta[c] - (float)mean * sc;
            if (upd) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                rm = (1.f - mom) * rm + mom * (float)mean;
                rv = (1.f - mom) * rv + mom * (float)unb;
            }
        }
    }
    if (upd) { rmean[c] = rm; rvar[c] = rv; }
}

__global__ __launch_bounds__(256) void bn_eval_stat_kernel(const float* gamma, const float* beta, const float* rm,
                                                           const float* rv, float eps, int C, float* stat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps), sc = gamma[c] * invstd;
    stat[c] = rm[c]; stat[C + c] = invstd; stat[2 * C + c] = sc; stat[3 * C + c] = beta[c] - rm[c] * sc;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ stat,
                                                       int C4, int64_t total4, int64_t rows_per_group, float* __restrict__ y,
                                                       int ldy, int relu) {
    const int C = C4 * 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const unsigned row = (unsigned)idx / (unsigned)C4;          // total4 < 2^31 (checked by the caller): 32-bit index math
        const int c = (int)((unsigned)idx - row * C4) * 4;
        const float* st = stat + (size_t)(row / (unsigned)rows_per_group) * 4 * C;
        const f32x4 v = *(const f32x4*)(z + (size_t)row * ldz + c);
        const f32x4 sc = *(const f32x4*)(st + 2 * C + c), sh = *(const f32x4*)(st + 3 * C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e] * sc[e] + sh[e]; o[e] = relu ? fmaxf(t, 0.f) : t; }
        *(f32x4*)(y + (size_t)row * ldy + c) = o;
    }
}

// ------------------------------------------------------------------------------ column reductions
// Generic per-channel reduction of two quantities over a block of rows.  MODE 0: BN backward
// (sum dzh, sum dzh*xhat); MODE 1: activation backward (sum dz, unused).
struct RedP {
    const float* a; int lda; const float* b; int ldb; const float* stat; int C4; int64_t rows, rows_per_group;
    int64_t rows_per_block; int act; float* part;   // part [nblk][2][C]
    int64_t a_gs, b_gs, stat_gs, part_gs;           // blockIdx.y = batch group: element strides of a, b, stat, part (0 for one group)
};

template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(RedP p) {
    p.a += blockIdx.y * p.a_gs; p.b += blockIdx.y * p.b_gs; p.stat += blockIdx.y * p.stat_gs; p.part += blockIdx.y * p.part_gs;
    const int C4 = p.C4, C = C4 * 4;
    const int rpi = 256 / C4;                       // rows per iteration
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
    const bool active = rl < rpi;
    const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t r1 = min(p.rows, r0 + p.rows_per_block);
    // A thread's sum runs over thousands of rows of mixed sign (a BatchNorm bias gradient is ~1e-3 of sum |dy|): fp32 chains of that
    // length left the BN parameter gradients 3x further from an fp64 run than the fp32 reference's (pairwise) sums.  Short fp32 runs of
    // four rows, flushed into double: the kernel stays HBM-bound.
    double sd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const int c = c4 * 4;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int run = 0;
        for (int64_t r = r0 + rl; r < r1; r += rpi) {
            const f32x4 dy = *(const f32x4*)(p.a + r * p.lda + c);
            const f32x4 zz = *(const f32x4*)(p.b + r * p.ldb + c);
            if (MODE == 0) {
                const float* st = p.stat;              // one batch group per blockIdx.y (p.rows == p.rows_per_group)
                const f32x4 mean = *(const f32x4*)(st + c), inv = *(const f32x4*)(st + C + c);
                const f32x4 sc = *(const f32x4*)(st + 2 * C + c), sh_ = *(const f32x4*)(st + 3 * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yv = zz[e] * sc[e] + sh_[e];
                    const float d = (p.act == PC_ACT_RELU && !(yv > 0.f)) ? 0.f : dy[e];
                    s[e] += d;
                    s[4 + e] += d * (zz[e] - mean[e]) * inv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = dy[e];
                    if (p.act == PC_ACT_RELU) d = zz[e] > 0.f ? d : 0.f;
                    else if (p.act == PC_ACT_SIGMOID) d = d * zz[e] * (1.f - zz[e]);
                    s[e] += d;
                }
            }
            if (++run == 4) {
                run = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) { sd[e] += (double)s[e]; s[e] = 0.f; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sd[e] += (double)s[e];
    }
    __shared__ double shd[256 * 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) shd[threadIdx.x * 8 + e] = sd[e];
    __syncthreads();
    if (threadIdx.x < C4) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < rpi; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += shd[(q * C4 + threadIdx.x) * 8 + e];
        float* o = p.part + (size_t)blockIdx.x * 2 * C + threadIdx.x * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (float)t[e]; o[C + e] = (float)t[4 + e]; }
    }
}

// partials [groups][npg][2][C] -> coef [groups][2][C] = (sum_dzh/count, sum_dzh_xhat/count);
// dgamma/dbeta (+)= sums over all groups.
__global__ __launch_bounds__(FIN_T) void bn_bwd_finalize_kernel(const float* __restrict__ part, int npg, int groups, int C,
                                                               double count, float* coef, float* dgamma, float* dbeta,
                                                               int accum) {
    __shared__ double sh[2][FIN_RL][FIN_CL];
    const int cl = threadIdx.x % FIN_CL, rl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl;
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < groups; ++g) {
        double s, s2;
        partial_colsum(part + (size_t)g * npg * 2 * C, npg, C, c, rl, sh, cl, s, s2);
        if (rl == 0 && c < C) {
            coef[(size_t)g * 2 * C + c] = (float)(s / count);
            coef[(size_t)g * 2 * C + C + c] = (float)(s2 / count);
            tb += s; tg += s2;
        }
    }
    if (rl == 0 && c < C) {
        if (dbeta) dbeta[c] = (accum ? dbeta[c] : 0.f) + (float)tb;
        if (dgamma) dgamma[c] = (accum ? dgamma[c] : 0.f) + (float)tg;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ z,
                                                           int ldz, const float* __restrict__ stat, const float* __restrict__ coef,
                                                           int C4, int64_t total4, int64_t rows_per_group, int relu,
                                                           float* __restrict__ dz, int lddz) {
    const int C = C4 * 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const unsigned row = (unsigned)idx / (unsigned)C4;
        const int c = (int)((unsigned)idx - row * C4) * 4;
        const int g = (int)(row / (unsigned)rows_per_group);
        const float* st = stat + (size_t)g * 4 * C;
        const float* cf = coef + (size_t)g * 2 * C;
        const f32x4 d = *(const f32x4*)(dy + (size_t)row * lddy + c), zz = *(const f32x4*)(z + (size_t)row * ldz + c);
        const f32x4 mean = *(const f32x4*)(st + c), inv = *(const f32x4*)(st + C + c);
        const f32x4 sc = *(const f32x4*)(st + 2 * C + c), sh_ = *(const f32x4*)(st + 3 * C + c);
        const f32x4 c1 = *(const f32x4*)(cf + c), c2 = *(const f32x4*)(cf + C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float yv = zz[e] * sc[e] + sh_[e];
            const float dd = (relu && !(yv > 0.f)) ? 0.f : d[e];
            o[e] = sc[e] * (dd - c1[e] - (zz[e] - mean[e]) * inv[e] * c2[e]);
        }
        *(f32x4*)(dz + (size_t)row * lddz + c) = o;
    }
}


// ------------------------------------------------------------------------------ BatchNorm with the finalize folded into the apply kernel (round 5)
// OPTIONAL (PICONS_BN_FUSED=1; off by default: measured no gain on the step, +0.1 ms of single-stream kernel time -- DESIGN.md 6, round 5).
// The finalize kernels are a handful of blocks between two streaming kernels on the step's dependency chain: 6 - 8 us alone, 20 - 26 us each
// in the four-lane step, where they wait for a free slot beside the other lanes' whole-CU GEMM blocks (17 + 15 such launches on lane 0 alone).
// For layers with few partial rows (the 28 x 28 layers: <= 256 per batch group) every block of the apply kernel instead reduces the partial
// rows of ITS 64 channels itself -- 16 row lanes x 16 float4 columns, doubles, fixed order; a few tens of KB from L2 -- and goes on to its
// rows.  Grid = (row blocks, 64-channel blocks, batch groups); the row block 0 of group 0 also writes what leaves the layer (stat / running
// statistics; d gamma / d beta), walking the other groups' partial rows for that.
constexpr int FA_CB = 64;        // channels per block
// sums of the partial rows [npg][2][C] over rows, for the channels cb * 64 .. + 63: valid on threads tid < 64 (channel cb * 64 + tid)
__device__ __forceinline__ void fa_partial_sums(const float* __restrict__ part, int npg, int C, int cb, int tid, double (*shd)[FA_CB][2], double& s, double& s2) {
    const int c4 = tid & 15, rl = tid >> 4, c = cb * FA_CB + c4 * 4;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (c < C) {
#pragma unroll 4
        for (int i = rl; i < npg; i += 16) {
            const f32x4 x = *(const f32x4*)(part + (size_t)i * 2 * C + c), y = *(const f32x4*)(part + (size_t)i * 2 * C + C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += (double)x[e]; b[e] += (double)y[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {          // the four row lanes of a wave (lane bits 4, 5), then the four waves through LDS
        a[e] += __shfl_xor(a[e], 16, 64); a[e] += __shfl_xor(a[e], 32, 64);
        b[e] += __shfl_xor(b[e], 16, 64); b[e] += __shfl_xor(b[e], 32, 64);
    }
    __syncthreads();                       // (a previous call's readers are done with shd)
    if ((tid & 63) < 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { shd[tid >> 6][c4 * 4 + e][0] = a[e]; shd[tid >> 6][c4 * 4 + e][1] = b[e]; }
    }
    __syncthreads();
    s = 0; s2 = 0;
    if (tid < FA_CB) {
        s = ((shd[0][tid][0] + shd[1][tid][0]) + shd[2][tid][0]) + shd[3][tid][0];
        s2 = ((shd[0][tid][1] + shd[1][tid][1]) + shd[2][tid][1]) + shd[3][tid][1];
    }
}

__global__ __launch_bounds__(256) void bn_fin_apply_kernel(const float* __restrict__ part, int npg, int groups, int C, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float mom,
                                                           float* rmean, float* rvar, float* __restrict__ stat, const float* __restrict__ z, int ldz,
                                                           float* __restrict__ y, int ldy, int relu, int rows_per_group, int rows_per_block) {
    __shared__ double shd[4][FA_CB][2];
    __shared__ __attribute__((aligned(16))) float shs[2][FA_CB];
    const int tid = threadIdx.x, rb = blockIdx.x, cb = blockIdx.y, g = blockIdx.z;
    const int ch = cb * FA_CB + tid;                  // this thread's channel in the per-channel sections (tid < 64)
    double s, s2;
    fa_partial_sums(part + (size_t)g * npg * 2 * C, npg, C, cb, tid, shd, s, s2);
    double mean0 = 0, var0 = 0;
    if (tid < FA_CB && ch < C) {
        const double mean = s / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[ch] * invstd, sh = beta[ch] - (float)mean * sc;
        shs[0][tid] = sc; shs[1][tid] = sh;
        if (rb == 0) {
            float* st = stat + (size_t)g * 4 * C;
            st[ch] = (float)mean; st[C + ch] = invstd; st[2 * C + ch] = sc; st[3 * C + ch] = sh;
        }
        mean0 = mean; var0 = var;
    }
    if (rb == 0 && g == 0 && rmean != nullptr) {      // block-uniform: running statistics over the batch groups IN ORDER (two forward passes)
        float rm = 0.f, rv = 0.f;
        if (tid < FA_CB && ch < C) { rm = rmean[ch]; rv = rvar[ch]; }
        for (int q = 0; q < groups; ++q) {
            double mean = mean0, var = var0;
            if (q > 0) {
                double t, t2;
                fa_partial_sums(part + (size_t)q * npg * 2 * C, npg, C, cb, tid, shd, t, t2);
                mean = t / count; var = t2 / count - mean * mean;
                if (var < 0.0) var = 0.0;
            }
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rm = (1.f - mom) * rm + mom * (float)mean;
            rv = (1.f - mom) * rv + mom * (float)unb;
        }
        if (tid < FA_CB && ch < C) { rmean[ch] = rm; rvar[ch] = rv; }
    }
    __syncthreads();
    const int c4 = tid & 15, rl = tid >> 4, c = cb * FA_CB + c4 * 4;
    if (c >= C) return;
    const f32x4 sc = *(const f32x4*)&shs[0][c4 * 4], sh = *(const f32x4*)&shs[1][c4 * 4];
    const int64_t r0 = (int64_t)g * rows_per_group + (int64_t)rb * rows_per_block;
    const int64_t r1 = min((int64_t)(g + 1) * rows_per_group, r0 + rows_per_block);
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 16) {
        const f32x4 v = *(const f32x4*)(z + r * ldz + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e] * sc[e] + sh[e]; o[e] = relu ? fmaxf(t, 0.f) : t; }
        *(f32x4*)(y + r * ldy + c) = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_fin_apply_kernel(const float* __restrict__ part, int npg, int groups, int C, double count,
                                                               const float* __restrict__ dy, int lddy, const float* __restrict__ z, int ldz,
                                                               const float* __restrict__ stat, int relu, float* __restrict__ dz, int lddz,
                                                               float* dgamma, float* dbeta, int accum, int rows_per_group, int rows_per_block) {
    __shared__ double shd[4][FA_CB][2];
    __shared__ __attribute__((aligned(16))) float shs[2][FA_CB];
    const int tid = threadIdx.x, rb = blockIdx.x, cb = blockIdx.y, g = blockIdx.z;
    const int ch = cb * FA_CB + tid;
    double s, s2;
    fa_partial_sums(part + (size_t)g * npg * 2 * C, npg, C, cb, tid, shd, s, s2);
    if (tid < FA_CB && ch < C) { shs[0][tid] = (float)(s / count); shs[1][tid] = (float)(s2 / count); }
    if (rb == 0 && g == 0) {                          // block-uniform: d beta / d gamma = the sums over every group, in group order
        double tb = s, tg = s2;
        for (int q = 1; q < groups; ++q) {
            double t, t2;
            fa_partial_sums(part + (size_t)q * npg * 2 * C, npg, C, cb, tid, shd, t, t2);
            tb += t; tg += t2;
        }
        if (tid < FA_CB && ch < C) {
            if (dbeta) dbeta[ch] = (accum ? dbeta[ch] : 0.f) + (float)tb;
            if (dgamma) dgamma[ch] = (accum ? dgamma[ch] : 0.f) + (float)tg;
        }
    }
    __syncthreads();
    const int c4 = tid & 15, rl = tid >> 4, c = cb * FA_CB + c4 * 4;
    if (c >= C) return;
    const f32x4 c1 = *(const f32x4*)&shs[0][c4 * 4], c2 = *(const f32x4*)&shs[1][c4 * 4];
    const float* st = stat + (size_t)g * 4 * C;
    const f32x4 mean = *(const f32x4*)(st + c), inv = *(const f32x4*)(st + C + c);
    const f32x4 sc = *(const f32x4*)(st + 2 * C + c), sh_ = *(const f32x4*)(st + 3 * C + c);
    const int64_t r0 = (int64_t)g * rows_per_group + (int64_t)rb * rows_per_block;
    const int64_t r1 = min((int64_t)(g + 1) * rows_per_group, r0 + rows_per_block);
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 16) {
        const f32x4 d = *(const f32x4*)(dy + r * lddy + c), zz = *(const f32x4*)(z + r * ldz + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float yv = zz[e] * sc[e] + sh_[e];
            const float dd = (relu && !(yv > 0.f)) ? 0.f : d[e];
            o[e] = sc[e] * (dd - c1[e] - (zz[e] - mean[e]) * inv[e] * c2[e]);
        }
        *(f32x4*)(dz + r * lddz + c) = o;
    }
}

// partial [nblk][2][C] -> dbias (+)=
__global__ __launch_bounds__(FIN_T) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* out, int accum) {
    __shared__ double sh[2][FIN_RL][FIN_CL];
    const int cl = threadIdx.x % FIN_CL, rl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl;
    double s, s2;
    partial_colsum(part, nblk, C, c, rl, sh, cl, s, s2);
    if (rl == 0 && c < C) out[c] = (accum ? out[c] : 0.f) + (float)s;
}

__global__ __launch_bounds__(256) void act_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                            int act, int C4, int64_t total4, float* __restrict__ dz, int lddz) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const size_t row = (unsigned)idx / (unsigned)C4;
        const int c = (int)((unsigned)idx - (unsigned)row * C4) * 4;
        const f32x4 d = *(const f32x4*)(dy + row * lddy + c), yy = *(const f32x4*)(y + row * ldy + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = d[e];
            if (act == PC_ACT_RELU) v = yy[e] > 0.f ? v : 0.f;
            else if (act == PC_ACT_SIGMOID) v = v * yy[e] * (1.f - yy[e]);
            o[e] = v;
        }
        *(f32x4*)(dz + row * lddz + c) = o;
    }
}

// ------------------------------------------------------------------------------ max pool
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const pc_pool_desc d, const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ am, int64_t total4) {
    const int C4 = d.C / 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        unsigned pos = (unsigned)idx / (unsigned)C4;
        const int c = (int)((unsigned)idx - pos * C4) * 4;
        const size_t opos = pos;
        const int wo = (int)(pos % (unsigned)d.Wo); pos /= (unsigned)d.Wo;
        const int ho = (int)(pos % (unsigned)d.Ho); pos /= (unsigned)d.Ho;
        const int to = (int)(pos % (unsigned)d.To); const int n = (int)(pos / (unsigned)d.To);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {255, 255, 255, 255};
        int tap = 0;
        for (int a = 0; a < d.k[0]; ++a) {
            const int t = to * d.s[0] - d.padf[0] + a;
            for (int b = 0; b < d.k[1]; ++b) {
                const int h = ho * d.s[1] - d.padf[1] + b;
                for (int cc = 0; cc < d.k[2]; ++cc, ++tap) {
                    const int w = wo * d.s[2] - d.padf[2] + cc;
                    const bool inb = (unsigned)t < (unsigned)d.Ti && (unsigned)h < (unsigned)d.Hi && (unsigned)w < (unsigned)d.Wi;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};   // F.pad zero (pytorch_i3d.py:44)
                    if (inb) v = *(const f32x4*)(x + ((size_t)((n * d.Ti + t) * d.Hi + h) * d.Wi + w) * d.ldi + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] > best[e]) { best[e] = v[e]; bi[e] = inb ? tap : 255; }
                }
            }
        }
        *(f32x4*)(y + opos * d.ldo + c) = best;
        *(uchar4*)(am + opos * d.C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const pc_pool_desc d, const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                          float* __restrict__ dx, int accum, int64_t total4) {
    const int C4 = d.C / 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        unsigned pos = (unsigned)idx / (unsigned)C4;
        const int c = (int)((unsigned)idx - pos * C4) * 4;
        const size_t ipos = pos;
        const int w = (int)(pos % (unsigned)d.Wi); pos /= (unsigned)d.Wi;
        const int h = (int)(pos % (unsigned)d.Hi); pos /= (unsigned)d.Hi;
        const int t = (int)(pos % (unsigned)d.Ti); const int n = (int)(pos / (unsigned)d.Ti);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows that cover this input position: per dimension the outputs o with 0 <= x + padf - o*s < k
        int lo[3], hi[3];
        const int xs[3] = {t, h, w}, Os[3] = {d.To, d.Ho, d.Wo};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int top = xs[q] + d.padf[q], bot = top - d.k[q] + 1;
            hi[q] = min(top / d.s[q], Os[q] - 1);
            lo[q] = bot <= 0 ? 0 : (bot + d.s[q] - 1) / d.s[q];
        }
        for (int to = lo[0]; to <= hi[0]; ++to) {
            const int a = t + d.padf[0] - to * d.s[0];
            for (int ho = lo[1]; ho <= hi[1]; ++ho) {
                const int b = h + d.padf[1] - ho * d.s[1];
                for (int wo = lo[2]; wo <= hi[2]; ++wo) {
                    const int tap = (a * d.k[1] + b) * d.k[2] + (w + d.padf[2] - wo * d.s[2]);
                    const size_t op = (size_t)((n * d.To + to) * d.Ho + ho) * d.Wo + wo;
                    const uchar4 ix = *(const uchar4*)(am + op * d.C + c);
                    const f32x4 v = *(const f32x4*)(dy + op * d.ldo + c);
                    if (ix.x == tap) g[0] += v[0];
                    if (ix.y == tap) g[1] += v[1];
                    if (ix.z == tap) g[2] += v[2];
                    if (ix.w == tap) g[3] += v[3];
                }
            }
        }
        float* o = dx + ipos * d.ldi + c;
        if (accum) { const f32x4 old = *(const f32x4*)o; g += old; }
        *(f32x4*)o = g;
    }
}

// ------------------------------------------------------------------------------ misc elementwise
__global__ __launch_bounds__(256) void chscale_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ sc, int64_t pos_per_n,
                                                      int C4, int64_t total4, float* __restrict__ y, int ldy, int accum) {
    const int C = C4 * 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const size_t row = (unsigned)idx / (unsigned)C4;
        const int c = (int)((unsigned)idx - (unsigned)row * C4) * 4;
        const size_t n = (unsigned)row / (unsigned)pos_per_n;
        const f32x4 v = *(const f32x4*)(x + row * ldx + c), s = *(const f32x4*)(sc + n * C + c);
        f32x4 o = v * s;
        float* q = y + row * ldy + c;
        if (accum) o += *(const f32x4*)q;
        *(f32x4*)q = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void to_ndhwc_kernel(const T* __restrict__ src, int N, int C, int64_t thw, int W, int Cpad, int flipw,
                                                       float* __restrict__ dst) {
    const int64_t total = (int64_t)N * thw;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t n = idx / thw, pos = idx - n * thw;
        int64_t sp = pos;
        if (flipw) { const int w = (int)(pos % W); sp = pos - w + (W - 1 - w); }
        if (Cpad == 4) {          // the RGB clip (3 -> 4 channels): one 16-byte store per position instead of four 4-byte stores at a 16-byte stride
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = c < C ? (float)src[(n * C + c) * thw + sp] : 0.f;
            *(f32x4*)(dst + idx * 4) = v;
            continue;
        }
        for (int c = 0; c < Cpad; ++c)
            dst[idx * Cpad + c] = c < C ? (float)src[(n * C + c) * thw + sp] : 0.f;
    }
}

__global__ __launch_bounds__(256) void to_ncdhw_kernel(const float* __restrict__ src, int ld, int N, int C, int64_t thw, float* __restrict__ dst) {
    const int64_t total = (int64_t)N * C * thw;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pos = idx % thw; const int64_t nc = idx / thw;
        const int c = (int)(nc % C); const int64_t n = nc / C;
        dst[idx] = src[(n * thw + pos) * ld + c];
    }
}

// dst[b][c][r] (+)= src[b][r][c]   (32x32 LDS tiles; used for every weight / weight-grad re-layout)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int R, int Cc, int64_t sbs, int sld,
                                                        float* __restrict__ dst, int64_t dbs, int dld, int accum) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(size_t)b * sbs + (size_t)r * sld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < R) {
            float* o = dst + (size_t)b * dbs + (size_t)c * dld + r;
            *o = (accum ? *o : 0.f) + tile[tx][i];
        }
    }
}

// Many independent transposes in one launch (the ~160 weight re-layouts at the start of every step): the jobs travel
// by value in the kernel arguments, a block finds its job from the tile prefix sums.
constexpr int TM_JOBS = 40;
struct TJobK { const float* src; float* dst; long long sbs, dbs, sst; int R, Cc, sld, dld, accum, tiles_c, tiles_rc, nsl, tr, pad_; };
struct TPack { TJobK j[TM_JOBS]; int first[TM_JOBS + 1]; int n; };

__global__ __launch_bounds__(256) void transpose_multi_kernel(const TPack pk) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < pk.n && (int)blockIdx.x >= pk.first[k + 1]) ++k;
    const TJobK& J = pk.j[k];
    const int t = blockIdx.x - pk.first[k];
    const int b = t / J.tiles_rc, rem = t - b * J.tiles_rc;
    // tr = rows of a tile: 32, or 8 for a job that sums many K-slice images (four times the blocks, one element per thread)
    const int r0 = (rem / J.tiles_c) * J.tr, c0 = (rem % J.tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (J.nsl <= 1) {
        for (int i = ty; i < J.tr; i += 8) {
            const int r = r0 + i, c = c0 + tx;
            tile[i][tx] = (r < J.R && c < J.Cc) ? J.src[(size_t)b * J.sbs + (size_t)r * J.sld + c] : 0.f;
        }
    } else {
        // src is nsl K-slice images of a split-K weight gradient, sst floats apart (pc_conv_wgrad with ws_slices): added in slice
        // order -- the sum does not depend on which block finished when (no atomics anywhere on the way).  A thread owns four consecutive
        // columns of one row (one 16-byte load per image, sixteen in flight); J.tr = 8: a quarter of the threads' rows, four times the blocks
        const int i = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
        if (i < J.tr) {
            const int r = r0 + i, c = c0 + c4;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            if (r < J.R && c < J.Cc) {                     // src_ld % 4 == 0 (checked by the host): the quad lies inside the row (columns >= Cc are padding, never stored)
                const f32x4* q = (const f32x4*)(J.src + (size_t)b * J.sbs + (size_t)r * J.sld + c);
                const size_t st4 = (size_t)J.sst / 4;
                int k = 0;
                for (; k + 16 <= J.nsl; k += 16) {
                    f32x4 t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = q[(size_t)(k + u) * st4];
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum += t[u];
                }
                for (; k + 4 <= J.nsl; k += 4) {
                    f32x4 t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = q[(size_t)(k + u) * st4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum += t[u];
                }
                for (; k < J.nsl; ++k) sum += q[(size_t)k * st4];
            }
            tile[i][c4] = sum[0]; tile[i][c4 + 1] = sum[1]; tile[i][c4 + 2] = sum[2]; tile[i][c4 + 3] = sum[3];
        }
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (tx < J.tr && c < J.Cc && r < J.R) {
            float* o = J.dst + (size_t)b * J.dbs + (size_t)c * J.dld + r;
            *o = (J.accum ? *o : 0.f) + tile[tx][i];
        }
    }
}

// K-slice images of an ordered split-K weight gradient, folded in place: image g * group of every group of `group` consecutive images
// becomes the sum of the group's images, added in slice order.  One thread per float4 of the image and group: `group` independent loads.
template <int GROUP>
__global__ __launch_bounds__(256) void slices_fold_kernel(float* __restrict__ ws, int64_t image4, int nslices, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t e = idx % image4; const int g = (int)(idx / image4);
        float4* base = (float4*)ws + (int64_t)g * GROUP * image4 + e;
        const int n = min(GROUP, nslices - g * GROUP);
        float4 t[GROUP];
#pragma unroll
        for (int k = 0; k < GROUP; ++k) if (k < n) t[k] = base[(int64_t)k * image4];
        float4 sum = t[0];
#pragma unroll
        for (int k = 1; k < GROUP; ++k) if (k < n) { sum.x += t[k].x; sum.y += t[k].y; sum.z += t[k].z; sum.w += t[k].w; }
        base[0] = sum;
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, int64_t n, float a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] += a * x[i];
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_bc1, float b1, float b2, float omb1, float omb2, float eps, float inv_sqrt_bc2,
                                                   float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// dx[n,h,w,c] (+)= sum_{b,c'} cols[n, h-b, w-c'][(b*KW+c')*C + c]   (valid (h-b, w-c') only): the gather half of an exact
// 'full' 2-D correlation computed as GEMM (cols = dY x W^T, one column block per tap) + col2im.
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ cols, int N, int Ho, int Wo, int KH, int KW, int C4,
                                                     float* __restrict__ dx, int lddx, int accum, int64_t total4) {
    const int Hi = Ho + KH - 1, Wi = Wo + KW - 1, C = C4 * 4;
    const size_t ldc = (size_t)KH * KW * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        int64_t pos = idx / C4;
        const int c = (int)(idx - pos * C4) * 4;
        const int64_t ipos = pos;
        const int w = (int)(pos % Wi); pos /= Wi;
        const int h = (int)(pos % Hi); const int n = (int)(pos / Hi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int b0 = max(0, h - (Ho - 1)), b1 = min(KH - 1, h), c0 = max(0, w - (Wo - 1)), c1 = min(KW - 1, w);
        for (int b = b0; b <= b1; ++b)
            for (int cc = c0; cc <= c1; ++cc) {
                const size_t op = (size_t)((n * Ho + (h - b)) * Wo + (w - cc));
                acc += *(const f32x4*)(cols + op * ldc + (size_t)(b * KW + cc) * C + c);
            }
        float* o = dx + ipos * lddx + c;
        if (accum) acc += *(const f32x4*)o;
        *(f32x4*)o = acc;
    }
}

inline int grid_for(int64_t n, int per_thread = 1) {
    int64_t b = (n + 256ll * per_thread - 1) / (256ll * per_thread);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}
// the double nearest to the shortest decimal string that converts back to exactly this float (0.999f -> 0.999, 0.9f -> 0.9)
inline double pc_decimal_of_float(float x) {
    char buf[40];
    for (int prec = 1; prec <= 9; ++prec) {
        snprintf(buf, sizeof(buf), "%.*g", prec, (double)x);
        const double d = strtod(buf, nullptr);
        if ((float)d == x) return d;
    }
    return (double)x;
}
inline int64_t red_rows_per_block(int64_t rows) {
    int64_t r = (rows + 1023) / 1024;
    if (r < 32) r = 32;
    return r;
}

}  // namespace

// ================================================================================ C ABI
extern "C" int64_t pc_bn_finalize_ws_floats(int nparts_per_group, int groups, int C) {
    return nparts_per_group >= 512 ? (int64_t)groups * FIN_S * 2 * C * 2 + 2 : 0;       // doubles, + room to align
}

extern "C" int pc_bn_finalize_ws(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group,
                                 const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* stat, float* ws, pc_stream s) {
    PC_CHECK_ARG(part && gamma && beta && stat && groups >= 1 && C > 0, "pc_bn_finalize: bad args");
    if (ws && nparts_per_group >= 512) {
        double* w = (double*)(((uintptr_t)ws + 7) & ~(uintptr_t)7);
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, FIN_CL), FIN_S, groups), dim3(FIN_T), 0, (hipStream_t)s, part, nparts_per_group, C, w);
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(C, FIN_CL)), dim3(FIN_T), 0, (hipStream_t)s, (const double*)w, FIN_S, groups, C,
                           (double)count_per_group, gamma, beta, eps, momentum, running_mean, running_var, stat);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(cdiv(C, FIN_CL)), dim3(FIN_T), 0, (hipStream_t)s, part, nparts_per_group, groups, C,
                           (double)count_per_group, gamma, beta, eps, momentum, running_mean, running_var, stat);
    }
    PC_CHECK_LAUNCH("bn_finalize");
    return PC_OK;
}

extern "C" int pc_bn_finalize(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group,
                              const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* stat, pc_stream s) {
    return pc_bn_finalize_ws(part, nparts_per_group, groups, C, count_per_group, gamma, beta, eps, momentum, running_mean, running_var, stat, nullptr, s);
}

// Finalize + apply in ONE launch (bn_fin_apply_kernel): for layers with at most FA_MAX_NPG partial rows per batch group.  Same results as
// pc_bn_finalize followed by pc_bn_apply (the per-channel sums are taken in double in another, equally fixed order).
constexpr int FA_MAX_NPG = 256;
static int fa_rows_per_block(int64_t rpg, int cblocks, int groups) {
    // about 2048 blocks, at least 128 rows each (the partial rows a block re-reads must stay a fraction of the rows it streams), whole 16-row passes
    int64_t rb = (rpg * cblocks * groups + 2047) / 2048;
    if (rb < 128) rb = 128;
    return (int)((rb + 15) / 16 * 16);
}
extern "C" int pc_bn_finalize_apply_ok(int nparts_per_group, int C) { return nparts_per_group >= 1 && nparts_per_group <= FA_MAX_NPG && C % 4 == 0 ? 1 : 0; }

extern "C" int pc_bn_finalize_apply(const float* part, int nparts_per_group, int groups, int C, int64_t count_per_group, const float* gamma, const float* beta,
                                    float eps, float momentum, float* running_mean, float* running_var, float* stat, const float* z, int ldz, int64_t rows,
                                    float* y, int ldy, int relu, pc_stream s) {
    PC_CHECK_ARG(part && gamma && beta && stat && z && y && groups >= 1 && C > 0 && C % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0 && rows % groups == 0,
                 "pc_bn_finalize_apply: bad args (C=%d)", C);
    PC_CHECK_ARG(pc_bn_finalize_apply_ok(nparts_per_group, C), "pc_bn_finalize_apply: %d partial rows per group (at most %d: use pc_bn_finalize + pc_bn_apply)",
                 nparts_per_group, FA_MAX_NPG);
    PC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pc_bn_finalize_apply: running_mean / running_var go together");
    const int64_t rpg = rows / groups;
    PC_CHECK_ARG(rows * (int64_t)(C / 4) < (1ll << 31) && rpg < (1ll << 31), "elementwise kernels index with 32 bits");
    const int cblocks = cdiv(C, FA_CB), rpb = fa_rows_per_block(rpg, cblocks, groups);
    hipLaunchKernelGGL(bn_fin_apply_kernel, dim3((unsigned)cdiv(rpg, rpb), cblocks, groups), dim3(256), 0, (hipStream_t)s, part, nparts_per_group, groups, C,
                       (double)count_per_group, gamma, beta, eps, momentum, running_mean, running_var, stat, z, ldz, y, ldy, relu, (int)rpg, rpb);
    PC_CHECK_LAUNCH("bn_fin_apply");
    return PC_OK;
}

extern "C" int pc_bn_eval_stat(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                               float eps, int C, float* stat, pc_stream s) {
    PC_CHECK_ARG(gamma && beta && running_mean && running_var && stat, "pc_bn_eval_stat: null");
    hipLaunchKernelGGL(bn_eval_stat_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, gamma, beta, running_mean, running_var, eps, C, stat);
    PC_CHECK_LAUNCH("bn_eval_stat");
    return PC_OK;
}

extern "C" int pc_bn_apply(const float* z, int ldz, const float* stat, int C, int64_t rows, int groups, float* y, int ldy,
                           int relu, pc_stream s) {
    PC_CHECK_ARG(z && stat && y && C % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0 && groups >= 1 && rows % groups == 0, "pc_bn_apply: bad args (C=%d)", C);
    const int64_t total4 = rows * (C / 4);
    PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total4, 2)), dim3(256), 0, (hipStream_t)s, z, ldz, stat, C / 4, total4, rows / groups, y, ldy, relu);
    PC_CHECK_LAUNCH("bn_apply");
    return PC_OK;
}

extern "C" int64_t pc_bn_bwd_ws_floats(int64_t rows, int C, int groups) {
    const int64_t rpg = rows / groups;
    const int64_t npg = (rpg + red_rows_per_block(rpg) - 1) / red_rows_per_block(rpg);
    return (npg * groups * 2 + groups * 2) * (int64_t)C + 64;
}

extern "C" int pc_bn_bwd(const float* dy, int lddy, const float* z, int ldz, const float* stat, int C, int64_t rows, int groups,
                         int relu_flags, float* dz, int lddz, float* dgamma, float* dbeta, int accum, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(dy && z && stat && dz && ws && C % 4 == 0 && C <= 1024 && lddy % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 &&
                 groups >= 1 && rows % groups == 0, "pc_bn_bwd: bad args (C=%d)", C);
    const int64_t rpg = rows / groups;
    // PICONS_BN_FUSED=1 (read per call: a test switch): layers of at most 16 384 rows per group (the 28 x 28 layers) take at most 192 partial rows
    // per group and the finalize folded into the apply kernel (bn_bwd_fin_apply_kernel) -- two launches instead of three on the dependency chain.
    // Built for VERDICT r4 #4b and measured (profiles/r05_switch_ab.txt): the step does not move (19.25 against 19.25 ms) and the family's
    // single-stream time goes UP (1.43 -> 1.54 ms: every block of the apply kernel repeats the reduction over the partial rows), so it is OFF.
    const bool fused = (relu_flags & 2) && rpg <= 16384;          // asked for by the caller (bit 1 of `relu`): the planner's decision, not an environment read per call
    const int relu = relu_flags & 1;
    int64_t rpb = red_rows_per_block(rpg);
    if (fused && (rpg + 191) / 192 > rpb) rpb = (rpg + 191) / 192;
    const int npg = (int)((rpg + rpb - 1) / rpb);
    float* part = ws;
    float* coef = ws + (size_t)npg * groups * 2 * C;
    {   // blockIdx.y = group keeps block row ranges inside the group
        RedP p;
        p.a = dy; p.lda = lddy; p.b = z; p.ldb = ldz;
        p.stat = stat; p.C4 = C / 4; p.rows = rpg; p.rows_per_group = rpg; p.rows_per_block = rpb;
        p.act = relu ? PC_ACT_RELU : PC_ACT_NONE; p.part = part;
        p.a_gs = rpg * lddy; p.b_gs = rpg * ldz; p.stat_gs = 4 * C; p.part_gs = (int64_t)npg * 2 * C;
        hipLaunchKernelGGL(colreduce_kernel<0>, dim3(npg, groups), dim3(256), 0, s, p);
    }
    PC_CHECK_LAUNCH("bn_bwd reduce");
    if (fused) {
        PC_CHECK_ARG(rows * (int64_t)(C / 4) < (1ll << 31), "elementwise kernels index with 32 bits");
        const int cblocks = cdiv(C, FA_CB), rb2 = fa_rows_per_block(rpg, cblocks, groups);
        hipLaunchKernelGGL(bn_bwd_fin_apply_kernel, dim3((unsigned)cdiv(rpg, rb2), cblocks, groups), dim3(256), 0, s, part, npg, groups, C, (double)rpg, dy, lddy, z, ldz,
                           stat, relu, dz, lddz, dgamma, dbeta, accum, (int)rpg, rb2);
        PC_CHECK_LAUNCH("bn_bwd fin_apply");
        return PC_OK;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, FIN_CL)), dim3(FIN_T), 0, s, part, npg, groups, C, (double)rpg, coef, dgamma, dbeta, accum);
    const int64_t total4 = rows * (C / 4);
    PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4, 2)), dim3(256), 0, s, dy, lddy, z, ldz, stat, coef, C / 4, total4, rpg, relu, dz, lddz);
    PC_CHECK_LAUNCH("bn_bwd apply");
    return PC_OK;
}

extern "C" int pc_maxpool_fwd(const pc_pool_desc* d, const float* x, float* y, uint8_t* argmax, pc_stream s) {
    PC_CHECK_ARG(d && x && y && argmax && d->C % 4 == 0 && d->ldi % 4 == 0 && d->ldo % 4 == 0, "pc_maxpool_fwd: bad args");
    PC_CHECK_ARG(d->k[0] * d->k[1] * d->k[2] < 255, "pc_maxpool_fwd: window too large");
    const int64_t total4 = (int64_t)d->N * d->To * d->Ho * d->Wo * (d->C / 4);
    PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, x, y, argmax, total4);
    PC_CHECK_LAUNCH("maxpool_fwd");
    return PC_OK;
}

extern "C" int pc_maxpool_bwd(const pc_pool_desc* d, const float* dy, const uint8_t* argmax, float* dx, int accum, pc_stream s) {
    PC_CHECK_ARG(d && dy && dx && argmax && d->C % 4 == 0 && d->ldi % 4 == 0 && d->ldo % 4 == 0, "pc_maxpool_bwd: bad args");
    const int64_t total4 = (int64_t)d->N * d->Ti * d->Hi * d->Wi * (d->C / 4);
    PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, *d, dy, argmax, dx, accum, total4);
    PC_CHECK_LAUNCH("maxpool_bwd");
    return PC_OK;
}

extern "C" int pc_channel_scale(const float* x, int ldx, const float* scale, int N, int64_t pos_per_n, int C, float* y, int ldy,
                                int accum, pc_stream s) {
    PC_CHECK_ARG(x && scale && y && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "pc_channel_scale: bad args");
    const int64_t total4 = (int64_t)N * pos_per_n * (C / 4);
    PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
    hipLaunchKernelGGL(chscale_kernel, dim3(grid_for(total4, 2)), dim3(256), 0, (hipStream_t)s, x, ldx, scale, pos_per_n, C / 4, total4, y, ldy, accum);
    PC_CHECK_LAUNCH("channel_scale");
    return PC_OK;
}

extern "C" int64_t pc_act_bwd_ws_floats(int64_t rows, int C) {
    const int64_t rpb = red_rows_per_block(rows);
    return ((rows + rpb - 1) / rpb) * 2 * (int64_t)C + 64;
}

extern "C" int pc_act_bwd(const float* dy, int lddy, const float* y, int ldy, int act, int C, int64_t rows, float* dz, int lddz,
                          float* dbias, int accum, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(dy && y && C % 4 == 0 && C <= 1024 && lddy % 4 == 0 && ldy % 4 == 0, "pc_act_bwd: bad args (C=%d)", C);
    if (dbias) {
        PC_CHECK_ARG(ws, "pc_act_bwd: ws required for dbias");
        const int64_t rpb = red_rows_per_block(rows);
        const int nblk = (int)((rows + rpb - 1) / rpb);
        RedP p;
        p.a = dy; p.lda = lddy; p.b = y; p.ldb = ldy; p.stat = nullptr; p.C4 = C / 4; p.rows = rows; p.rows_per_group = rows;
        p.rows_per_block = rpb; p.act = act; p.part = ws;
        p.a_gs = p.b_gs = p.stat_gs = p.part_gs = 0;
        hipLaunchKernelGGL(colreduce_kernel<1>, dim3(nblk), dim3(256), 0, s, p);
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, FIN_CL)), dim3(FIN_T), 0, s, ws, nblk, C, dbias, accum);
    }
    if (dz && (act != PC_ACT_NONE || dz != dy)) {
        PC_CHECK_ARG(lddz % 4 == 0, "pc_act_bwd: lddz");
        const int64_t total4 = rows * (C / 4);
        PC_CHECK_ARG(total4 < (1ll << 31), "elementwise kernels index with 32 bits: %lld float4 elements", (long long)total4);
        hipLaunchKernelGGL(act_bwd_apply_kernel, dim3(grid_for(total4, 2)), dim3(256), 0, s, dy, lddy, y, ldy, act, C / 4, total4, dz, lddz);
    }
    PC_CHECK_LAUNCH("act_bwd");
    return PC_OK;
}

extern "C" int pc_ncdhw_to_ndhwc(const void* src, int src_is_f64, int N, int C, int64_t thw, int W, int Cpad, int flipw, float* dst, pc_stream s) {
    PC_CHECK_ARG(src && dst && Cpad >= C && W > 0 && thw % W == 0, "pc_ncdhw_to_ndhwc: bad args");
    const int64_t total = (int64_t)N * thw;
    if (src_is_f64) hipLaunchKernelGGL(to_ndhwc_kernel<double>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const double*)src, N, C, thw, W, Cpad, flipw, dst);
    else hipLaunchKernelGGL(to_ndhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const float*)src, N, C, thw, W, Cpad, flipw, dst);
    PC_CHECK_LAUNCH("to_ndhwc");
    return PC_OK;
}

extern "C" int pc_ndhwc_to_ncdhw(const float* src, int ld, int N, int C, int64_t thw, float* dst, pc_stream s) {
    PC_CHECK_ARG(src && dst, "pc_ndhwc_to_ncdhw: null");
    hipLaunchKernelGGL(to_ncdhw_kernel, dim3(grid_for((int64_t)N * C * thw)), dim3(256), 0, (hipStream_t)s, src, ld, N, C, thw, dst);
    PC_CHECK_LAUNCH("to_ncdhw");
    return PC_OK;
}

extern "C" int pc_transpose_batched(const float* src, int batch, int R, int Cc, int64_t src_batch_stride, int src_ld, float* dst,
                                    int64_t dst_batch_stride, int dst_ld, int accum, pc_stream s) {
    PC_CHECK_ARG(src && dst && batch >= 1 && batch <= 65535 && R >= 1 && Cc >= 1, "pc_transpose_batched: bad args");
    PC_CHECK_ARG(cdiv(R, 32) <= 65535, "pc_transpose_batched: too many row tiles");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), batch), dim3(256), 0, (hipStream_t)s, src, R, Cc, src_batch_stride,
                       src_ld, dst, dst_batch_stride, dst_ld, accum);
    PC_CHECK_LAUNCH("transpose");
    return PC_OK;
}

extern "C" int pc_transpose_multi(const pc_transpose_job* jobs, int njobs, pc_stream s) {
    PC_CHECK_ARG(jobs && njobs >= 1, "pc_transpose_multi: bad args");
    for (int j0 = 0; j0 < njobs; j0 += TM_JOBS) {
        TPack pk;
        pk.n = njobs - j0 < TM_JOBS ? njobs - j0 : TM_JOBS;
        int tiles = 0;
        for (int q = 0; q < pk.n; ++q) {
            const pc_transpose_job& a = jobs[j0 + q];
            PC_CHECK_ARG(a.src && a.dst && a.batch >= 1 && a.R >= 1 && a.C >= 1, "pc_transpose_multi: bad job %d", j0 + q);
            TJobK& k = pk.j[q];
            k.src = (const float*)(uintptr_t)a.src; k.dst = (float*)(uintptr_t)a.dst; k.sbs = a.src_batch_stride; k.dbs = a.dst_batch_stride;
            PC_CHECK_ARG(a.nslices >= 0 && (a.nslices <= 1 || a.slice_stride > 0), "pc_transpose_multi: job %d has %d slices %lld floats apart", j0 + q, a.nslices, (long long)a.slice_stride);
            k.R = a.R; k.Cc = a.C; k.sld = a.src_ld; k.dld = a.dst_ld; k.accum = a.accum; k.nsl = a.nslices; k.sst = a.slice_stride;
            PC_CHECK_ARG(a.nslices <= 1 || (a.C <= a.src_ld && a.src_ld % 4 == 0 && a.slice_stride % 4 == 0 && a.src_batch_stride % 4 == 0 && a.src % 16 == 0),
                         "pc_transpose_multi: job %d sums slice images: src_ld (>= C) and the strides must be multiples of 4 floats and src 16-byte aligned", j0 + q);
            // 32-row tiles (one 16-byte quad per thread and image, 256 threads busy) unless the job is too small to give every CU a block: then
            // 8-row tiles, four times the blocks (the sum over many images is a chain of dependent loads per thread)
            k.tr = (a.nslices > 1 && (long long)a.batch * cdiv(a.C, 32) * cdiv(a.R, 32) < 192) ? 8 : 32; k.pad_ = 0;
            k.tiles_c = cdiv(a.C, 32); k.tiles_rc = k.tiles_c * cdiv(a.R, k.tr);
            pk.first[q] = tiles;
            tiles += a.batch * k.tiles_rc;
        }
        pk.first[pk.n] = tiles;
        hipLaunchKernelGGL(transpose_multi_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)s, pk);
        PC_CHECK_LAUNCH("transpose_multi");
    }
    return PC_OK;
}

extern "C" int pc_wgrad_fold_group(void) { return 16; }

extern "C" int pc_wgrad_fold(float* ws, int64_t image_floats, int nslices, pc_stream s) {
    PC_CHECK_ARG(ws && image_floats > 0 && image_floats % 4 == 0 && nslices >= 1 && (uintptr_t)ws % 16 == 0, "pc_wgrad_fold: bad args (image %lld floats, %d slices)", (long long)image_floats, nslices);
    constexpr int GROUP = 16;
    if (nslices <= 1) return PC_OK;
    const int64_t total = (image_floats / 4) * cdiv(nslices, GROUP);
    hipLaunchKernelGGL(slices_fold_kernel<GROUP>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, ws, image_floats / 4, nslices, total);
    PC_CHECK_LAUNCH("slices_fold");
    return PC_OK;
}

extern "C" int pc_col2im(const float* cols, int N, int Ho, int Wo, int KH, int KW, int C, float* dx, int lddx, int accum, pc_stream s) {
    PC_CHECK_ARG(cols && dx && C % 4 == 0 && lddx % 4 == 0, "pc_col2im: bad args");
    const int64_t total4 = (int64_t)N * (Ho + KH - 1) * (Wo + KW - 1) * (C / 4);
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, cols, N, Ho, Wo, KH, KW, C / 4, dx, lddx, accum, total4);
    PC_CHECK_LAUNCH("col2im");
    return PC_OK;
}

extern "C" int pc_fill(float* p, int64_t n, float v, pc_stream s) {
    PC_CHECK_ARG(p || n == 0, "pc_fill: null");
    if (n == 0) return PC_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)s, p, n, v);
    PC_CHECK_LAUNCH("fill");
    return PC_OK;
}

extern "C" int pc_axpy(float* y, const float* x, int64_t n, float a, pc_stream s) {
    PC_CHECK_ARG((y && x) || n == 0, "pc_axpy: null");
    if (n == 0) return PC_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)s, y, x, n, a);
    PC_CHECK_LAUNCH("axpy");
    return PC_OK;
}

extern "C" int pc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step,
                            float gscale, pc_stream s) {
    PC_CHECK_ARG(p && g && m && v && step >= 1 && n >= 0, "pc_adam_step: bad args");
    if (n == 0) return PC_OK;                  // an un-armed early-Adam op of the backward list
    // torch.optim.Adam (main_ucf101.py:416) holds its betas as Python doubles: 1 - beta and the bias corrections 1 - beta^t are computed in
    // double and only then rounded to fp32 (1 - 0.999 -> 0.001f; in fp32, 1.f - 0.999f = 0.00100004673: 4.7e-5 off in every second-moment
    // increment).  The ABI carries floats, so the double the caller meant is recovered as the shortest decimal that round-trips the float.
    const double b1d = pc_decimal_of_float(b1), b2d = pc_decimal_of_float(b2);
    const double bc1 = 1.0 - pow(b1d, step), bc2 = 1.0 - pow(b2d, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, (float)(lr / bc1), b1, b2,
                       (float)(1.0 - b1d), (float)(1.0 - b2d), eps, (float)(1.0 / sqrt(bc2)), gscale);
    PC_CHECK_LAUNCH("adam");
    return PC_OK;
}
