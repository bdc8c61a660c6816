// Winograd F(2x2, 3x3) over (h, w) for the stride-1, pad-1 3x3x3 / 1x3x3 convolutions that dominate the step (the decoder's
// skip convs conv112 / conv56, capsules_ucf101.py:382-384,497,501; Conv3d_2c, pytorch_i3d.py:236-238; and their input
// gradients, which are the same correlation with mirrored taps): 2.25x fewer multiply-accumulates than the gather-GEMM form,
// the temporal taps stay direct.  ONE fused kernel: the 4x4 input patches are transformed in registers on their way into LDS,
// the 16 transform-domain GEMMs run on v_mfma_f32_32x32x2_f32 with all 16 accumulators of a (32 tiles x 32 channels) sub-tile
// in one wave's registers, and the inverse transform is register-local in the epilogue -- the transform-domain tensors (4x the
// activations) never exist in HBM.
//
//   Y = A^T [ sum_ci sum_kt (G g_kt G^T) .* (B^T d_kt B) ] A       per 2x2 output tile, d = 4x4 input patch at (2i-1, 2j-1)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) float w_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct WinoK {
    const float* in; const float* U; const float* bias; float* out; float* bnpart;
    int N, T, H, W, Ci, ldi, Co, ldo;
    int TH, TW, BTH, BTW, nbh, nbw, nct, nc8;
    int KT, act, flags;
};

constexpr int WT = 64;            // tiles per block (rows of the transform-domain GEMMs)
constexpr int WC = 64;            // output channels per block
constexpr int WK = 8;             // input channels per K chunk
constexpr int PLANE = 16 * 2 * 64 * 4;     // floats of one LDS operand image: [xi*4+nu][k half][row 64][4]

// ---- weight transform: U[kt][ct][c8][xi*4+nu][kh][co 64][4] = (G g G^T)[xi][nu] of g = w[o][kt][.][.][i], read through strides so
// the master OIDHW tensor (forward) and its transpose with mirrored taps (input gradient) need no intermediate layout.
__global__ void wino_weights_kernel(const float* __restrict__ w, long long sO, long long sT, long long sI, int O, int I, int KT, int flip,
                                    float* __restrict__ U, int nct, int nc8) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)KT * nct * 64 * I;
    if (e >= total) return;
    const int i = (int)(e % I);
    long long r = e / I;
    const int o = (int)(r % (nct * 64));
    const int kt = (int)(r / (nct * 64));
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ks = flip ? KT - 1 - kt : kt, as = flip ? 2 - a : a, bs = flip ? 2 - b : b;
            g[a][b] = o < O ? w[(long long)o * sO + (long long)((ks * 3 + as) * 3 + bs) * sT + (long long)i * sI] : 0.f;
        }
    float t[4][3];          // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    float* dst = U + ((((long long)kt * nct + o / 64) * nc8 + i / 8) * 16) * (2 * 64 * 4) + ((long long)((i & 7) >> 2) * 64 + (o & 63)) * 4 + (i & 3);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float u0 = t[x][0], u1 = 0.5f * (t[x][0] + t[x][1] + t[x][2]), u2 = 0.5f * (t[x][0] - t[x][1] + t[x][2]), u3 = t[x][2];
        dst[(x * 4 + 0) * (2 * 64 * 4)] = u0;
        dst[(x * 4 + 1) * (2 * 64 * 4)] = u1;
        dst[(x * 4 + 2) * (2 * 64 * 4)] = u2;
        dst[(x * 4 + 3) * (2 * 64 * 4)] = u3;
    }
}

// ---- the fused convolution.  Block = 64 tiles (a BTH x BTW rectangle of 2x2-output tiles of one (n, t) plane) x 64 output channels,
// four waves as 2 (tile halves) x 2 (channel halves), one wave per SIMD with all 16 transform-domain accumulators (256 registers).
// K chunk = 8 input channels of one temporal tap: V (transformed input, 32 KB) is produced by the block itself -- thread =
// (tile, 4-channel half, two of the four B^T rows): 12 global_load_dwordx4, 16 float4 adds, 8 ds_write_b128 -- and U (32 KB, contiguous
// in HBM by construction) arrives by LDS-DMA; both double-buffered, one barrier per chunk.
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(const WinoK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                  // [2][PLANE]
    float* Us = smem + 2 * PLANE;      // [2][PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int sb = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = sb % p.nct; sb /= p.nct;
    const int sblock = sb;             // spatial block id (n, t, bh, bw): the BatchNorm partial row
    const int bw = sb % p.nbw; sb /= p.nbw;
    const int bh = sb % p.nbh; sb /= p.nbh;
    const int t = sb % p.T, n = sb / p.T;

    // transform role: thread = (tile, k half, row pair)
    const int ttile = tid & 63, tkh = (tid >> 6) & 1, thalf = tid >> 7;
    const int tli = ttile / p.BTW, tlj = ttile - tli * p.BTW;
    const int ti = bh * p.BTH + tli, tj = bw * p.BTW + tlj;
    const bool tval = ttile < p.BTH * p.BTW && ti < p.TH && tj < p.TW;
    int poff[12];
    unsigned pm = 0;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int h = 2 * ti - 1 + thalf + rr, w = 2 * tj - 1 + c;
            const bool ok = tval && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            poff[rr * 4 + c] = ok ? (h * p.W + w) * p.ldi + tkh * 4 : 0;
            pm |= (ok ? 1u : 0u) << (rr * 4 + c);
        }
    // temporal taps whose source plane exists: tt = t + kt - (KT >> 1)
    const int tpad = p.KT >> 1;
    const int kt_lo = max(0, tpad - t), kt_hi = min(p.KT - 1, p.T - 1 - t + tpad);
    const int nchunks = (kt_hi - kt_lo + 1) * p.nc8;
    const size_t plane_in = (size_t)p.H * p.W * p.ldi;

    // Chunk c -> (temporal tap, 8-channel slice).  The fetch of chunk c + 1 and its transform are spread over the 16 MFMA groups of chunk
    // c (one wave per SIMD: nothing else hides them): the eight LDS-DMA pieces of U ride in groups 0..1, the twelve patch loads in groups
    // 2..4 and transform row j (one ds_write_b128) in group 8 + j.  The DMA is issued BEFORE the loads because the compiler waits for
    // the loaded registers with s_waitcnt vmcnt(0), i.e. also for every younger LDS-DMA piece.  After the last chunk the same slots
    // re-fetch that chunk (harmless).
    f32x4 d[12];
    const float* pbase = nullptr;
    const float* ug = nullptr;
    float* ul = nullptr;
    float* vb = nullptr;
    auto set_chunk = [&](int c, int buf) {
        const int kt = kt_lo + c / p.nc8, c8 = c - (c / p.nc8) * p.nc8;
        pbase = p.in + ((size_t)n * p.T + (t + kt - tpad)) * plane_in + c8 * WK;
        ug = p.U + (((size_t)kt * p.nct + ct) * p.nc8 + c8) * PLANE + wave * 8 * 256 + lane * 4;
        ul = Us + buf * PLANE + wave * 8 * 256;
        vb = Vs + buf * PLANE + (tkh * 64 + ttile) * 4;
    };
    auto load2 = [&](int g) {
#pragma unroll
        for (int k = 2 * g; k < 2 * g + 2; ++k) {
            const float* src = ((pm >> k) & 1u) ? pbase + poff[k] : w_zero16;
            d[k] = *(const f32x4*)src;
        }
    };
    // transform row j = q * 4 + nu: rows r0, r1, r2 of the patch held by this thread; half 0 -> B^T rows 0, 1 (d0 - d2, d1 + d2),
    // half 1 -> rows 3, 2 (d1 - d3, d2 - d1)
    auto xrow = [&](int q, int c) -> f32x4 {
        if (q == 0) return d[c] - d[8 + c];
        return thalf ? (d[4 + c] - d[c]) : (d[4 + c] + d[8 + c]);
    };
    auto store_row = [&](int j) {
        const int q = j >> 2, nu = j & 3;
        const int xi = thalf ? (q == 0 ? 3 : 2) : q;
        f32x4 v;
        if (nu == 0) v = xrow(q, 0) - xrow(q, 2);
        else if (nu == 1) v = xrow(q, 1) + xrow(q, 2);
        else if (nu == 2) v = xrow(q, 2) - xrow(q, 1);
        else v = xrow(q, 1) - xrow(q, 3);
        *(f32x4*)(vb + (xi * 4 + nu) * 512) = v;
    };

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int kh = lane >> 5;
    const int aoff = (kh * 64 + wm * 32 + (lane & 31)) * 4, boff = (kh * 64 + wn * 32 + (lane & 31)) * 4;
    if (nchunks > 0) {
        set_chunk(0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(ug + j * 256, ul + j * 256);
#pragma unroll
        for (int g = 0; g < 6; ++g) load2(g);
#pragma unroll
        for (int j = 0; j < 8; ++j) store_row(j);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        set_chunk(min(c + 1, nchunks - 1), buf ^ 1);
        const float* va = Vs + buf * PLANE + aoff;
        const float* ub = Us + buf * PLANE + boff;
        f32x4 a0 = *(const f32x4*)va, b0 = *(const f32x4*)ub, a1, b1;
        // one group = the four MFMAs of one transform-domain position; the side work of the group is pinned between them
        // (sched_barrier: the compiler otherwise gathers the loads, waits for them at once and strands the matrix pipe)
#define WINO_GROUP(I, A, B, AN, BN)                                                                            \
        {                                                                                                      \
            if ((I) + 1 < 16) { AN = *(const f32x4*)(va + ((I) + 1) * 512); BN = *(const f32x4*)(ub + ((I) + 1) * 512); } \
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0], B[0], acc[I], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if ((I) < 2) { glds16(ug + (4 * (I)) * 256, ul + (4 * (I)) * 256); glds16(ug + (4 * (I) + 1) * 256, ul + (4 * (I) + 1) * 256); } \
            if ((I) >= 2 && (I) < 5) load2(2 * ((I) - 2));                                                     \
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1], B[1], acc[I], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if ((I) < 2) { glds16(ug + (4 * (I) + 2) * 256, ul + (4 * (I) + 2) * 256); glds16(ug + (4 * (I) + 3) * 256, ul + (4 * (I) + 3) * 256); } \
            if ((I) >= 2 && (I) < 5) load2(2 * ((I) - 2) + 1);                                                 \
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2], B[2], acc[I], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if ((I) >= 8) store_row((I) - 8);                                                                  \
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[3], B[3], acc[I], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        WINO_GROUP(0, a0, b0, a1, b1)   WINO_GROUP(1, a1, b1, a0, b0)   WINO_GROUP(2, a0, b0, a1, b1)   WINO_GROUP(3, a1, b1, a0, b0)
        WINO_GROUP(4, a0, b0, a1, b1)   WINO_GROUP(5, a1, b1, a0, b0)   WINO_GROUP(6, a0, b0, a1, b1)   WINO_GROUP(7, a1, b1, a0, b0)
        WINO_GROUP(8, a0, b0, a1, b1)   WINO_GROUP(9, a1, b1, a0, b0)   WINO_GROUP(10, a0, b0, a1, b1)  WINO_GROUP(11, a1, b1, a0, b0)
        WINO_GROUP(12, a0, b0, a1, b1)  WINO_GROUP(13, a1, b1, a0, b0)  WINO_GROUP(14, a0, b0, a1, b1)  WINO_GROUP(15, a1, b1, a0, b0)
#undef WINO_GROUP
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A per (tile, channel), register-local: accumulator register r is tile row (r&3) + 8*(r>>2) + 4*(lane>>5) of
    // this wave's 32 tiles, the lane's column is the output channel
    const int co = ct * WC + wn * 32 + (lane & 31);
    const bool cval = co < p.Co;
    const float bv = (p.flags & PC_F_BIAS) && cval ? p.bias[co] : 0.f;
    const bool accum = p.flags & PC_F_ACCUM;
    float s1 = 0.f, s2 = 0.f;
    const size_t plane_out = (size_t)p.H * p.W * p.ldo;
    float* obase = p.out + ((size_t)n * p.T + t) * plane_out + co;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int li = m / p.BTW, lj = m - li * p.BTW;
        const int oi = bh * p.BTH + li, oj = bw * p.BTW + lj;
        const bool ok = cval && m < p.BTH * p.BTW && oi < p.TH && oj < p.TW;
        float s[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            s[x][0] = acc[x * 4 + 0][r] + acc[x * 4 + 1][r] + acc[x * 4 + 2][r];
            s[x][1] = acc[x * 4 + 1][r] - acc[x * 4 + 2][r] - acc[x * 4 + 3][r];
        }
        float y[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            y[0][b] = s[0][b] + s[1][b] + s[2][b];
            y[1][b] = s[1][b] - s[2][b] - s[3][b];
        }
        if (!ok) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v = y[a][b] + bv;
                s1 += v; s2 += v * v;
                if (p.act == PC_ACT_RELU) v = fmaxf(v, 0.f);
                float* o = obase + ((size_t)(2 * oi + a) * p.W + 2 * oj + b) * p.ldo;
                if (accum) v += *o;
                *o = v;
            }
    }
    if (p.flags & PC_F_BNPART) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* part = p.bnpart + ((size_t)sblock * 2 + wm) * 2 * p.Co;
        if (lane < 32 && cval) { part[co] = s1; part[p.Co + co] = s2; }
    }
}

void choose_block(int TH, int TW, int& bth, int& btw) {
    double best = 1e30;
    bth = 8; btw = 8;
    for (int w = 1; w <= 64 && w <= TW; ++w) {
        int h = 64 / w;
        if (h > TH) h = TH;
        if (h < 1) continue;
        const double blocks = (double)cdiv(TH, h) * cdiv(TW, w);
        const double waste = blocks * 64.0 / ((double)TH * TW);
        const double aspect = (double)(2 * w + 2) * (2 * h + 2) / (4.0 * w * h);      // patch read amplification: prefer square-ish
        const double cost = waste * (1.0 + 0.05 * aspect);
        if (cost < best - 1e-9) { best = cost; bth = h; btw = w; }
    }
}

int fill(const pc_wino_desc* d, WinoK& k) {
    PC_CHECK_ARG(d, "pc_wino: null descriptor");
    PC_CHECK_ARG(d->N >= 1 && d->T >= 1 && d->H >= 2 && d->W >= 2 && d->H % 2 == 0 && d->W % 2 == 0, "pc_wino: H, W must be even (H=%d W=%d)", d->H, d->W);
    PC_CHECK_ARG(d->Ci >= 8 && d->Ci % 8 == 0 && d->ldi % 4 == 0 && d->Co >= 1 && d->ldo >= d->Co, "pc_wino: Ci %% 8, ldi %% 4 (Ci=%d ldi=%d Co=%d ldo=%d)", d->Ci, d->ldi, d->Co, d->ldo);
    PC_CHECK_ARG(d->KT == 1 || d->KT == 3, "pc_wino: KT must be 1 or 3");
    PC_CHECK_ARG((int64_t)d->N * d->T * d->H * d->W * (int64_t)(d->ldi > d->ldo ? d->ldi : d->ldo) < (1ll << 40), "pc_wino: tensor too large");
    PC_CHECK_ARG((int64_t)d->H * d->W * d->ldi < (1ll << 31), "pc_wino: plane too large");
    k.N = d->N; k.T = d->T; k.H = d->H; k.W = d->W; k.Ci = d->Ci; k.ldi = d->ldi; k.Co = d->Co; k.ldo = d->ldo;
    k.TH = d->H / 2; k.TW = d->W / 2;
    choose_block(k.TH, k.TW, k.BTH, k.BTW);
    k.nbh = cdiv(k.TH, k.BTH); k.nbw = cdiv(k.TW, k.BTW);
    k.nct = cdiv(d->Co, WC); k.nc8 = d->Ci / WK;
    k.KT = d->KT; k.act = d->act; k.flags = d->flags;
    return PC_OK;
}

}  // namespace

extern "C" int64_t pc_wino_u_floats(int O, int I, int KT) {
    if (O < 1 || I < 8 || I % 8 || (KT != 1 && KT != 3)) return -1;
    return (int64_t)KT * cdiv(O, WC) * (I / WK) * PLANE;
}

extern "C" int pc_wino_weights(const float* w, int64_t sO, int64_t sT, int64_t sI, int O, int I, int KT, int flip, float* U, pc_stream s) {
    PC_CHECK_ARG(w && U, "pc_wino_weights: null pointer");
    PC_CHECK_ARG(O >= 1 && I >= 8 && I % 8 == 0 && (KT == 1 || KT == 3), "pc_wino_weights: O=%d I=%d KT=%d", O, I, KT);
    const int nct = cdiv(O, WC), nc8 = I / WK;
    const long long total = (long long)KT * nct * 64 * I;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, (long long)sO, (long long)sT, (long long)sI, O, I, KT, flip, U, nct, nc8);
    PC_CHECK_LAUNCH("wino_weights_kernel");
    return PC_OK;
}

extern "C" int pc_wino_bnpart_rows(const pc_wino_desc* d) {
    WinoK k;
    if (fill(d, k) != PC_OK) return -1;
    return k.N * k.T * k.nbh * k.nbw * 2;
}

// Host-only: multiply-accumulates the launch issues to the matrix cores / performs on real outputs (see pc_conv_work)
extern "C" int pc_wino_work(const pc_wino_desc* d, double* out) {
    WinoK k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(out, "pc_wino_work: null pointer");
    double taps = 0;                                       // valid temporal taps summed over t
    for (int t = 0; t < k.T; ++t)
        for (int a = 0; a < k.KT; ++a) { const int tt = t + a - (k.KT >> 1); taps += tt >= 0 && tt < k.T; }
    const double blocks = (double)k.N * k.nbh * k.nbw * k.nct;
    out[0] = blocks * taps * 16.0 * WT * WC * k.Ci;                                  // issued: 16 transform-domain GEMMs of 64 x 64 x Ci per tap
    out[1] = (double)k.N * taps * 16.0 * ((double)k.TH * k.TW) * k.Co * k.Ci;        // executed on real tiles / channels
    out[2] = blocks * k.T;                                                          // blocks
    return PC_OK;
}

extern "C" int pc_wino_conv(const pc_wino_desc* d, const float* in, const float* U, const float* bias, float* out, float* bnpart, pc_stream s) {
    WinoK k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(in && U && out, "pc_wino_conv: null pointer");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)U % 16 == 0), "pc_wino_conv: in / U must be 16-byte aligned");
    PC_CHECK_ARG(!(d->flags & PC_F_BIAS) || bias, "pc_wino_conv: bias flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_BNPART) || bnpart, "pc_wino_conv: bnpart flag without pointer");
    PC_CHECK_ARG(!(d->flags & ~(PC_F_BIAS | PC_F_ACCUM | PC_F_BNPART)), "pc_wino_conv: unsupported flag");
    PC_CHECK_ARG(d->act == PC_ACT_NONE || d->act == PC_ACT_RELU, "pc_wino_conv: activation");
    k.in = in; k.U = U; k.bias = bias; k.out = out; k.bnpart = bnpart;
    static bool attr_set = false;
    const size_t lds = (size_t)4 * PLANE * sizeof(float);
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wino_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid((unsigned)((int64_t)k.N * k.T * k.nbh * k.nbw * k.nct));
    if (pc_tl_ev_start) hipExtLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), lds, (hipStream_t)s, pc_tl_ev_start, pc_tl_ev_stop, 0, k);
    else hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), lds, (hipStream_t)s, k);
    PC_CHECK_LAUNCH("wino_conv_kernel");
    return PC_OK;
}
