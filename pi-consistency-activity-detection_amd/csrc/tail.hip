// Collapsed decoder tail for gfx950.  The reference ends with three LINEAR ops
//   upsample4 = ConvTranspose3d(128->128,k3,s2) -> Dropout3d (per-(sample,channel) scale) -> smooth =
//   ConvTranspose3d(128->1,k3,p1)          (/root/reference/models/capsules_ucf101.py:504-509)
// with no non-linearity in between, so  smooth(drop(upsample4(x)))  ==  tapsum( convT(x, Wc[n]) + bc[n] )  with the
// per-sample combined weight  Wc[n][ci][tap][j] = sum_co W4[ci][co][tap] * cs[n][co] * Wp[co][j]  (j = the 27 smooth taps).
// That is a 128->27 transposed conv instead of 128->128 followed by 128->27: 4x fewer FLOPs in forward, dgrad and wgrad
// and the 205 MB/clip upsample4 activation is never materialised.  These kernels build Wc / bc and map
// the gradient of Wc back onto upsample4.{weight,bias} and smooth.{weight,bias}.
#include "common.h"

namespace {

constexpr int J32 = 32;   // smooth taps padded to one 32-wide MFMA column tile

// Wt [N][Ci][taps][32] (dgrad layout [O=ci][taps][I=j]),  Wf [N][32][taps][Ci] (forward layout [O=j][taps][I=ci])
__global__ __launch_bounds__(256) void tail_combine_kernel(const float* __restrict__ W4, const float* __restrict__ b4, const float* __restrict__ cs,
                                                           const float* __restrict__ Wp, int N, int Ci, int Co, int taps, int J,
                                                           float* __restrict__ Wt, float* __restrict__ Wf, float* __restrict__ bc) {
    const int64_t total = (int64_t)N * Ci * taps * J32;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx & 31);
        int64_t r = idx >> 5;
        const int tap = (int)(r % taps); r /= taps;
        const int ci = (int)(r % Ci); const int n = (int)(r / Ci);
        float acc = 0.f;
        if (j < J) {
            const float* w4 = W4 + (size_t)ci * Co * taps + tap;
            for (int co = 0; co < Co; ++co) {
                const float sc = cs ? cs[n * Co + co] : 1.f;
                acc += w4[(size_t)co * taps] * sc * Wp[co * J + j];
            }
        }
        Wt[idx] = acc;
        Wf[(((size_t)n * J32 + j) * taps + tap) * Ci + ci] = acc;
    }
    if (blockIdx.x == 0) {
        for (int e = threadIdx.x; e < N * J32; e += 256) {
            const int n = e >> 5, j = e & 31;
            float acc = 0.f;
            if (j < J)
                for (int co = 0; co < Co; ++co) acc += b4[co] * (cs ? cs[n * Co + co] : 1.f) * Wp[co * J + j];
            bc[e] = acc;
        }
    }
}

// s[n][32] += column sums of dproj rows of sample n
__global__ __launch_bounds__(256) void tail_colsum_kernel(const float* __restrict__ dproj, int64_t rows_per_n, int64_t rows_per_block, float* __restrict__ s) {
    __shared__ float sh[8][32];
    const int n = blockIdx.y, j = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows_per_n, r0 + rows_per_block);
    const float* base = dproj + (size_t)n * rows_per_n * J32;
    float acc = 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 8) acc += base[r * J32 + j];
    sh[rl][j] = acc;
    __syncthreads();
    if (rl == 0) {
        float t = 0.f;
        for (int q = 0; q < 8; ++q) t += sh[q][j];
        atomicAdd(s + n * J32 + j, t);
    }
}

// dW4[ci][co][tap] (+)= sum_n sum_j G[n][ci][tap][j] * cs[n][co] * Wp[co][j]
// Block = one (ci, tap): the N x 32 slab G[:, ci, tap, :] sits in LDS (every thread reads the same element: broadcast),
// thread co keeps its row of Wp in registers.
constexpr int DW4_MAXN = 64;
__global__ __launch_bounds__(128) void tail_dw4_kernel(const float* __restrict__ G, const float* __restrict__ cs, const float* __restrict__ Wp,
                                                       int N, int Ci, int Co, int taps, int J, float* __restrict__ dW4, int accum) {
    __shared__ float g[DW4_MAXN][J32];
    const int ci = blockIdx.x / taps, tap = blockIdx.x - ci * taps;
    for (int e = threadIdx.x; e < N * J32; e += 128) g[e >> 5][e & 31] = G[(((size_t)(e >> 5) * Ci + ci) * taps + tap) * J32 + (e & 31)];
    __syncthreads();
    for (int co = threadIdx.x; co < Co; co += 128) {
        float wp[J32];
#pragma unroll
        for (int j = 0; j < J32; ++j) wp[j] = j < J ? Wp[co * J + j] : 0.f;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < J32; ++j) a += g[n][j] * wp[j];
            acc += (cs ? cs[n * Co + co] : 1.f) * a;
        }
        const size_t idx = ((size_t)ci * Co + co) * taps + tap;
        dW4[idx] = (accum ? dW4[idx] : 0.f) + acc;
    }
}

// dWp[co][j] += cs[n][co] * sum_{ci,tap} W4[ci][co][tap] * G[n][ci][tap][j]
// Block = (sample n, 8 input channels): its 8 x taps x 32 slab of G sits in LDS; thread (co, half) accumulates 16 of the 32
// columns over the slab.  part == NULL: one atomic per element (N * Ci/8 adds per element, arrival order).  part != NULL (round 6): the block
// leaves its [Co][32] partial in part[n * Ci/8 + slab] with plain stores and tail_dwp_sum_kernel adds the partials in block order.
constexpr int DWP_CI = 8;
__global__ __launch_bounds__(256) void tail_dwp_kernel(const float* __restrict__ G, const float* __restrict__ W4, const float* __restrict__ cs,
                                                       int Ci, int Co, int taps, int J, float* __restrict__ dWp, float* __restrict__ part) {
    extern __shared__ float gs[];                       // [DWP_CI * taps][32]
    const int n = blockIdx.y, ci0 = blockIdx.x * DWP_CI;
    const int rows = DWP_CI * taps;
    const float* gsrc = G + ((size_t)n * Ci + ci0) * taps * J32;
    for (int e = threadIdx.x; e < rows * J32; e += 256) gs[e] = gsrc[e];
    __syncthreads();
    const int half = threadIdx.x & 1;
    float* pb = part ? part + ((size_t)n * gridDim.x + blockIdx.x) * Co * J32 : nullptr;
    for (int co = threadIdx.x >> 1; co < Co; co += 128) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        for (int r = 0; r < rows; ++r) {
            const int ci = ci0 + r / taps, tap = r - (r / taps) * taps;
            const float w = W4[((size_t)ci * Co + co) * taps + tap];
            const float* gr = gs + r * J32 + half * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += w * gr[q];
        }
        const float sc = cs ? cs[n * Co + co] : 1.f;
        if (pb) {
#pragma unroll
            for (int q = 0; q < 16; q += 4) *(float4*)(pb + co * J32 + half * 16 + q) = make_float4(sc * acc[q], sc * acc[q + 1], sc * acc[q + 2], sc * acc[q + 3]);
            continue;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int j = half * 16 + q;
            if (j < J) atomicAdd(dWp + co * J + j, sc * acc[q]);
        }
    }
}

// dWp[co][j] = (accum ? dWp : 0) + sum over the np block partials [np][Co][32], in block order (16 loads in flight per thread)
__global__ __launch_bounds__(256) void tail_dwp_sum_kernel(const float* __restrict__ part, int np, int Co, int J, float* __restrict__ dWp, int accum) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Co * J32) return;
    const int co = e / J32, j = e - co * J32;
    const size_t st = (size_t)Co * J32;
    float v = 0.f;
    int q = 0;
    for (; q + 16 <= np; q += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = part[(size_t)(q + u) * st + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; q < np; ++q) v += part[(size_t)q * st + e];
    if (j < J) dWp[co * J + j] = (accum ? dWp[co * J + j] : 0.f) + v;
}

// bias-path terms: dWp[co][j] += b4[co]*sb[j], db4[co] (+)= sum_j Wp[co][j]*sb[j], dbp (+)= sum_n s[n][center];  sb[j] = sum_n cs[n][co] s[n][j]
__global__ __launch_bounds__(64) void tail_dbias_kernel(const float* __restrict__ s, const float* __restrict__ b4, const float* __restrict__ cs,
                                                        const float* __restrict__ Wp, int N, int Co, int J, int center, float* __restrict__ dWp,
                                                        float* __restrict__ db4, float* __restrict__ dbp, int accum) {
    const int co = blockIdx.x, j = threadIdx.x & 31;
    if (threadIdx.x >= 32) return;
    float sb = 0.f;
    for (int n = 0; n < N; ++n) sb += (cs ? cs[n * Co + co] : 1.f) * s[n * J32 + j];
    if (j < J) dWp[co * J + j] += b4[co] * sb;
    float v = j < J ? Wp[co * J + j] * sb : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (j == 0) db4[co] = (accum ? db4[co] : 0.f) + v;
    if (co == 0 && j == 0) {
        float t2 = 0.f;
        for (int n = 0; n < N; ++n) t2 += s[n * J32 + center];
        dbp[0] = (accum ? dbp[0] : 0.f) + t2;
    }
}

}  // namespace

extern "C" int pc_tail_combine(const float* W4, const float* b4, const float* cs, const float* Wp, int N, int Ci, int Co, int taps, int J,
                               float* Wt, float* Wf, float* bc, pc_stream s) {
    PC_CHECK_ARG(W4 && b4 && Wp && Wt && Wf && bc && J <= J32 && N >= 1, "pc_tail_combine: bad args");
    const int64_t total = (int64_t)N * Ci * taps * J32;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tail_combine_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, W4, b4, cs, Wp, N, Ci, Co, taps, J, Wt, Wf, bc);
    PC_CHECK_LAUNCH("tail_combine");
    return PC_OK;
}

extern "C" int pc_tail_colsum(const float* dproj, int N, int64_t rows_per_n, float* sums, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(dproj && sums && N >= 1 && N <= 65535, "pc_tail_colsum: bad args");
    (void)hipMemsetAsync(sums, 0, sizeof(float) * N * J32, s);
    int64_t rpb = (rows_per_n + 255) / 256; if (rpb < 64) rpb = 64;
    const int nb = (int)((rows_per_n + rpb - 1) / rpb);
    hipLaunchKernelGGL(tail_colsum_kernel, dim3(nb, N), dim3(256), 0, s, dproj, rows_per_n, rpb, sums);
    PC_CHECK_LAUNCH("tail_colsum");
    return PC_OK;
}

extern "C" int64_t pc_tail_grads_ws_floats(int N, int Ci, int Co) {
    if (N < 1 || Ci < DWP_CI || Ci % DWP_CI || Co < 1) return -1;
    return (int64_t)N * (Ci / DWP_CI) * Co * J32;
}

// ws != NULL (pc_tail_grads_ws_floats floats, no initialisation needed): the smooth-weight gradient's N * Ci/8 block partials are stored and added in
// block order -- no fp32 atomics, bit-identical from run to run.  ws == NULL: pc_tail_grads (atomics in arrival order).
extern "C" int pc_tail_grads_ws(const float* G, const float* sums, const float* W4, const float* b4, const float* cs, const float* Wp, int N,
                                int Ci, int Co, int taps, int J, int center, float* dW4, float* db4, float* dWp, float* dbp, int accum, float* ws,
                                pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(G && sums && W4 && b4 && Wp && dW4 && db4 && dWp && dbp && J <= J32, "pc_tail_grads: bad args");
    PC_CHECK_ARG(N <= DW4_MAXN && Ci % DWP_CI == 0, "pc_tail_grads: N <= %d and Ci %% %d == 0 expected (N=%d Ci=%d)", DW4_MAXN, DWP_CI, N, Ci);
    hipLaunchKernelGGL(tail_dw4_kernel, dim3(Ci * taps), dim3(128), 0, s, G, cs, Wp, N, Ci, Co, taps, J, dW4, accum);
    if (!ws && !accum) (void)hipMemsetAsync(dWp, 0, sizeof(float) * Co * J, s);
    hipLaunchKernelGGL(tail_dwp_kernel, dim3(Ci / DWP_CI, N), dim3(256), (size_t)DWP_CI * taps * J32 * 4, s, G, W4, cs, Ci, Co, taps, J, dWp, ws);
    if (ws) hipLaunchKernelGGL(tail_dwp_sum_kernel, dim3((Co * J32 + 255) / 256), dim3(256), 0, s, ws, N * (Ci / DWP_CI), Co, J, dWp, accum);
    hipLaunchKernelGGL(tail_dbias_kernel, dim3(Co), dim3(64), 0, s, sums, b4, cs, Wp, N, Co, J, center, dWp, db4, dbp, accum);
    PC_CHECK_LAUNCH("tail_grads");
    return PC_OK;
}

extern "C" int pc_tail_grads(const float* G, const float* sums, const float* W4, const float* b4, const float* cs, const float* Wp, int N,
                             int Ci, int Co, int taps, int J, int center, float* dW4, float* db4, float* dWp, float* dbp, int accum, pc_stream s) {
    return pc_tail_grads_ws(G, sums, W4, b4, cs, Wp, N, Ci, Co, taps, J, center, dW4, db4, dWp, dbp, accum, nullptr, s);
}
