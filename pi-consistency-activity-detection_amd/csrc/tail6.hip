// Merged decoder tail (capsules_ucf101.py:504-509): upsample4 (ConvTranspose3d 128->128, k3 s2 p1 op1) -> Dropout3d ->
// smooth (ConvTranspose3d 128->1, k3 s1 p1) composes, per dimension, into ONE stride-2 transposed conv with five taps
// k5 = k4 + ks (o = 2i - 2 + k5) and a single output channel.  The only term the plain composition gets wrong is the
// one that goes through the cropped-away position m = -1 of upsample4's output (i = 0, k4 = 0), which only shows up in
// tap k5 = 2 of input index 0; a sixth column per dimension ("2'": k5 = 2 without that term) is used there instead.
// So:  cols[n][i][c6] = x[n][i][:] . W6[n][:][c6]   (one grouped 1x1 GEMM, 6^3 = 216 columns, 128 -> 216: 44 GFLOP instead of
// the 177 GFLOP of the 27-channel form), then a gather of <= 27 column entries per output voxel.  Backward is the
// transposed pair: scatter dout into dcols, two GEMMs (dx, dW6), and the map of dW6 back onto the 27x27 combined
// weights the existing pc_tail_grads consumes.  This file holds the small kernels around those GEMMs.
#include "common.h"

namespace {

constexpr int C6 = 216, C6P = 224, J32 = 32;

// per dimension: which (k4, ks) pairs make up column c (0..4: k4 + ks = c; 5: k4 + ks = 2 without k4 = 0)
__device__ __forceinline__ int npairs(int c) { return c == 0 || c == 4 ? 1 : (c == 2 ? 3 : 2); }
__device__ __forceinline__ int pair_k4(int c, int q) {
    switch (c) {
        case 0: return 0;
        case 1: return q;            // (0,1) (1,0)
        case 2: return q;            // (0,2) (1,1) (2,0)
        case 3: return 1 + q;        // (1,2) (2,1)
        case 4: return 2;
        default: return 1 + q;       // (1,1) (2,0)
    }
}
__device__ __forceinline__ int pair_ks(int c, int q) { return (c == 5 ? 2 : c) - pair_k4(c, q); }

// wf [N][32 j][27 tap][Ci]  ->  W6f [N][224][Ci] (forward GEMM weights),  W6t [N][Ci][224] (dgrad GEMM weights)
__global__ __launch_bounds__(256) void tail6_weights_kernel(const float* __restrict__ wf, int N, int Ci, float* __restrict__ W6f,
                                                            float* __restrict__ W6t) {
    const int64_t total = (int64_t)N * C6P * Ci;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ci = (int)(idx % Ci);
        const int c6 = (int)((idx / Ci) % C6P);
        const int n = (int)(idx / ((int64_t)Ci * C6P));
        float acc = 0.f;
        if (c6 < C6) {
            const int ct = c6 / 36, ch = (c6 / 6) % 6, cw = c6 % 6;
            const float* base = wf + (size_t)n * J32 * 27 * Ci + ci;
            for (int a = 0; a < npairs(ct); ++a)
                for (int b = 0; b < npairs(ch); ++b)
                    for (int c = 0; c < npairs(cw); ++c) {
                        const int tap = (pair_k4(ct, a) * 3 + pair_k4(ch, b)) * 3 + pair_k4(cw, c);
                        const int j = (pair_ks(ct, a) * 3 + pair_ks(ch, b)) * 3 + pair_ks(cw, c);
                        acc += base[((size_t)j * 27 + tap) * Ci];
                    }
        }
        W6f[idx] = acc;
        W6t[((size_t)n * Ci + ci) * C6P + c6] = acc;
    }
}

// (i, column) pairs contributing to output index o of a dimension with I inputs: 2i - 2 + k5 = o
__device__ __forceinline__ int dim_terms(int o, int I, int (&ii)[3], int (&cc)[3]) {
    int n = 0;
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int t = o + 2 - k5;
        if (t >= 0 && !(t & 1) && (t >> 1) < I) {
            const int i = t >> 1;
            ii[n] = i; cc[n] = (k5 == 2 && i == 0) ? 5 : k5; ++n;
        }
    }
    return n;
}

// out[n][o] = bsm + sum_{ks: o + 1 - ks in grid} bc[n][ks] + sum_terms cols[n][i][c6]
__global__ __launch_bounds__(256) void tail6_gather_kernel(const float* __restrict__ cols, const float* __restrict__ bc, const float* __restrict__ bsm,
                                                           int N, int It, int Ih, int Iw, float* __restrict__ out) {
    const int Ot = 2 * It, Oh = 2 * Ih, Ow = 2 * Iw;
    const int64_t total = (int64_t)N * Ot * Oh * Ow;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ow = (int)(idx % Ow), oh = (int)((idx / Ow) % Oh), ot = (int)((idx / ((int64_t)Ow * Oh)) % Ot), n = (int)(idx / ((int64_t)Ow * Oh * Ot));
    int it[3], ct[3], ih[3], ch[3], iw[3], cw[3];
    const int nt = dim_terms(ot, It, it, ct), nh = dim_terms(oh, Ih, ih, ch), nw = dim_terms(ow, Iw, iw, cw);
    float acc = bsm[0];
    for (int a = 0; a < 3; ++a) {
        const int mt = ot + 1 - a;
        if (mt < 0 || mt >= Ot) continue;
        for (int b = 0; b < 3; ++b) {
            const int mh = oh + 1 - b;
            if (mh < 0 || mh >= Oh) continue;
            for (int c = 0; c < 3; ++c) {
                const int mw = ow + 1 - c;
                if (mw < 0 || mw >= Ow) continue;
                acc += bc[n * J32 + (a * 3 + b) * 3 + c];
            }
        }
    }
    for (int a = 0; a < nt; ++a)
        for (int b = 0; b < nh; ++b) {
            const float* row = cols + ((((size_t)n * It + it[a]) * Ih + ih[b]) * Iw) * C6P + (ct[a] * 6 + ch[b]) * 6;
            for (int c = 0; c < nw; ++c) acc += row[(size_t)iw[c] * C6P + cw[c]];
        }
    out[idx] = acc;
}

// dcols[n][i][c6] = dout[n][o(i, c6)] where that column is in use for input index i and o is inside the grid, else 0
__global__ __launch_bounds__(256) void tail6_scatter_kernel(const float* __restrict__ dout, int N, int It, int Ih, int Iw, float* __restrict__ dcols) {
    const int Ot = 2 * It, Oh = 2 * Ih, Ow = 2 * Iw;
    const int64_t total = (int64_t)N * It * Ih * Iw * (C6P / 4);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q4 = (int)(idx % (C6P / 4));
    const int64_t pos = idx / (C6P / 4);
    const int iw = (int)(pos % Iw), ih = (int)((pos / Iw) % Ih), it = (int)((pos / ((int64_t)Iw * Ih)) % It), n = (int)(pos / ((int64_t)Iw * Ih * It));
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c6 = q4 * 4 + e;
        v[e] = 0.f;
        if (c6 >= C6) continue;
        const int c[3] = {c6 / 36, (c6 / 6) % 6, c6 % 6};
        const int i[3] = {it, ih, iw};
        const int O[3] = {Ot, Oh, Ow};
        int o[3];
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int k5 = c[d] == 5 ? 2 : c[d];
            // column 2 is replaced by 2' (5) at input index 0, and 2' is used nowhere else
            const bool used = c[d] == 5 ? i[d] == 0 : (c[d] == 2 ? i[d] != 0 : true);
            if (!used) ok = false;
            o[d] = 2 * i[d] - 2 + k5;
            if (o[d] < 0 || o[d] >= O[d]) ok = false;
        }
        if (ok) v[e] = dout[(((size_t)n * Ot + o[0]) * Oh + o[1]) * Ow + o[2]];
    }
    *(float4*)(dcols + pos * C6P + q4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// columns that contain component (k4, ks) of a dimension: k4 + ks, and 2' when k4 + ks = 2 and k4 != 0
__device__ __forceinline__ int comp_cols(int k4, int ks, int (&c)[2]) {
    c[0] = k4 + ks;
    if (k4 + ks == 2 && k4 != 0) { c[1] = 5; return 2; }
    return 1;
}

// dW6 [N][Ci][224] -> Gc [N][Ci][27 tap][32 j]
__global__ __launch_bounds__(256) void tail6_wgrad_map_kernel(const float* __restrict__ dW6, int N, int Ci, float* __restrict__ Gc) {
    const int64_t total = (int64_t)N * Ci * 27 * J32;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx & 31);
        const int tap = (int)((idx >> 5) % 27);
        const int64_t nci = idx / (27 * J32);
        float acc = 0.f;
        if (j < 27) {
            const int k4[3] = {tap / 9, (tap / 3) % 3, tap % 3}, ks[3] = {j / 9, (j / 3) % 3, j % 3};
            int ct[2], ch[2], cw[2];
            const int nt = comp_cols(k4[0], ks[0], ct), nh = comp_cols(k4[1], ks[1], ch), nw = comp_cols(k4[2], ks[2], cw);
            const float* row = dW6 + nci * C6P;
            for (int a = 0; a < nt; ++a)
                for (int b = 0; b < nh; ++b)
                    for (int c = 0; c < nw; ++c) acc += row[(ct[a] * 6 + ch[b]) * 6 + cw[c]];
        }
        Gc[idx] = acc;
    }
}

// sums[n][j = ks] = sum over the outputs o whose tap-ks source position o + 1 - ks lies in the grid of dout[n][o]
__global__ __launch_bounds__(256) void tail6_bias_sums_kernel(const float* __restrict__ dout, int Ot, int Oh, int Ow, int64_t per_block, float* __restrict__ sums) {
    __shared__ float sh[27][8];
    const int n = blockIdx.y;
    const int64_t per_n = (int64_t)Ot * Oh * Ow;
    const int64_t r0 = (int64_t)blockIdx.x * per_block, r1 = min(per_n, r0 + per_block);
    float acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const int ow = (int)(r % Ow), oh = (int)((r / Ow) % Oh), ot = (int)(r / ((int64_t)Ow * Oh));
        const float v = dout[(size_t)n * per_n + r];
        // ks = 0 needs o + 1 < O, ks = 2 needs o >= 1
        const float vt[3] = {ot + 1 < Ot ? 1.f : 0.f, 1.f, ot >= 1 ? 1.f : 0.f};
        const float vh[3] = {oh + 1 < Oh ? 1.f : 0.f, 1.f, oh >= 1 ? 1.f : 0.f};
        const float vw[3] = {ow + 1 < Ow ? 1.f : 0.f, 1.f, ow >= 1 ? 1.f : 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[(a * 3 + b) * 3 + c] += v * vt[a] * vh[b] * vw[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        float s = acc[q];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sh[q][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 27) atomicAdd(sums + n * J32 + threadIdx.x, sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

}  // namespace

extern "C" int pc_tail6_weights(const float* wf, int N, int Ci, float* W6f, float* W6t, pc_stream s) {
    PC_CHECK_ARG(wf && W6f && W6t && N >= 1 && Ci >= 1, "pc_tail6_weights: bad args");
    const int64_t total = (int64_t)N * C6P * Ci;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tail6_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, wf, N, Ci, W6f, W6t);
    PC_CHECK_LAUNCH("tail6_weights");
    return PC_OK;
}

extern "C" int pc_tail6_gather(const float* cols, const float* bc, const float* bsm, int N, int It, int Ih, int Iw, float* out, pc_stream s) {
    PC_CHECK_ARG(cols && bc && bsm && out && N >= 1 && It >= 1 && Ih >= 1 && Iw >= 1, "pc_tail6_gather: bad args");
    const int64_t total = (int64_t)N * 8 * It * Ih * Iw;
    PC_CHECK_ARG((total + 255) / 256 < (1ll << 31), "pc_tail6_gather: too large");
    hipLaunchKernelGGL(tail6_gather_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
    PC_CHECK_LAUNCH("tail6_gather");
    return PC_OK;
}

extern "C" int pc_tail6_scatter(const float* dout, int N, int It, int Ih, int Iw, float* dcols, pc_stream s) {
    PC_CHECK_ARG(dout && dcols && N >= 1 && It >= 1 && Ih >= 1 && Iw >= 1, "pc_tail6_scatter: bad args");
    const int64_t total = (int64_t)N * It * Ih * Iw * (C6P / 4);
    PC_CHECK_ARG((total + 255) / 256 < (1ll << 31), "pc_tail6_scatter: too large");
    hipLaunchKernelGGL(tail6_scatter_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, dout, N, It, Ih, Iw, dcols);
    PC_CHECK_LAUNCH("tail6_scatter");
    return PC_OK;
}

extern "C" int pc_tail6_wgrad_map(const float* dW6, int N, int Ci, float* Gc, pc_stream s) {
    PC_CHECK_ARG(dW6 && Gc && N >= 1 && Ci >= 1, "pc_tail6_wgrad_map: bad args");
    const int64_t total = (int64_t)N * Ci * 27 * J32;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tail6_wgrad_map_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, dW6, N, Ci, Gc);
    PC_CHECK_LAUNCH("tail6_wgrad_map");
    return PC_OK;
}

extern "C" int pc_tail6_bias_sums(const float* dout, int N, int It, int Ih, int Iw, float* sums, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(dout && sums && N >= 1 && N <= 65535, "pc_tail6_bias_sums: bad args");
    (void)hipMemsetAsync(sums, 0, sizeof(float) * N * J32, s);
    const int64_t per_n = (int64_t)8 * It * Ih * Iw;
    int64_t pb = (per_n + 255) / 256; if (pb < 4096) pb = 4096;
    const int nb = (int)((per_n + pb - 1) / pb);
    hipLaunchKernelGGL(tail6_bias_sums_kernel, dim3(nb, N), dim3(256), 0, s, dout, 2 * It, 2 * Ih, 2 * Iw, pb, sums);
    PC_CHECK_LAUNCH("tail6_bias_sums");
    return PC_OK;
}
