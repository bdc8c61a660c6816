// Merged decoder tail (capsules_ucf101.py:504-509): upsample4 (ConvTranspose3d 128->128, k3 s2 p1 op1) -> Dropout3d ->
// smooth (ConvTranspose3d 128->1, k3 s1 p1) composes, per dimension, into ONE stride-2 transposed conv with five taps
// k5 = k4 + ks (o = 2i - 2 + k5) and a single output channel.  The only term the plain composition gets wrong is the
// one that goes through the cropped-away position m = -1 of upsample4's output (i = 0, k4 = 0), which only shows up in
// tap k5 = 2 of input index 0: positions whose index is 0 in a dimension use a weight for that tap without that term.
// So the input positions fall into 8 classes z = (i_t == 0, i_h == 0, i_w == 0), each with its own 128 x 125 weight
// matrix, and   cols[n][i][slot] = x[n][i][:] . W5[n][z(i)][:][slot],  slot = (k5_t*5 + k5_h)*5 + k5_w   (125 columns,
// padded to 128: 26 GFLOP per pass instead of the 177 GFLOP of the 27-channel form) is a grouped 1x1 GEMM per class over
// its sub-lattice of positions, followed by a gather of the <= 27 column entries that land on each output voxel.
// Backward is the transposed pair: scatter dout into dcols, two GEMMs per class (dx, dW5), and the map of dW5 back onto
// the 27x27 combined weights the existing pc_tail_grads consumes.  This file holds the small kernels around those GEMMs.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NSLOT = 125, SP = 128, J32 = 32;

// per dimension: the (k4, ks) pairs of tap k5 (k4 + ks = k5); `first` (input index 0) drops the k4 = 0 pair of k5 = 2
__device__ __forceinline__ int npairs(int k5, bool first) { return k5 == 0 || k5 == 4 ? 1 : (k5 == 2 ? (first ? 2 : 3) : 2); }
__device__ __forceinline__ int pair_k4(int k5, bool first, int q) {
    switch (k5) {
        case 0: return 0;
        case 1: return q;                    // (0,1) (1,0)
        case 2: return first ? 1 + q : q;    // (0,2) (1,1) (2,0)  /  (1,1) (2,0)
        case 3: return 1 + q;                // (1,2) (2,1)
        default: return 2;
    }
}

// wf [N][32 j][27 tap][Ci]  ->  W5f [N][8 z][128 slot][Ci] (forward GEMM weights),  W5t [N][8][Ci][128] (dgrad GEMM weights)
__global__ __launch_bounds__(256) void tail6_weights_kernel(const float* __restrict__ wf, int N, int Ci, float* __restrict__ W5f,
                                                            float* __restrict__ W5t) {
    const int64_t total = (int64_t)N * 8 * SP * Ci;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ci = (int)(idx % Ci);
        const int slot = (int)((idx / Ci) % SP);
        const int z = (int)((idx / ((int64_t)Ci * SP)) % 8);
        const int n = (int)(idx / ((int64_t)Ci * SP * 8));
        float acc = 0.f;
        if (slot < NSLOT) {
            const int st = slot / 25, sh = (slot / 5) % 5, sw = slot % 5;
            const bool ft = z & 4, fh = z & 2, fw = z & 1;
            const float* base = wf + (size_t)n * J32 * 27 * Ci + ci;
            for (int a = 0; a < npairs(st, ft); ++a)
                for (int b = 0; b < npairs(sh, fh); ++b)
                    for (int c = 0; c < npairs(sw, fw); ++c) {
                        const int ka = pair_k4(st, ft, a), kb = pair_k4(sh, fh, b), kc = pair_k4(sw, fw, c);
                        const int tap = (ka * 3 + kb) * 3 + kc;
                        const int j = ((st - ka) * 3 + (sh - kb)) * 3 + (sw - kc);
                        acc += base[((size_t)j * 27 + tap) * Ci];
                    }
        }
        W5f[idx] = acc;
        W5t[(((size_t)n * 8 + z) * Ci + ci) * SP + slot] = acc;
    }
}

// (i, k5) pairs contributing to output index o of a dimension with I inputs: 2i - 2 + k5 = o
__device__ __forceinline__ int dim_terms(int o, int I, int (&ii)[3], int (&kk)[3]) {
    int n = 0;
#pragma unroll
    for (int k5 = 0; k5 < 5; ++k5) {
        const int t = o + 2 - k5;
        if (t >= 0 && !(t & 1) && (t >> 1) < I) { ii[n] = t >> 1; kk[n] = k5; ++n; }
    }
    return n;
}

// out[n][o] = bsm + sum_{ks: o + 1 - ks in grid} bc[n][ks] + sum_terms colsT[n][slot][i]
// colsT is channel-major (the column GEMMs run with PC_F_TOUT): [n][128 slots][It][Ih][Iw].  Every (i, slot) entry lands on exactly
// one output voxel, and for a fixed slot consecutive outputs of one w parity read consecutive iw: the gather streams the 411 MB
// once (row-major columns cost 5x that: 4 bytes out of each of 27 different 512-byte rows per output).
__global__ __launch_bounds__(256) void tail6_gather_kernel(const float* __restrict__ cols, const float* __restrict__ bc, const float* __restrict__ bsm,
                                                           int N, int It, int Ih, int Iw, float* __restrict__ out) {
    const int Ot = 2 * It, Oh = 2 * Ih, Ow = 2 * Iw;
    const int64_t total = (int64_t)N * Ot * Oh * Ow;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ow = (int)(idx % Ow), oh = (int)((idx / Ow) % Oh), ot = (int)((idx / ((int64_t)Ow * Oh)) % Ot), n = (int)(idx / ((int64_t)Ow * Oh * Ot));
    int it[3], kt[3], ih[3], kh[3], iw[3], kw[3];
    const int nt = dim_terms(ot, It, it, kt), nh = dim_terms(oh, Ih, ih, kh), nw = dim_terms(ow, Iw, iw, kw);
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
    const size_t P3 = (size_t)It * Ih * Iw;
    const float* base = cols + (size_t)n * SP * P3;
    for (int a = 0; a < nt; ++a)
        for (int b = 0; b < nh; ++b) {
            const size_t pos = ((size_t)it[a] * Ih + ih[b]) * Iw;
            const int slot0 = (kt[a] * 5 + kh[b]) * 5;
            for (int c = 0; c < nw; ++c) acc += base[(size_t)(slot0 + kw[c]) * P3 + pos + iw[c]];
        }
    out[idx] = acc;
}

// The same gather with uniform control flow: a launch handles ONE parity class of (ot, oh), a thread the output pair
// (ow = 2 qw, 2 qw + 1) of one (n, ot, oh) row, so the term lists are compile-time (3 x 3 / 3 x 2 / 2 x 3 / 2 x 2 (t, h) pairs, five w
// loads each), every load of a wave reads consecutive iw of one slot plane, out-of-range inputs are clamped loads times zero, and
// the two outputs leave as one 8-byte store.  o even: (i, k5) = (o/2 + 1, 0), (o/2, 2), (o/2 - 1, 4); o odd: ((o+1)/2, 1), ((o-1)/2, 3).
template <int PT, int PH>
__global__ __launch_bounds__(128) void tail6_gather_rows_kernel(const float* __restrict__ cols, const float* __restrict__ bc, const float* __restrict__ bsm,
                                                                int N, int It, int Ih, int Iw, float* __restrict__ out) {
    __shared__ float sbc[28];
    const int Ot = 2 * It, Oh = 2 * Ih, Ow = 2 * Iw;
    int row = blockIdx.x;                                   // (n, qt, qh)
    const int qh = row % Ih; row /= Ih;
    const int qt = row % It; const int n = row / It;
    const int ot = 2 * qt + PT, oh = 2 * qh + PH;
    if (threadIdx.x < 27) sbc[threadIdx.x] = bc[n * J32 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int e = 0; e < 27; ++e) t += sbc[e]; sbc[27] = t; }
    __syncthreads();
    constexpr int NT = PT ? 2 : 3, NH = PH ? 2 : 3;
    const size_t P3 = (size_t)It * Ih * Iw;
    const float* base = cols + (size_t)n * SP * P3;
    const bool tb = ot == 0 || ot == Ot - 1, hb = oh == 0 || oh == Oh - 1;
    for (int qw = threadIdx.x; qw < Iw; qw += 128) {
        float acc0 = 0.f, acc1 = 0.f;                       // ow = 2 qw, 2 qw + 1
        const int iwp = qw + 1 < Iw ? qw + 1 : qw, iwm = qw >= 1 ? qw - 1 : 0;
        const float fwp = qw + 1 < Iw ? 1.f : 0.f, fwm = qw >= 1 ? 1.f : 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int it_ = PT ? qt + 1 - a : qt + 1 - a, kt = PT ? 1 + 2 * a : 2 * a;
            const bool vt = it_ >= 0 && it_ < It;
            const int itc = vt ? it_ : 0;
#pragma unroll
            for (int b = 0; b < NH; ++b) {
                const int ih_ = qh + 1 - b, kh = PH ? 1 + 2 * b : 2 * b;
                const bool vh = ih_ >= 0 && ih_ < Ih;
                const int ihc = vh ? ih_ : 0;
                const float f = (vt && vh) ? 1.f : 0.f;
                const float* pl = base + (size_t)((kt * 5 + kh) * 5) * P3 + ((size_t)itc * Ih + ihc) * Iw;
                const float e0 = pl[iwp], e2 = pl[2 * P3 + qw], e4 = pl[4 * P3 + iwm];          // even ow: k5w = 0, 2, 4
                const float o1 = pl[P3 + iwp], o3 = pl[3 * P3 + qw];                                // odd ow: k5w = 1, 3
                acc0 += f * ((fwp * e0 + e2) + fwm * e4);
                acc1 += f * (fwp * o1 + o3);
            }
        }
        // bias terms: every (kt, kh, kw) of `smooth` whose intermediate voxel o + 1 - k exists; all 27 away from the borders
        float b0, b1;
        const bool w0b = qw == 0, w1b = qw == Iw - 1;
        if (!tb && !hb && !w0b && !w1b) { b0 = b1 = sbc[27]; }
        else {
            b0 = b1 = 0.f;
            for (int a = 0; a < 3; ++a) {
                const int mt = ot + 1 - a;
                if (mt < 0 || mt >= Ot) continue;
                for (int b = 0; b < 3; ++b) {
                    const int mh = oh + 1 - b;
                    if (mh < 0 || mh >= Oh) continue;
                    for (int c = 0; c < 3; ++c) {
                        const int m0 = 2 * qw + 1 - c, m1 = 2 * qw + 2 - c;
                        const float v = sbc[(a * 3 + b) * 3 + c];
                        if (m0 >= 0 && m0 < Ow) b0 += v;
                        if (m1 >= 0 && m1 < Ow) b1 += v;
                    }
                }
            }
        }
        const float bs = bsm[0];
        *(float2*)(out + (((size_t)n * Ot + ot) * Oh + oh) * Ow + 2 * qw) = make_float2(bs + b0 + acc0, bs + b1 + acc1);
    }
}

// dcols[n][i][slot] = dout[n][2i - 2 + k5] where that output exists, else 0
__global__ __launch_bounds__(256) void tail6_scatter_kernel(const float* __restrict__ dout, int N, int It, int Ih, int Iw, float* __restrict__ dcols) {
    const int Ot = 2 * It, Oh = 2 * Ih, Ow = 2 * Iw;
    const int64_t total = (int64_t)N * It * Ih * Iw * (SP / 4);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q4 = (int)(idx % (SP / 4));
    const int64_t pos = idx / (SP / 4);
    const int iw = (int)(pos % Iw), ih = (int)((pos / Iw) % Ih), it = (int)((pos / ((int64_t)Iw * Ih)) % It), n = (int)(pos / ((int64_t)Iw * Ih * It));
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int slot = q4 * 4 + e;
        v[e] = 0.f;
        if (slot >= NSLOT) continue;
        const int o0 = 2 * it - 2 + slot / 25, o1 = 2 * ih - 2 + (slot / 5) % 5, o2 = 2 * iw - 2 + slot % 5;
        if (o0 >= 0 && o0 < Ot && o1 >= 0 && o1 < Oh && o2 >= 0 && o2 < Ow) v[e] = dout[(((size_t)n * Ot + o0) * Oh + o1) * Ow + o2];
    }
    *(float4*)(dcols + pos * SP + q4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// dW5 [N][8 z][Ci][128 slot] -> Gc [N][Ci][27 tap][32 j]: component (k4, ks) sits in slot k4 + ks of every class except
// the classes whose index is 0 in a dimension where k4 = 0 and ks = 2
// sl.image == 0: dW5 as above (pc_conv_wgrad's atomics summed the K slices).  Otherwise dW5 is the classes' K-slice workspaces back to back,
// [z][k < sl.n[z]][N][Ci][128] (class z at sl.zoff[z], pc_wgrad_desc.ws_slices); tail6_slices_sum_kernel has left every class's sum in its image 0.
struct T6Slices { int n[8]; long long zoff[8]; long long image; };

// image 0 of class blockIdx.y += images 1 .. n-1, in slice order (16 bytes per thread, eight loads in flight): the streaming half of the ordered sum,
// so that the map below gathers its <= 8 scattered values per element from one image per class instead of from every slice
__global__ __launch_bounds__(256) void tail6_slices_sum_kernel(float* __restrict__ ws, const T6Slices sl) {
    const int z = blockIdx.y, ns = sl.n[z];
    if (ns < 2) return;
    float4* base = (float4*)(ws + sl.zoff[z]);
    const long long img4 = sl.image / 4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < img4; e += (long long)gridDim.x * 256) {
        float4 v = base[e];
        int k = 1;
        for (; k + 8 <= ns; k += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = base[(long long)(k + u) * img4 + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
        }
        for (; k < ns; ++k) { const float4 t = base[(long long)k * img4 + e]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        base[e] = v;
    }
}
__global__ __launch_bounds__(256) void tail6_wgrad_map_kernel(const float* __restrict__ dW5, int N, int Ci, float* __restrict__ Gc, const T6Slices sl) {
    const int64_t total = (int64_t)N * Ci * 27 * J32;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx & 31);
        const int tap = (int)((idx >> 5) % 27);
        const int ci = (int)((idx / (27 * J32)) % Ci);
        const int n = (int)(idx / ((int64_t)27 * J32 * Ci));
        float acc = 0.f;
        if (j < 27) {
            const int k4[3] = {tap / 9, (tap / 3) % 3, tap % 3}, ks[3] = {j / 9, (j / 3) % 3, j % 3};
            const int slot = ((k4[0] + ks[0]) * 5 + k4[1] + ks[1]) * 5 + k4[2] + ks[2];
            const bool bad[3] = {k4[0] == 0 && ks[0] == 2, k4[1] == 0 && ks[1] == 2, k4[2] == 0 && ks[2] == 2};
            for (int z = 0; z < 8; ++z) {
                if (((z & 4) && bad[0]) || ((z & 2) && bad[1]) || ((z & 1) && bad[2])) continue;
                if (!sl.image) { acc += dW5[(((size_t)n * 8 + z) * Ci + ci) * SP + slot]; continue; }
                if (sl.n[z]) acc += dW5[sl.zoff[z] + ((size_t)n * Ci + ci) * SP + slot];
            }
        }
        Gc[idx] = acc;
    }
}

// sums[n][j = ks] = sum over the outputs o whose tap-ks source position o + 1 - ks lies in the grid of dout[n][o]
__global__ __launch_bounds__(256) void tail6_bias_sums_kernel(const float* __restrict__ dout, int Ot, int Oh, int Ow, int64_t per_block, float* __restrict__ sums,
                                                              float* __restrict__ part) {
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
    if (threadIdx.x < 27) {
        const float v = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        if (part) part[((size_t)n * gridDim.x + blockIdx.x) * J32 + threadIdx.x] = v;       // per-block partial row: summed in block order by bias_sums_final
        else atomicAdd(sums + n * J32 + threadIdx.x, v);
    }
}

// sums[n][j] = sum over the blocks' partial rows, in block order (no atomics: bit-identical from run to run)
__global__ __launch_bounds__(32) void tail6_bias_sums_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ sums) {
    const int n = blockIdx.x, j = threadIdx.x;
    float v = 0.f;
    if (j < 27) {
        const float* q = part + (size_t)n * nb * J32 + j;
        int b = 0;
        for (; b + 14 <= nb; b += 14) {          // fourteen loads in flight (nb = 98 at 8 x 224 x 224), added in block order
            float t[14];
#pragma unroll
            for (int u = 0; u < 14; ++u) t[u] = q[(size_t)(b + u) * J32];
#pragma unroll
            for (int u = 0; u < 14; ++u) v += t[u];
        }
        for (; b < nb; ++b) v += q[(size_t)b * J32];
    }
    sums[n * J32 + j] = v;
}

}  // namespace

extern "C" int pc_tail6_weights(const float* wf, int N, int Ci, float* W5f, float* W5t, pc_stream s) {
    PC_CHECK_ARG(wf && W5f && W5t && N >= 1 && Ci >= 1, "pc_tail6_weights: bad args");
    const int64_t total = (int64_t)N * 8 * SP * Ci;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tail6_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, wf, N, Ci, W5f, W5t);
    PC_CHECK_LAUNCH("tail6_weights");
    return PC_OK;
}

extern "C" int pc_tail6_gather(const float* cols, const float* bc, const float* bsm, int N, int It, int Ih, int Iw, float* out, pc_stream s) {
    PC_CHECK_ARG(cols && bc && bsm && out && N >= 1 && It >= 1 && Ih >= 1 && Iw >= 1, "pc_tail6_gather: bad args");
    const int64_t total = (int64_t)N * 8 * It * Ih * Iw;
    PC_CHECK_ARG((total + 255) / 256 < (1ll << 31), "pc_tail6_gather: too large");
    static const int rows_env = getenv("PICONS_TAIL6_GATHER_ROWS") ? atoi(getenv("PICONS_TAIL6_GATHER_ROWS")) : 1;
    if (rows_env && ((uintptr_t)out & 7) == 0) {
        const dim3 grid((unsigned)((int64_t)N * It * Ih));
        hipLaunchKernelGGL((tail6_gather_rows_kernel<0, 0>), grid, dim3(128), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
        hipLaunchKernelGGL((tail6_gather_rows_kernel<0, 1>), grid, dim3(128), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
        hipLaunchKernelGGL((tail6_gather_rows_kernel<1, 0>), grid, dim3(128), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
        hipLaunchKernelGGL((tail6_gather_rows_kernel<1, 1>), grid, dim3(128), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
    } else {
        hipLaunchKernelGGL(tail6_gather_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, cols, bc, bsm, N, It, Ih, Iw, out);
    }
    PC_CHECK_LAUNCH("tail6_gather");
    return PC_OK;
}

extern "C" int pc_tail6_scatter(const float* dout, int N, int It, int Ih, int Iw, float* dcols, pc_stream s) {
    PC_CHECK_ARG(dout && dcols && N >= 1 && It >= 1 && Ih >= 1 && Iw >= 1, "pc_tail6_scatter: bad args");
    const int64_t total = (int64_t)N * It * Ih * Iw * (SP / 4);
    PC_CHECK_ARG((total + 255) / 256 < (1ll << 31), "pc_tail6_scatter: too large");
    hipLaunchKernelGGL(tail6_scatter_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, dout, N, It, Ih, Iw, dcols);
    PC_CHECK_LAUNCH("tail6_scatter");
    return PC_OK;
}

static int t6_wgrad_map(float* dW5, const int32_t* nslices8, int N, int Ci, float* Gc, pc_stream s) {
    PC_CHECK_ARG(dW5 && Gc && N >= 1 && Ci >= 1, "pc_tail6_wgrad_map: bad args");
    T6Slices sl;
    sl.image = nslices8 ? (long long)N * Ci * SP : 0;
    long long at = 0;
    for (int z = 0; z < 8; ++z) {
        sl.n[z] = nslices8 ? nslices8[z] : 1;
        PC_CHECK_ARG(sl.n[z] >= 0 && sl.n[z] <= 65536, "pc_tail6_wgrad_map_slices: class %d has %d slice images", z, sl.n[z]);
        sl.zoff[z] = at;
        at += sl.n[z] * sl.image;
    }
    const int64_t total = (int64_t)N * Ci * 27 * J32;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    if (sl.image) {
        int g4 = (int)((sl.image / 4 + 255) / 256); if (g4 > 512) g4 = 512;
        hipLaunchKernelGGL(tail6_slices_sum_kernel, dim3(g4, 8), dim3(256), 0, (hipStream_t)s, dW5, sl);
    }
    hipLaunchKernelGGL(tail6_wgrad_map_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, dW5, N, Ci, Gc, sl);
    PC_CHECK_LAUNCH("tail6_wgrad_map");
    return PC_OK;
}

extern "C" int pc_tail6_wgrad_map_slices(float* ws, const int32_t* nslices8, int N, int Ci, float* Gc, pc_stream s) {
    PC_CHECK_ARG(nslices8, "pc_tail6_wgrad_map_slices: no slice counts");
    return t6_wgrad_map(ws, nslices8, N, Ci, Gc, s);
}

extern "C" int pc_tail6_wgrad_map(const float* dW5, int N, int Ci, float* Gc, pc_stream s) {
    return t6_wgrad_map(const_cast<float*>(dW5), nullptr, N, Ci, Gc, s);
}

namespace {
inline void t6_bias_blocks(int It, int Ih, int Iw, int64_t& pb, int& nb) {
    const int64_t per_n = (int64_t)8 * It * Ih * Iw;
    pb = (per_n + 255) / 256; if (pb < 4096) pb = 4096;
    nb = (int)((per_n + pb - 1) / pb);
}
}  // namespace

extern "C" int64_t pc_tail6_bias_sums_ws_floats(int N, int It, int Ih, int Iw) {
    if (N < 1 || It < 1 || Ih < 1 || Iw < 1) return -1;
    int64_t pb; int nb;
    t6_bias_blocks(It, Ih, Iw, pb, nb);
    return (int64_t)N * nb * J32;
}

// ws != NULL (pc_tail6_bias_sums_ws_floats floats): per-block partial rows + a fixed-order final sum instead of fp32 atomics
extern "C" int pc_tail6_bias_sums_ws(const float* dout, int N, int It, int Ih, int Iw, float* sums, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(dout && sums && N >= 1 && N <= 65535, "pc_tail6_bias_sums: bad args");
    int64_t pb; int nb;
    t6_bias_blocks(It, Ih, Iw, pb, nb);
    if (!ws) (void)hipMemsetAsync(sums, 0, sizeof(float) * N * J32, s);
    hipLaunchKernelGGL(tail6_bias_sums_kernel, dim3(nb, N), dim3(256), 0, s, dout, 2 * It, 2 * Ih, 2 * Iw, pb, sums, ws);
    if (ws) hipLaunchKernelGGL(tail6_bias_sums_final_kernel, dim3(N), dim3(32), 0, s, ws, nb, sums);
    PC_CHECK_LAUNCH("tail6_bias_sums");
    return PC_OK;
}

extern "C" int pc_tail6_bias_sums(const float* dout, int N, int It, int Ih, int Iw, float* sums, pc_stream s) {
    return pc_tail6_bias_sums_ws(dout, N, It, Ih, Iw, sums, nullptr, s);
}
