// Row-spectral form of PrimaryCaps (capsules_ucf101.py:43-49, a 9x9 stride-1 Conv2d 832 -> 512+32 on 28x28):
// a length-P real DFT along the image rows turns the kx taps into a per-frequency product, leaving a 9-tap
// conv along y with complex channels per frequency u = 0..P/2.  With the three-multiplication form of the
// complex product that is one grouped REAL conv (3 groups per frequency, Ci -> Co channels, 9x1 taps) run by
// the ordinary gather-GEMM kernels: 4x fewer multiply-adds than the 81-tap direct form, equal to it in exact
// arithmetic.  This file holds the three small
// HBM-bound kernels around those GEMMs: a dense matrix applied along one tensor axis (DFT, inverse DFT and
// their transposes), the weight spectrum in the GEMM layouts, and its adjoint.
#include "common.h"

namespace {

struct AxK {
    const float* in; const float* M; const float* bias; float* out;
    int R, I, O, C4;
    int in_split, out_split, act, act_c0, accum;
    int in_sr, in_hi, in_lo, out_sr, out_hi, out_lo;
};

constexpr int AX_OC = 8;        // outputs accumulated per pass over the input axis

__global__ __launch_bounds__(256) void axis_linear_kernel(const AxK p) {
    extern __shared__ float Ms[];                       // [O][I]
    for (int k = threadIdx.x; k < p.O * p.I; k += 256) Ms[k] = p.M[k];
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.R * p.C4) return;
    const int r = (int)(idx / p.C4), c = (int)(idx - (int64_t)r * p.C4) * 4;
    const float* ip = p.in + (int64_t)r * p.in_sr + c;
    float* op = p.out + (int64_t)r * p.out_sr + c;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b = *(const float4*)(p.bias + c);
    {   // blockIdx.y picks the chunk of AX_OC outputs: more, shorter threads (the tensors are small and the kernel latency-bound)
        const int o0 = blockIdx.y * AX_OC;
        float4 acc[AX_OC];
#pragma unroll
        for (int q = 0; q < AX_OC; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < p.I; ++i) {
            const float4 v = *(const float4*)(ip + (int64_t)(i / p.in_split) * p.in_hi + (int64_t)(i % p.in_split) * p.in_lo);
#pragma unroll
            for (int q = 0; q < AX_OC; ++q) {
                const float m = (o0 + q < p.O) ? Ms[(o0 + q) * p.I + i] : 0.f;
                acc[q].x += m * v.x; acc[q].y += m * v.y; acc[q].z += m * v.z; acc[q].w += m * v.w;
            }
        }
#pragma unroll
        for (int q = 0; q < AX_OC; ++q) {
            const int o = o0 + q;
            if (o >= p.O) break;
            float* dst = op + (int64_t)(o / p.out_split) * p.out_hi + (int64_t)(o % p.out_split) * p.out_lo;
            float4 y = acc[q];
            y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
            if (p.accum) {
                const float4 old = *(const float4*)dst;
                y.x += old.x; y.y += old.y; y.z += old.z; y.w += old.w;
            }
            if (p.act == PC_ACT_SIGMOID && c >= p.act_c0) {
                y.x = 1.f / (1.f + expf(-y.x)); y.y = 1.f / (1.f + expf(-y.y));
                y.z = 1.f / (1.f + expf(-y.z)); y.w = 1.f / (1.f + expf(-y.w));
            } else if (p.act == PC_ACT_RELU && c >= p.act_c0) {
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
            }
            *(float4*)dst = y;
        }
    }
}

constexpr int WS_MAXK = 16;     // tap counts 1..16 are instantiated (the tap arrays must stay in registers)

// in [A][KY*KX][B] -> planes [A][KY][B]: three per complex frequency (V0 = Wr, V1 = Wr - Wi, V2 = Wr + Wi), then ONE (Wr)
// for each of the Ur trailing frequencies whose spectrum is real (DC, Nyquist)
template <int KX>
__global__ __launch_bounds__(256) void wspec_fwd_kernel(const float* __restrict__ in, const float* __restrict__ tw, int A, int B4, int KY,
                                                         int U, int Ur, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)A * KY * B4) return;
    const int b = (int)(idx % B4) * 4;
    const int ky = (int)((idx / B4) % KY);
    const int a = (int)(idx / ((int64_t)B4 * KY));
    const int B = B4 * 4;
    float4 v[KX];
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) v[kx] = *(const float4*)(in + ((int64_t)a * KY * KX + ky * KX + kx) * B + b);
    const int64_t plane = (int64_t)A * KY * B;               // floats per (u, j) plane [A][KY][B]
    float* o = out + ((int64_t)a * KY + ky) * B + b;
    for (int u = 0; u < U; ++u) {       // (one frequency per block instead re-reads the taps U times: 1.3 vs 0.5 ms)
        float4 wr = make_float4(0.f, 0.f, 0.f, 0.f), wi = wr;
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) {
            const float c = tw[(u * KX + kx) * 2], s = tw[(u * KX + kx) * 2 + 1];
            wr.x += c * v[kx].x; wr.y += c * v[kx].y; wr.z += c * v[kx].z; wr.w += c * v[kx].w;
            wi.x += s * v[kx].x; wi.y += s * v[kx].y; wi.z += s * v[kx].z; wi.w += s * v[kx].w;
        }
        if (u >= U - Ur) { *(float4*)(o + (int64_t)(3 * (U - Ur) + u - (U - Ur)) * plane) = wr; continue; }
        float* ou = o + (int64_t)u * 3 * plane;
        *(float4*)ou = wr;
        *(float4*)(ou + plane) = make_float4(wr.x - wi.x, wr.y - wi.y, wr.z - wi.z, wr.w - wi.w);
        *(float4*)(ou + 2 * plane) = make_float4(wr.x + wi.x, wr.y + wi.y, wr.z + wi.z, wr.w + wi.w);
    }
}

// plane gradients (same order) -> kg [A][KY*KX][B]:  dWr = d0 + d1 + d2, dWi = d2 - d1  (real frequencies: dWr = d0)
template <int KX>
__global__ __launch_bounds__(256) void wspec_bwd_kernel(const float* __restrict__ dV, const float* __restrict__ tw, int A, int B4, int KY,
                                                         int U, int Ur, float* __restrict__ kg) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)A * KY * B4) return;
    const int b = (int)(idx % B4) * 4;
    const int ky = (int)((idx / B4) % KY);
    const int a = (int)(idx / ((int64_t)B4 * KY));
    const int B = B4 * 4;
    float4 acc[KX];
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) acc[kx] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t plane = (int64_t)A * KY * B;
    const float* d = dV + ((int64_t)a * KY + ky) * B + b;
    for (int u = U - Ur; u < U; ++u) {
        const float4 d0 = *(const float4*)(d + (int64_t)(3 * (U - Ur) + u - (U - Ur)) * plane);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) {
            const float c = tw[(u * KX + kx) * 2];
            acc[kx].x += c * d0.x; acc[kx].y += c * d0.y; acc[kx].z += c * d0.z; acc[kx].w += c * d0.w;
        }
    }
#pragma unroll 5
    for (int u = 0; u < U - Ur; ++u) {
        const float* du = d + (int64_t)u * 3 * plane;
        const float4 d0 = *(const float4*)du, d1 = *(const float4*)(du + plane), d2 = *(const float4*)(du + 2 * plane);
        const float4 gr = make_float4(d0.x + d1.x + d2.x, d0.y + d1.y + d2.y, d0.z + d1.z + d2.z, d0.w + d1.w + d2.w);
        const float4 gi = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) {
            const float c = tw[(u * KX + kx) * 2], s = tw[(u * KX + kx) * 2 + 1];
            acc[kx].x += c * gr.x + s * gi.x; acc[kx].y += c * gr.y + s * gi.y;
            acc[kx].z += c * gr.z + s * gi.z; acc[kx].w += c * gr.w + s * gi.w;
        }
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
        *(float4*)(kg + ((int64_t)a * KY * KX + ky * KX + kx) * B + b) = acc[kx];
    }
}

// The same two maps straight from / to the reference's OI(H)W master layout w[A][B][KY][KX] (taps contiguous), so a big
// weight needs no kernel-layout copies at all.  One thread per (a, b) holds all KY*KX taps in registers.
// ALONG_A: consecutive threads walk a (for the [g][B][KY][Atot] dgrad layout), else b (for the [g][Atot][KY][B] forward layout).
// PLANES: the weight planes leave as the three bf16 terms of every value (h, m, l: csrc/conv_x6.hip) at out16 + p * pstride, for the
// bf16-split conv kernel -- 6 bytes per element instead of 4, and no fp32 copy at all.
__device__ __forceinline__ void store_terms(uint16_t* o, long long pstride, float x) {
    const uint32_t u = __float_as_uint(x);
    const uint32_t h = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;          // round to nearest even (finite weights: no NaN handling needed)
    const float r = x - __uint_as_float(h);
    const uint32_t v = __float_as_uint(r);
    const uint32_t m = (v + 0x7fffu + ((v >> 16) & 1u)) & 0xffff0000u;
    const float l = r - __uint_as_float(m);
    o[0] = (uint16_t)(h >> 16); o[pstride] = (uint16_t)(m >> 16); o[2 * pstride] = (uint16_t)(__float_as_uint(l) >> 16);
}

template <int KX, int KY, bool ALONG_A, bool PLANES = false>
__global__ __launch_bounds__(128) void wspec_master_fwd_kernel(const float* __restrict__ w, const float* __restrict__ tw, int Acnt, int a0, int Atot,
                                                                int B, int U, int Ur, float* __restrict__ out, long long pstride = 0) {
    const int fast = blockIdx.x * 128 + threadIdx.x, slow = blockIdx.y;
    const int a = ALONG_A ? fast : slow, b = ALONG_A ? slow : fast;
    if (a >= Acnt || b >= B) return;
    // (ALONG_A: consecutive threads are B * KY * KX floats apart in w -- 3.97 TB/s against the forward layout's 4.67.  Staging the block's 128 rows
    // through LDS made the loads contiguous and the launch SLOWER, 0.289 -> 0.369 ms: 41 KB of LDS per 128 threads leaves 1.5 waves per SIMD for
    // an HBM-bound kernel.  profiles/r06_wspec_master.txt)
    float v[KY][KX];
    const float* src = w + ((size_t)a * B + b) * (KY * KX);
#pragma unroll
    for (int ky = 0; ky < KY; ++ky)
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) v[ky][kx] = src[ky * KX + kx];
    // element (g, ky) of this (a, b): forward layout [g][Atot][KY][B], dgrad layout [g][B][KY][Atot]
    const int64_t plane = (int64_t)Atot * KY * B;
    const int64_t base = ALONG_A ? ((int64_t)b * KY) * Atot + a0 + a : ((int64_t)(a0 + a) * KY) * B + b;
    const int64_t kystep = ALONG_A ? Atot : B;
    for (int u = 0; u < U; ++u) {
        const bool real = u >= U - Ur;
        const int64_t g0 = real ? 3 * (U - Ur) + (u - (U - Ur)) : 3 * u;
#pragma unroll
        for (int ky = 0; ky < KY; ++ky) {
            float wr = 0.f, wi = 0.f;
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) { wr += tw[(u * KX + kx) * 2] * v[ky][kx]; wi += tw[(u * KX + kx) * 2 + 1] * v[ky][kx]; }
            if constexpr (PLANES) {
                uint16_t* o = (uint16_t*)out + g0 * plane + base + ky * kystep;
                store_terms(o, pstride, wr);
                if (!real) { store_terms(o + plane, pstride, wr - wi); store_terms(o + 2 * plane, pstride, wr + wi); }
            } else {
                float* o = out + g0 * plane + base + ky * kystep;
                o[0] = wr;
                if (!real) { o[plane] = wr - wi; o[2 * plane] = wr + wi; }
            }
        }
    }
}

// dV planes in the forward layout [g][Atot][KY][B] -> dw[A][B][KY][KX] (+)= ...
// Block = (a, 64 consecutive b), wave = one ky: a wave-load is 256 contiguous bytes of one plane row; a thread holds KX accumulators.  The block's
// results are one CONTIGUOUS piece of dw (64 x KY x KX floats): they meet in LDS ([b][ky][kx], written at a stride of 81 floats: conflict-free) and
// leave as coalesced stores.  (Rounds 2 - 5: thread = (a, b) with all 81 accumulators, every store instruction 64 scattered 4-byte writes 324 bytes
// apart: 2.45 TB/s.)  Same sums in the same order: bit-identical.
template <int KX, int KY>
__global__ __launch_bounds__(64 * KY) void wspec_master_bwd_kernel(const float* __restrict__ dV, const float* __restrict__ tw, int Acnt, int a0, int Atot,
                                                                    int B, int U, int Ur, float* __restrict__ dw, int accum) {
    __shared__ float tile[64 * KY * KX];
    const int bl = threadIdx.x & 63, ky = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 64, a = blockIdx.y, b = b0 + bl;
    if (b < B) {
        float acc[KX];
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) acc[kx] = 0.f;
        const int64_t plane = (int64_t)Atot * KY * B;
        const float* d = dV + ((int64_t)(a0 + a) * KY + ky) * B + b;
        for (int u = 0; u < U; ++u) {
            const bool real = u >= U - Ur;
            const int64_t g0 = real ? 3 * (U - Ur) + (u - (U - Ur)) : 3 * u;
            const float* du = d + g0 * plane;
            const float d0 = du[0], d1 = real ? 0.f : du[plane], d2 = real ? 0.f : du[2 * plane];
            const float gr = d0 + d1 + d2, gi = d2 - d1;
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) acc[kx] += tw[(u * KX + kx) * 2] * gr + tw[(u * KX + kx) * 2 + 1] * gi;
        }
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) tile[bl * (KY * KX) + ky * KX + kx] = acc[kx];
    }
    __syncthreads();
    const int n = min(64, B - b0) * (KY * KX);
    float* dst = dw + ((size_t)a * B + b0) * (KY * KX);
    for (int e = threadIdx.x; e < n; e += 64 * KY) dst[e] = (accum ? dst[e] : 0.f) + tile[e];
}

}  // namespace

extern "C" int pc_axis_linear(const pc_axis_desc* d, const float* in, const float* M, const float* bias, float* out, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(d && in && M && out, "pc_axis_linear: null pointer");
    PC_CHECK_ARG(d->R > 0 && d->I > 0 && d->O > 0 && d->C > 0 && d->C % 4 == 0, "pc_axis_linear: bad extents (R=%d I=%d O=%d C=%d)", d->R, d->I, d->O, d->C);
    PC_CHECK_ARG(d->in_split >= 1 && d->out_split >= 1, "pc_axis_linear: split < 1");
    PC_CHECK_ARG(d->O * d->I * 4 <= 48 * 1024, "pc_axis_linear: matrix too large for LDS");
    PC_CHECK_ARG((d->in_sr | d->in_hi | d->in_lo | d->out_sr | d->out_hi | d->out_lo) % 4 == 0, "pc_axis_linear: strides must be multiples of 4 floats");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0), "pc_axis_linear: 16-byte alignment");
    AxK k;
    k.in = in; k.M = M; k.bias = bias; k.out = out;
    k.R = d->R; k.I = d->I; k.O = d->O; k.C4 = d->C / 4;
    k.in_split = d->in_split; k.out_split = d->out_split; k.act = d->act; k.act_c0 = d->act_c0; k.accum = d->accum;
    k.in_sr = d->in_sr; k.in_hi = d->in_hi; k.in_lo = d->in_lo; k.out_sr = d->out_sr; k.out_hi = d->out_hi; k.out_lo = d->out_lo;
    const int64_t n = (int64_t)k.R * k.C4;
    hipLaunchKernelGGL(axis_linear_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)cdiv(d->O, AX_OC)), dim3(256), (size_t)d->O * d->I * 4, s, k);
    PC_CHECK_LAUNCH("axis_linear_kernel");
    return PC_OK;
}

static int wspec_check(const void* a, const void* b, const void* c, int A, int B, int KY, int KX, int U, int Ur, const char* who) {
    PC_CHECK_ARG(a && b && c, "%s: null pointer", who);
    PC_CHECK_ARG(A > 0 && B > 0 && B % 4 == 0 && KY > 0 && KX > 0 && KX <= WS_MAXK && U > 0 && Ur >= 0 && Ur <= U, "%s: bad extents (A=%d B=%d KY=%d KX=%d U=%d Ur=%d)", who, A, B, KY, KX, U, Ur);
    PC_CHECK_ARG(((uintptr_t)a % 16 == 0) && ((uintptr_t)c % 16 == 0), "%s: 16-byte alignment", who);
    return PC_OK;
}

extern "C" int pc_wspec_fwd(const float* in, const float* tw, int A, int B, int KY, int KX, int U, int Ur, float* out, pc_stream s_) {
    const int rc = wspec_check(in, tw, out, A, B, KY, KX, U, Ur, "pc_wspec_fwd");
    if (rc != PC_OK) return rc;
    const int64_t n = (int64_t)A * KY * (B / 4);
    const dim3 grid((unsigned)cdiv(n, 256));
#define WS_CASE(K) case K: hipLaunchKernelGGL(wspec_fwd_kernel<K>, grid, dim3(256), 0, (hipStream_t)s_, in, tw, A, B / 4, KY, U, Ur, out); break;
    switch (KX) { WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(7) WS_CASE(8) WS_CASE(9) WS_CASE(10) WS_CASE(11)
                  WS_CASE(12) WS_CASE(13) WS_CASE(14) WS_CASE(15) WS_CASE(16) }
#undef WS_CASE
    PC_CHECK_LAUNCH("wspec_fwd_kernel");
    return PC_OK;
}

extern "C" int pc_wspec_bwd(const float* dV, const float* tw, int A, int B, int KY, int KX, int U, int Ur, float* kg, pc_stream s_) {
    const int rc = wspec_check(dV, tw, kg, A, B, KY, KX, U, Ur, "pc_wspec_bwd");
    if (rc != PC_OK) return rc;
    const int64_t n = (int64_t)A * KY * (B / 4);
    const dim3 grid((unsigned)cdiv(n, 256));
#define WS_CASE(K) case K: hipLaunchKernelGGL(wspec_bwd_kernel<K>, grid, dim3(256), 0, (hipStream_t)s_, dV, tw, A, B / 4, KY, U, Ur, kg); break;
    switch (KX) { WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(7) WS_CASE(8) WS_CASE(9) WS_CASE(10) WS_CASE(11)
                  WS_CASE(12) WS_CASE(13) WS_CASE(14) WS_CASE(15) WS_CASE(16) }
#undef WS_CASE
    PC_CHECK_LAUNCH("wspec_bwd_kernel");
    return PC_OK;
}

extern "C" int pc_wspec_master_fwd(const float* w, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                                   float* out_f, float* out_t, pc_stream s_) {
    PC_CHECK_ARG(w && tw && (out_f || out_t) && Acnt > 0 && a0 >= 0 && a0 + Acnt <= Atot && B > 0 && U > 0 && Ur >= 0 && Ur <= U,
                 "pc_wspec_master_fwd: bad args");
    PC_CHECK_ARG(KY == 9 && KX == 9, "pc_wspec_master_fwd: only the 9x9 PrimaryCaps kernel is instantiated (KY=%d KX=%d)", KY, KX);
    hipStream_t s = (hipStream_t)s_;
    if (out_f) hipLaunchKernelGGL((wspec_master_fwd_kernel<9, 9, false>), dim3(cdiv(B, 128), Acnt), dim3(128), 0, s, w, tw, Acnt, a0, Atot, B, U, Ur, out_f);
    if (out_t) hipLaunchKernelGGL((wspec_master_fwd_kernel<9, 9, true>), dim3(cdiv(Acnt, 128), B), dim3(128), 0, s, w, tw, Acnt, a0, Atot, B, U, Ur, out_t);
    PC_CHECK_LAUNCH("wspec_master_fwd_kernel");
    return PC_OK;
}

extern "C" int pc_wspec_master_planes(const float* w, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                                      uint16_t* out_f, uint16_t* out_t, int64_t plane_stride, pc_stream s_) {
    PC_CHECK_ARG(w && tw && (out_f || out_t) && Acnt > 0 && a0 >= 0 && a0 + Acnt <= Atot && B > 0 && U > 0 && Ur >= 0 && Ur <= U && plane_stride > 0,
                 "pc_wspec_master_planes: bad args");
    PC_CHECK_ARG(KY == 9 && KX == 9, "pc_wspec_master_planes: only the 9x9 PrimaryCaps kernel is instantiated (KY=%d KX=%d)", KY, KX);
    hipStream_t s = (hipStream_t)s_;
    if (out_f) hipLaunchKernelGGL((wspec_master_fwd_kernel<9, 9, false, true>), dim3(cdiv(B, 128), Acnt), dim3(128), 0, s, w, tw, Acnt, a0, Atot, B, U, Ur, (float*)out_f, (long long)plane_stride);
    if (out_t) hipLaunchKernelGGL((wspec_master_fwd_kernel<9, 9, true, true>), dim3(cdiv(Acnt, 128), B), dim3(128), 0, s, w, tw, Acnt, a0, Atot, B, U, Ur, (float*)out_t, (long long)plane_stride);
    PC_CHECK_LAUNCH("wspec_master_fwd_kernel(planes)");
    return PC_OK;
}

extern "C" int pc_wspec_master_bwd(const float* dV, const float* tw, int Acnt, int a0, int Atot, int B, int KY, int KX, int U, int Ur,
                                   float* dw, int accum, pc_stream s_) {
    PC_CHECK_ARG(dV && tw && dw && Acnt > 0 && a0 >= 0 && a0 + Acnt <= Atot && B > 0 && U > 0 && Ur >= 0 && Ur <= U, "pc_wspec_master_bwd: bad args");
    PC_CHECK_ARG(KY == 9 && KX == 9, "pc_wspec_master_bwd: only the 9x9 PrimaryCaps kernel is instantiated (KY=%d KX=%d)", KY, KX);
    hipLaunchKernelGGL((wspec_master_bwd_kernel<9, 9>), dim3(cdiv(B, 64), Acnt), dim3(64 * 9), 0, (hipStream_t)s_, dV, tw, Acnt, a0, Atot, B, U, Ur, dw, accum);
    PC_CHECK_LAUNCH("wspec_master_bwd_kernel");
    return PC_OK;
}
